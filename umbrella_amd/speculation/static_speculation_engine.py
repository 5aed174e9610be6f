"""Static (Sequoia growmap) speculation engine on the HIP path.

Same constructor / methods as the reference class
(umbrella/speculation/static_speculation_engine.py:21-566).  Differences by
design: the growmap becomes device tables (depth, parents, bit-packed ancestor
mask, per-level child placement) instead of a dense [Lmax, 2*Lmax] bool mask;
the draft levels, top-k child placement, verify forward, arg-max, accept scan
and KV compaction of one iteration replay as a single hipGraph.
"""
from __future__ import annotations

import json
import os

import torch

from .. import _lib
from ..models.llama import pack_mask_bits
from .engine_common import HipEngine, logger
from ..utils import TextColors


def resolve_growmap_path(path: str) -> str:
    """Reference configs point at ``../umbrella/trees/<name>.json`` relative to its examples directory; when that
    path does not exist here, the tree of the same file name shipped in ``umbrella_amd/trees`` is used."""
    if os.path.exists(path):
        return path
    shipped = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "trees", os.path.basename(path))
    if os.path.exists(shipped):
        return shipped
    raise FileNotFoundError(f"growmap '{path}' not found (also looked for {shipped})")


class StaticSpeculationEngine(HipEngine):
    MASK_FIRST_EOS = False                      # static:173 plain argmax of the last logit

    def __init__(self, draft_model_name: str, target_model_name: str, dtype=torch.float16, device: str = "cuda:0",
                 **kwargs) -> None:
        super().__init__()
        self.draft_model_name, self.target_model_name = draft_model_name, target_model_name
        self.dtype, self.device = dtype, device
        self.growmap_path = kwargs.pop("growmap_path", None)
        self.growmap = kwargs.pop("growmap", None)
        assert self.growmap_path is not None or self.growmap is not None, "Please specify growmap path for static trees"
        # Stochastic verification.  Default (round 5: a drop-in behaves like what it replaces): the reference's own draw --
        # flashinfer's rejection sampler over ONE rand(3, tree_size) tensor taken at initialize() and reused by every
        # verify (static:131,310); an explicit uniform_samples [3, tree_size] tensor gives the reference's tokens draw for
        # draw.  reference_sampler=False: fresh counter-based draws per (seed, position, node) -- umb_sample_rows -- the
        # distribution-equivalent sampler without the frozen uniforms.
        self._uniform_arg = kwargs.pop("uniform_samples", None)
        self.reference_sampler = bool(kwargs.pop("reference_sampler", True)) or self._uniform_arg is not None
        self._common_kwargs(kwargs)
        self.config = kwargs

    def initialize(self):
        if self.growmap is None:
            with open(resolve_growmap_path(self.growmap_path), "r") as f:
                self.growmap = json.load(f)
        gm, dev = self.growmap, self.device
        T = gm["size"]
        levels = gm["roots"]
        self.tree_depth = len(levels)
        self.branch_lists = gm["branches"]
        logger.info(TextColors.colorize("Tree Size {} | Tree Depth {}".format(T - 1, self.tree_depth - 1), "magenta"))
        parents = [0] * T
        for v, succ in enumerate(gm["Successors"]):
            for c in succ:
                parents[c] = v
        self.parents = torch.tensor(parents, dtype=torch.int32, device=dev)
        self.depth = torch.tensor(gm["depth"], dtype=torch.int32, device=dev)
        self.tree_mask = (torch.tensor(gm["mask"]) == 1).to(dev)
        self.mask_bits = pack_mask_bits(self.tree_mask).contiguous()
        self.mask_words = self.mask_bits.shape[1]
        # per-level child placement: children of level i are laid out by (parent order, rank) (static:115-123)
        self.levels = []
        for i, ids in enumerate(levels):
            assert ids == list(range(ids[0], ids[0] + len(ids))), "growmap levels must be contiguous tree offsets"
            w, k = len(ids), (max(self.branch_lists[i]) if i < self.tree_depth - 1 else 0)
            starts, cur = [], (levels[i + 1][0] if i + 1 < self.tree_depth else 0)
            for j in range(w):
                b = self.branch_lists[i][j] if i < self.tree_depth - 1 else 0
                assert gm["Successors"][ids[j]] == list(range(cur, cur + b)), "Successors must match branches order"
                starts.append(cur)
                cur += b
            self.levels.append(dict(off=ids[0], w=w, k=k,
                                    child_start=torch.tensor(starts, dtype=torch.int32, device=dev),
                                    child_cnt=torch.tensor(self.branch_lists[i] if k else [0] * w, dtype=torch.int32, device=dev)))
        self.draft_rows = max(l["w"] for l in self.levels)
        # row counters (zeroed once, self-resetting) + per-part keys of the split top-k (umb_topk_rows_ws)
        self.topk_ws = torch.zeros(4096 + max(l["w"] * max(l["k"], 1) for l in self.levels) * 16 * 8, dtype=torch.uint8,
                                   device=dev)
        self._load_models(dict(offload=False, cuda_graph=True), dict(offload=False))
        self._alloc_state(T, self.tree_depth)
        self.uniform_samples = None
        if self.reference_sampler:
            if self._uniform_arg is not None:
                u = torch.as_tensor(self._uniform_arg, dtype=torch.float32)
                assert u.dim() == 2 and u.shape[1] == T, f"uniform_samples must be [rounds, {T}]"
            else:
                u = torch.rand(3, T, generator=torch.Generator().manual_seed(int(self.seed)))     # static:131
            self.uniform_samples = u.to(dev).contiguous()
        self.override_tbl = torch.full((T,), -1, dtype=torch.int32, device=dev)
        self.override_host = torch.full((T,), -1, dtype=torch.int32).pin_memory()
        self.enable_override = False
        self._succ = gm["Successors"]

    # ---- measurement knob (bench.py): controllable-acceptance draft for synthetic weights ----------
    def set_oracle_draft(self, truth, truth_start: int, acc, seed: int = 0):
        """Random-weight draft/target pairs accept ~0 drafted tokens, so benchmarks on synthetic
        checkpoints steer acceptance explicitly: `truth` is the target's own greedy continuation
        (truth[i] is the token at absolute position truth_start + i).  Before each iteration the host
        marks, for every depth d, the rank-r child of the on-path node (r ~ Categorical(acc), seeded by
        the absolute position) to carry the true token; the device applies the table right after each
        level's top-k.  All draft/verify work still runs; only token ids of <= depth slots change.
        Must be called before the first step() (the override kernels are part of the captured graph)."""
        import numpy as np
        self._truth, self._truth_start = list(truth), truth_start
        p = np.asarray(list(acc) + [max(0.0, 1.0 - float(sum(acc)))], dtype=np.float64)
        self._acc_p, self._acc_seed = p / p.sum(), seed
        # rank drawn for absolute position q (seeded by q alone, so every pass / run sees the same draws); tabulated
        # once -- the per-iteration host work inside a timed loop is then a few list lookups
        lo, hi = truth_start, truth_start + len(self._truth) + self.tree_depth + 1
        self._rank_lo = lo
        self._rank_tbl = [int(np.random.RandomState((seed * 1000003 + q) % (2**31 - 1)).choice(len(self._acc_p), p=self._acc_p))
                          for q in range(lo, hi)]
        self.enable_override = True
        self._graph = None
        self.diverged = 0

    def _fill_override(self):
        tbl = self.override_host
        tbl.fill_(-1)
        i0 = self.num_nodes - self._truth_start
        if getattr(self, "last_bonus", None) is None:
            self.last_bonus = int(self.tokens[self.num_nodes])
        if 0 <= i0 < len(self._truth) and self.last_bonus != self._truth[i0]:
            self.diverged += 1                     # target left the recorded continuation (16-bit near-tie)
        node = 0
        for d in range(1, self.tree_depth):
            if i0 + d >= len(self._truth) or i0 < 0:
                break
            r = self._rank_tbl[self.num_nodes + d - self._rank_lo]
            kids = self._succ[node]
            if r >= len(kids):
                break
            node = kids[r]
            tbl[node] = self._truth[i0 + d]
        self.override_tbl.copy_(tbl, non_blocking=True)

    @torch.inference_mode()
    def build_tree(self):
        d = self.draft_model
        for li, lv in enumerate(self.levels):
            has_head = lv["k"] > 0
            if not has_head and self.lookback:
                continue                              # deepest level: its draft KV is re-derived by the next root forward
            if li == 0:
                self._draft_root()
            else:
                d.forward_tree(self.tokens, self.n_dev, self.depth, lv["off"], lv["w"], self.mask_bits,
                               self.mask_words, head_from=0 if has_head else lv["w"])
            if has_head:
                _lib.call("umb_topk_rows_ws", None, None, d.logits_buffer, lv["w"], self.vocab_size, lv["k"],
                          self.tokens, self.n_dev, lv["child_start"], lv["child_cnt"], self.topk_ws,
                          self.topk_ws.numel())
                if self.enable_override:
                    nxt = lv["off"] + lv["w"]
                    cnt = int(sum(self.branch_lists[self.levels.index(lv)]))
                    _lib.call("umb_apply_override", self.tokens, self.n_dev, self.override_tbl, self.parents, nxt, cnt)

    def _verify_forward(self):
        self.target_model.forward_tree(self.tokens, self.n_dev, self.depth, 0, self.tree_size, self.mask_bits,
                                       self.mask_words, head_from=0)
