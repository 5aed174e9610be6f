"""Tensor-parallel verify across the GPUs of one node (the reference has no distributed code).

Layer sharding (parallel.py) adds capacity; the only route to a faster single request beyond one GPU's HBM is to
split every layer: rank r of P holds
    q / k / v rows of its Hq/P query and Hkv/P key-value heads   (column split: no communication)
    the o_proj columns of those heads                            (row split: fp32 partial [T, H] -> all-reduce)
    gate / up rows [r I/P, (r+1) I/P)                            (column split)
    the down_proj columns of that slice                          (row split: fp32 partial [T, H] -> all-reduce)
    lm_head rows [r V/P, (r+1) V/P)                              (column split: local arg-max, (value, index) pairs gathered)
with norms, the residual stream and the embedding replicated.  Two all-reduces of a [T, H] fp32 tile per layer
(426 KB at T = 13, H = 8192) over RCCL -- xGMI is point to point, so at these sizes the collective is latency
bound and the right degree is small (2-4): 160 all-reduces per 70B verify at ~10-20 us each is the price of
streaming 1/P of the weights.  Unmeasured here (one GPU per box); correct by construction:
  * the sharding helpers are pure torch and are pinned on CPU against the unsharded oracle (gloo, world 2);
  * on the GPU the P shards run in ONE process with an in-process sum standing in for the all-reduce and must
    reproduce the unsharded model up to fp32 summation order (tests/test_tensor_parallel.py).
SPMD: every rank runs the same engine (the 1B draft is replicated -- it is small and deterministic, so all ranks
grow the same tree without talking), and the sharded target forward meets at the collectives.  No control channel.
Kernels: the low-latency GEMM family (csrc/lowlat.hip) on the local shard; T <= 64 rows per forward.
"""
from __future__ import annotations

import copy

import torch

from . import _lib
from .models.config import LlamaCfg
from .models.synthetic import linear_shapes


# ------------------------------------------------------------------ sharding (pure torch: CPU-testable)
def shard_range(n: int, rank: int, world: int):
    assert n % world == 0, (n, world)
    return n // world * rank, n // world * (rank + 1)


def local_config(cfg: LlamaCfg, world: int) -> LlamaCfg:
    """The per-rank architecture: 1/P of the heads, of the MLP width and of the vocabulary."""
    assert cfg.num_attention_heads % world == 0 and cfg.num_key_value_heads % world == 0, "heads must divide by the TP degree"
    assert cfg.intermediate_size % (128 * world) == 0 and cfg.vocab_size % (16 * world) == 0
    c = copy.copy(cfg)
    c.num_attention_heads //= world
    c.num_key_value_heads //= world
    c.intermediate_size //= world
    c.vocab_size //= world
    c.tie_word_embeddings = False
    return c


def shard_tensor(name: str, t: torch.Tensor, cfg: LlamaCfg, rank: int, world: int):
    """This rank's slice of one HF-named tensor (dense ``.weight`` [N, K], or AutoAWQ ``.qweight`` [K, N/8] /
    ``.qzeros`` [K/G, N/8] / ``.scales`` [K/G, N]).  Column-split linears keep their K, row-split ones their N."""
    D, G = cfg.head_dim, cfg.awq_group
    q0, q1 = (x * D for x in shard_range(cfg.num_attention_heads, rank, world))
    k0, k1 = (x * D for x in shard_range(cfg.num_key_value_heads, rank, world))
    i0, i1 = shard_range(cfg.intermediate_size, rank, world)
    v0, v1 = shard_range(cfg.vocab_size, rank, world)
    col = {"self_attn.q_proj": (q0, q1), "self_attn.k_proj": (k0, k1), "self_attn.v_proj": (k0, k1),
           "mlp.gate_proj": (i0, i1), "mlp.up_proj": (i0, i1)}
    row = {"self_attn.o_proj": (q0, q1), "mlp.down_proj": (i0, i1)}
    base, _, leaf = name.rpartition(".")
    lin = next((l for l in list(col) + list(row) if base.endswith(l)), None)
    if name == "lm_head.weight":
        return t[v0:v1]
    if lin in col:
        lo, hi = col[lin]
        if leaf in ("weight", "bias"):
            return t[lo:hi]
        return t[:, lo // 8:hi // 8] if leaf in ("qweight", "qzeros") else t[:, lo:hi]
    if lin in row:
        lo, hi = row[lin]
        if leaf == "weight":
            return t[:, lo:hi]
        if leaf == "qweight":
            return t[lo:hi]
        assert lo % G == 0 and hi % G == 0                 # qzeros / scales: one row per group of G input features
        return t[lo // G:hi // G]
    return t


def shard_state_dict(sd: dict, cfg: LlamaCfg, rank: int, world: int) -> dict:
    """shard_tensor over a whole state dict; tied embeddings get an explicit lm_head slice (the table stays whole)."""
    out = {name: shard_tensor(name, t, cfg, rank, world) for name, t in sd.items()}
    if cfg.tie_word_embeddings and "lm_head.weight" not in sd:
        out["lm_head.weight"] = shard_tensor("lm_head.weight", sd["model.embed_tokens.weight"], cfg, rank, world)
    return {k: v.contiguous() for k, v in out.items()}


class LazySyntheticShard:
    """Mapping name -> this rank's slice of a seeded synthetic checkpoint, generated tensor by tensor on the device
    (a 70B-AWQ state dict does not fit host RAM twice over).  Every rank draws the same full tensor from a generator
    seeded by (seed, name), so the shards of all ranks are slices of one consistent model."""

    def __init__(self, cfg: LlamaCfg, rank: int, world: int, device, dtype, seed: int = 0):
        self.cfg, self.rank, self.world, self.device, self.dtype, self.seed = cfg, rank, world, device, dtype, seed

    def _full(self, name):
        import math
        import zlib
        from .models.synthetic import synth_tensor
        cfg = self.cfg
        gen = torch.Generator(device=self.device).manual_seed((self.seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31 - 1))
        H, V = cfg.hidden_size, cfg.vocab_size
        if name in ("model.embed_tokens.weight", "lm_head.weight"):
            return synth_tensor((V, H), 1.0 if name.startswith("model.embed") else 0.05, self.dtype, self.device, gen)
        if name.endswith("layernorm.weight") or name == "model.norm.weight":
            return torch.ones(H, dtype=self.dtype, device=self.device)
        base, _, leaf = name.rpartition(".")
        lin = next(l for l in linear_shapes(cfg) if base.endswith(l))
        n, k = linear_shapes(cfg)[lin]
        std = 0.02 / (math.sqrt(2.0 * cfg.num_hidden_layers) if lin in ("self_attn.o_proj", "mlp.down_proj") else 1.0)
        if leaf == "weight":
            return synth_tensor((n, k), std, self.dtype, self.device, gen)
        if leaf == "bias":
            return synth_tensor((n,), 0.1, self.dtype, self.device, gen)
        # AWQ triple: one generator stream per linear, so the three tensors are consistent whichever is asked for first
        from .models.synthetic import synth_awq_tensors
        g2 = torch.Generator(device=self.device).manual_seed((self.seed * 1000003 + zlib.crc32(base.encode())) % (2 ** 31 - 1))
        qw, qz, sc = synth_awq_tensors(n, k, cfg.awq_group, self.device, g2, std)
        return {"qweight": qw, "qzeros": qz, "scales": sc}[leaf]

    def __getitem__(self, name):
        if name == "model.embed_tokens.weight":
            return self._full(name)
        src = "model.embed_tokens.weight" if (name == "lm_head.weight" and self.cfg.tie_word_embeddings) else name
        t = self._full(src)
        return shard_tensor(name, t, self.cfg, self.rank, self.world).contiguous()


# ------------------------------------------------------------------ communicators
class DistComm:
    """one shard per process; RCCL ("nccl") on GPUs, gloo on CPU"""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def all_reduce(self, tensors):
        self.dist.all_reduce(tensors[0], group=self.group)

    def gather_max(self, vals, idxs, vocab_per_rank):
        """per-row (max value, global index) over the ranks' vocabulary slices; ties -> lowest index"""
        v, i = vals[0], idxs[0]
        pack = torch.stack([v, (i + self.rank * vocab_per_rank).float()], dim=-1).contiguous()
        allp = [torch.empty_like(pack) for _ in range(self.world)]
        self.dist.all_gather(allp, pack, group=self.group)
        return [_pick(allp)]


class LocalComm:
    """all P shards in this process (single-GPU tests): the all-reduce is a sum over the shard list, in rank order --
    the order a ring all-reduce of a tile this small would be reduced in is not specified, so parity is to fp32 order"""

    def __init__(self, world):
        self.rank, self.world = 0, world

    def all_reduce(self, tensors):
        total = tensors[0].clone()
        for t in tensors[1:]:
            total += t
        for t in tensors:
            t.copy_(total)

    def gather_max(self, vals, idxs, vocab_per_rank):
        allp = [torch.stack([v, (i + r * vocab_per_rank).float()], dim=-1) for r, (v, i) in enumerate(zip(vals, idxs))]
        return [_pick(allp)] * len(vals)


def _pick(allp):
    vals = torch.stack([p[:, 0] for p in allp])            # [P, T]
    idx = torch.stack([p[:, 1] for p in allp])
    # ties -> lowest global vocabulary index, by construction (torch.argmax does not promise which maximal entry it
    # returns): the minimum index among the entries equal to the column maximum; a NaN shard maximum counts as -inf on
    # every rank alike, so all ranks agree on the token.
    vals = torch.nan_to_num(vals, nan=-float("inf"))
    mx = vals.max(dim=0, keepdim=True).values
    cand = torch.where(vals == mx, idx, torch.full_like(idx, float("inf")))
    return cand.min(dim=0).values.to(torch.int32)


# ------------------------------------------------------------------ the sharded target
class TensorParallelLlama:
    """P shards of one Llama target.  ``shards`` are `Llama` objects built on `local_config` / `shard_state_dict`
    (one per process with DistComm, all P with LocalComm); the embedding table is replicated."""

    CHUNK = 64
    PREFILL_CHUNK = 64

    def __init__(self, cfg: LlamaCfg, shards, embed: torch.Tensor, comm):
        self.config, self.shards, self.comm, self.embed = cfg, shards, comm, embed.contiguous()
        self.world = comm.world
        m = shards[0]
        self.device, self.dtype, self.max_length, self.eos_tokens = m.device, m.dtype, m.max_length, list(cfg.eos_token_id)
        self.num_layers = m.num_layers
        self.kv_cache = _ShardedKV([s.kv_cache for s in shards])
        self.sampled_ids = torch.zeros(1024, dtype=torch.int32, device=self.device)
        self._off = None
        T = 64
        self._part = [torch.zeros(T, cfg.hidden_size, dtype=torch.float32, device=self.device) for _ in shards]
        self._xfm = [torch.zeros(T * cfg.hidden_size, dtype=self.dtype, device=self.device) for _ in shards]
        self._topv = [torch.zeros(T, dtype=torch.float32, device=self.device) for _ in shards]
        self._topi = [torch.zeros(T, dtype=torch.int32, device=self.device) for _ in shards]

    @classmethod
    def build(cls, cfg: LlamaCfg, state_dict: dict, world: int, comm, max_length, device, dtype, ranks=None, name="tp"):
        from .models.llama import Llama
        lc = local_config(cfg, world)
        shards = []
        for r in (ranks if ranks is not None else [comm.rank]):
            if isinstance(state_dict, dict):
                sd = dict(shard_state_dict(state_dict, cfg, r, world))
            else:                                           # lazy per-rank mapping (LazySyntheticShard): already sliced
                assert ranks is None or len(ranks) == 1
                sd = state_dict
            m = Llama(f"{name}-shard{r}", max_length=max_length, device=device, dtype=dtype,
                      state_dict=_EmbedPlaceholder(sd, lc.vocab_size), config=lc)
            m.alloc()
            shards.append(m)
        embed = state_dict["model.embed_tokens.weight"].to(device=device, dtype=dtype)
        return cls(cfg, shards, embed, comm)

    def reserve(self, tokens, logit_rows=None):
        pass                                                # forwards run in <= 64-row pieces on the shards' default workspace

    def clear(self):
        for s in self.shards:
            s.clear()

    def weight_bytes(self):
        return self.shards[0].weight_bytes()

    # ---- one <= 64-row forward; returns nothing: arg-max ids of rows [head_from, T) land in self.sampled_ids[:T - head_from]
    def _forward(self, step_args, T, head_from, mask_first_eos=None):
        call, dt = _lib.call, _lib.dtype_code(self.dtype)
        cfg, lc = self.config, self.shards[0].config
        H, tt = cfg.hidden_size, _lib.load().umb_ll_token_tiles(T)
        for sh, part, xfm in zip(self.shards, self._part, self._xfm):
            b = sh._bufs
            tok, pos, slot, prefix, tokens_all, n_ptr, off, depth = step_args
            # embedding gather + index resolution (replicated); hw / ssq outputs of the kernel are not used here
            call("umb_embed_ll", b["h"], self.embed, H, cfg.vocab_size, sh.max_length, T, tok, pos, slot, prefix, tokens_all,
                 n_ptr, off, depth, b["pos"], b["slot"], b["prefix"], b["hw"], sh.norms[0][0], b["ssq"], sh.ssq_stride, dt)
            call("umb_rmsnorm", b["xn"], b["h"], sh.norms[0][0], cfg.rms_norm_eps, T, H, dt)
        for l in range(self.num_layers):
            last = l + 1 == self.num_layers
            for sh, part, xfm in zip(self.shards, self._part, self._xfm):
                b, lins = sh._bufs, sh.layers[l]
                kc, vt = sh.kv_cache.k[l], sh.kv_cache.vt[l]
                call("umb_to_fm", xfm, b["xn"], T, H, dt)
                fq = _lib.UmbGemmLL()
                fq.pos, fq.slot, fq.cosT, fq.sinT = b["pos"].data_ptr(), b["slot"].data_ptr(), sh.cos_cache.data_ptr(), sh.sin_cache.data_ptr()
                fq.q_out, fq.k_cache, fq.vt_cache = b["q"].data_ptr(), kc.data_ptr(), vt.data_ptr()
                fq.bias = sh.qkv_biases[l].data_ptr() if sh.qkv_biases[l] is not None else None
                fq.Hq, fq.Hkv, fq.D, fq.Lmax = lc.num_attention_heads, lc.num_key_value_heads, lc.head_dim, sh.max_length
                q = lins["qkv"]
                call("umb_gemm_ll", None, xfm, q.w, q.meta, T, q.N, q.K, q.awq, 3, fq, dt)
                call("umb_tree_attn2", b["attn"], b["q"], kc, vt, b["po"], b["ml"], b["prefix"], self._mask[0], self._mask[1],
                     self._mask[2], T, lc.num_attention_heads, lc.num_key_value_heads, lc.head_dim, sh.max_length,
                     sh.attn_chunk, sh.attn_splits, 1.0 / (lc.head_dim ** 0.5), sh._attn_counters, tt, dt)
                o = lins["o"]
                call("umb_gemm_ll", part, b["attn"], o.w, o.meta, T, o.N, o.K, o.awq, 0, _lib.UmbGemmLL(), dt)
            self.comm.all_reduce([p[:T] for p in self._part])
            for sh, part, xfm in zip(self.shards, self._part, self._xfm):
                b, lins = sh._bufs, sh.layers[l]
                call("umb_reduce_residual_norm", part, 1, T, H, b["h"], b["h"], b["xn"], sh.norms[l][1], cfg.rms_norm_eps, dt)
                call("umb_to_fm", xfm, b["xn"], T, H, dt)
                gu, dn = lins["gu"], lins["down"]
                call("umb_gemm_ll", b["act"], xfm, gu.w, gu.meta, T, gu.N, gu.K, gu.awq, 2, _lib.UmbGemmLL(), dt)
                call("umb_gemm_ll", part, b["act"], dn.w, dn.meta, T, dn.N, dn.K, dn.awq, 0, _lib.UmbGemmLL(), dt)
            self.comm.all_reduce([p[:T] for p in self._part])
            for sh, part in zip(self.shards, self._part):
                b = sh._bufs
                nxt = sh.norm_weight if last else sh.norms[l + 1][0]
                call("umb_reduce_residual_norm", part, 1, T, H, b["h"], b["h"], b["xn"], nxt, cfg.rms_norm_eps, dt)
        rows = T - head_from
        if rows <= 0:
            return
        Vl = lc.vocab_size
        for r, (sh, xfm, tv, ti) in enumerate(zip(self.shards, self._xfm, self._topv, self._topi)):
            b = sh._bufs
            call("umb_to_fm", xfm, b["xn"], T, H, dt)
            fh = _lib.UmbGemmLL()
            fh.row_from, fh.round_out = head_from, 1
            hd = sh.lm_head
            call("umb_gemm_ll", b["logits"], xfm, hd.w, hd.meta, T, hd.N, hd.K, hd.awq, 0, fh, dt)
            if mask_first_eos:                             # dynamic engine: EOS ids to -inf on the last row before the arg-max
                base = (self.comm.rank if len(self.shards) == 1 else r) * Vl
                loc = [e - base for e in mask_first_eos if base <= e < base + Vl]
                if loc:
                    b["logits"][rows - 1, loc] = -float("inf")
            call("umb_topk_rows", ti, tv, b["logits"], rows, Vl, 1, None, None, None, None)
        ids = self.comm.gather_max([v[:rows] for v in self._topv], [i[:rows] for i in self._topi], Vl)
        self.sampled_ids[:rows] = ids[0]

    # ---- the model-runtime face the engines use
    def forward_tree(self, tokens_all, n_ptr, depth, tree_off, T, mask_bits, mask_words, head_from=0, **kw):
        assert T <= 64, "tensor-parallel verify handles trees of <= 64 nodes (low-latency kernels)"
        self._mask = (mask_bits.data_ptr() + tree_off * mask_words * 8, mask_words, tree_off + T)
        self._keep = (mask_bits,)
        self._forward((None, None, None, None, tokens_all, n_ptr, tree_off, depth), T, head_from)

    def prefill_tokens(self, ids, start, want_logits=True):
        """causal forward in <= 64-row pieces; returns int32[1] = arg-max id of the last row when asked"""
        P = ids.shape[0]
        dev = self.device
        out = None
        for lo in range(0, P, self.CHUNK):
            hi = min(P, lo + self.CHUNK)
            T = hi - lo
            pos = torch.arange(start + lo, start + hi, dtype=torch.int32, device=dev)
            pre = torch.tensor([start + lo], dtype=torch.int32, device=dev)
            last = hi == P and want_logits
            self._mask = (None, 0, T)
            self._keep = (pos, pre)
            self._forward((ids[lo:hi].contiguous(), pos, pos, pre, None, None, 0, None), T, T - 1 if last else T,
                          mask_first_eos=self._first_eos if last else None)
            if last:
                out = self.sampled_ids[:1].clone()
        self.kv_cache.kv_offset = start + P
        return out

    _first_eos = None


class _EmbedPlaceholder:
    """what a shard's `Llama` sees: its slice of everything, and a V/P-row stand-in for the embedding table (the real,
    replicated table lives in TensorParallelLlama.embed; the shard never embeds)"""

    def __init__(self, sd, rows):
        self.sd, self.rows = sd, rows

    def __getitem__(self, name):
        t = self.sd[name]
        return t[:self.rows] if name == "model.embed_tokens.weight" else t


class _ShardedKV:
    def __init__(self, caches):
        self.caches = caches
        self.kv_offset = 0

    def compact(self, result, path, max_path):
        for c in self.caches:
            c.compact(result, path, max_path)

    def clear(self):
        for c in self.caches:
            c.clear()
        self.kv_offset = 0


# ------------------------------------------------------------------ engine
from .speculation.static_speculation_engine import StaticSpeculationEngine as _Static  # noqa: E402


class TensorParallelStaticEngine(_Static):
    """Static (Sequoia) engine over a TensorParallelLlama target.  SPMD: every rank constructs and drives the same
    engine; the draft tree is a hipGraph, the sharded verify is launched eagerly (collectives in between)."""

    def __init__(self, *a, tp_target=None, **kw):
        super().__init__(*a, target_model_obj=tp_target, **kw)

    def initialize(self):
        if not self._greedy():
            raise ValueError("the tensor-parallel engine verifies greedily (the logits stay sharded over the ranks)")
        super().initialize()
        self.graph_scope = "draft"

    def update_generation_args(self, **generation_args):
        super().update_generation_args(**generation_args)
        if not self._greedy():
            raise ValueError("the tensor-parallel engine verifies greedily")

    def _feed(self, lo, hi):
        ids = self.tokens[lo:hi]
        dlo = lo - 1 if (self.lookback and lo > 0) else lo
        self.draft_model.prefill_tokens(self.tokens[dlo:hi], dlo, want_logits=False)
        first = self.target_model.prefill_tokens(ids, lo, want_logits=True)
        self.tokens[hi:hi + 1] = first
        self.num_nodes = hi
        self.n_dev.fill_(hi)
        self.last_bonus = None

    def _sample(self, dbg=None):
        self.sampled.copy_(self.target_model.sampled_ids[:self.tree_size])


def tp_measure(args, wl, dtype, device, rank, world):
    """ONE request, every layer of the target split over the ranks of the default process group (static 3x4, greedy).
    SPMD -- every rank runs this; all return the result dict.  Nothing is printed, the group stays up."""
    import time

    import torch.distributed as dist

    import __graft_entry__ as ge
    from .models.auto_model import AutoModelLM
    from .models.config import KNOWN
    from .sequoia_utils import generate_sequoia_tree
    ge.build()
    cfg = KNOWN[wl["target"]]
    comm = DistComm()
    tp = TensorParallelLlama.build(cfg, LazySyntheticShard(cfg, rank, world, device, dtype, seed=args.seed), world, comm,
                                   args.max_length, device, dtype, name=wl["target"])
    draft = AutoModelLM.from_pretrained(wl["draft"], max_length=args.max_length, device=device, dtype=dtype, cuda_graph=True)
    draft.alloc(exit_layer=16)
    eng = TensorParallelStaticEngine(wl["draft"], wl["target"], dtype=dtype, device=device, growmap=generate_sequoia_tree(3, 4),
                                     max_length=args.max_length, draft_model_obj=draft, tp_target=tp, seed=args.seed)
    eng.initialize()
    g = torch.Generator().manual_seed(1234)
    prompt = torch.randint(3, wl.get("vocab_hi", 128000), (1, args.prompt_len), generator=g)
    assert eng._prefill(prompt)
    for _ in range(args.warmup):
        eng.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    start = eng.num_nodes
    t0 = time.time()
    for _ in range(args.steps):
        eng.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.time() - t0
    tokens = eng.num_nodes - start
    return {"ms_per_step": round(dt / args.steps * 1e3, 4), "tokens_per_s_raw_draft": round(tokens / dt, 2),
            "accept_len_raw_draft": round(tokens / args.steps, 3), "n_ranks_rccl": world,
            "backend": dist.get_backend() if dist.is_initialized() else "none",
            "allreduces_per_verify": 2 * cfg.num_hidden_layers, "allreduce_bytes": eng.tree_size * cfg.hidden_size * 4,
            "parallelism": f"tp{world}: heads / MLP width / vocabulary split, 2 all-reduces of the [T, H] fp32 tile per "
                           "layer; draft replicated", "tree": "3x4", "scaling": "strong"}


def run_tp_bench(args, wl, dtype, device, rank, world):
    """bench.py --parallel tp: the tensor-parallel engine alone, one JSON line from rank 0."""
    import json

    import torch.distributed as dist
    r = tp_measure(args, wl, dtype, device, rank, world)
    out = None
    if rank == 0:
        out = {"metric": "tokens/s @ bs=1 (speculative decoding)", "value": r["tokens_per_s_raw_draft"], "unit": "tokens/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": wl["dtype"],
               "data": "synthetic: random-init weights, random prompt; raw draft (no acceptance knob)",
               "config": {"workload": wl["desc"], "parallelism": r["parallelism"], "tree": "3x4", "prompt_len": args.prompt_len},
               "accept_len": r["accept_len_raw_draft"], "tp": r}
        print(json.dumps(out), flush=True)
    if dist.is_initialized():                                  # also the 1-rank group bench.py opens for the RCCL smoke run
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()
    return out
