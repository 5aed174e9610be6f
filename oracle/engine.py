"""CPU restatement of the two reference speculation engines (oracle; test-only).

Token-id level (no tokenizer): restates
umbrella/speculation/static_speculation_engine.py:48-131,143-210,257-364,414-417
and umbrella/speculation/dynamic_speculation_engine.py:46-88,100-168,215-335.
``draft`` / ``target`` are objects with the reference model-runtime face
(``inference``, ``graph_inference``, ``gather_kv_incremental``, ``clear``,
``kv_cache``) -- normally oracle.model.OracleLlama.
Every iteration appends a dict to ``self.trace`` (what the golden fixtures pin).
"""
from __future__ import annotations

import torch

from . import ops


def causal(n):
    return torch.tril(torch.ones(n, n, dtype=torch.bool))


class _Common:
    def _init_common(self, draft, target, eos_tokens, max_length, safe_buffer, temperature,
                     topp, topk, repetition_penalty):
        self.draft_model, self.target_model = draft, target
        self.eos_tokens = list(eos_tokens)
        self.max_length, self.safe_buffer = max_length, safe_buffer
        self.temperature, self.topp, self.topk = temperature, topp, topk
        self.repetition_penalty = repetition_penalty
        self.storage_ids = torch.arange(max_length)
        self.tokens = torch.zeros(1, max_length, dtype=torch.long)
        self.position_ids = torch.zeros(1, max_length, dtype=torch.long)
        self.num_nodes = 0
        self.trace = []

    def _window(self):
        raise NotImplementedError

    def _feed(self, lo, hi, mask_first_eos):
        """Run draft then target over tokens[lo:hi]; pick the first new token
        (static:165-175 / dynamic:116-133)."""
        sl = slice(lo, hi)
        kw = dict(input_ids=self.tokens[:, sl], storage_ids=self.storage_ids[sl],
                  position_ids=self.position_ids[:, sl], attention_mask=self.mask_iter[sl])
        self.draft_model.inference(**kw)
        logits = self.target_model.inference(**kw)[0]
        if mask_first_eos:
            logits[-1:, self.eos_tokens] = -torch.inf
        self.tokens[:, self.num_nodes] = logits[-1:].argmax(dim=-1)

    def _prefill(self, input_ids):
        P = input_ids.shape[1]
        if P >= self.max_length - 2 * self.safe_buffer:
            return False
        self.num_nodes += P
        self.mask_iter = self._window()
        self.cur = self.num_nodes
        self.tokens[:, :P] = input_ids
        self.position_ids[:, :P] = torch.arange(P)
        self.position_ids[:, P:P + self.tree_size] = P + self.depth
        self._feed(0, P, self.MASK_FIRST_EOS)
        return True

    def _append(self, input_ids):
        A = input_ids.shape[1]
        if A + self.num_nodes >= self.max_length - 2 * self.safe_buffer:
            return False
        self.tokens[:, self.num_nodes + 1:self.num_nodes + 1 + A] = input_ids
        old = self.num_nodes
        self.num_nodes += A + 1
        self.mask_iter = self._window()
        self.cur = self.num_nodes
        self.position_ids[:, :self.num_nodes] = torch.arange(self.num_nodes)
        self.position_ids[:, self.num_nodes:self.num_nodes + self.tree_size] = self.num_nodes + self.depth
        self._feed(old, self.num_nodes, self.MASK_FIRST_EOS)
        return True

    def _sample(self, logits):
        n, T = self.num_nodes, self.cur - self.num_nodes
        if self.repetition_penalty > 1.01:
            logits = ops.repetition_penalty(self.tokens[:, :n + 1].expand(T, -1), logits, self.repetition_penalty)
        if self.temperature < 0.05:
            return logits.argmax(dim=-1)
        return self._stochastic(logits)

    def _commit(self, sampled, tree_mask, want):
        """Accept scan + state update, shared tail of verify()."""
        n = self.num_nodes
        spec = self.tokens[0, n:self.cur].clone()
        path, bonus = ops.accept_scan(sampled, spec, self.parents, tree_mask, want)
        a = path.shape[0]
        self.tokens[0, n:n + a] = spec[path]
        self.tokens[0, n + a] = bonus
        go_on = True
        e = ops.first_eos(self.tokens[0, n:n + a + 1], self.eos_tokens)
        if e >= 0:
            go_on, path = False, path[:e]
            a = len(path)
        abs_path = path + n
        self.draft_model.gather_kv_incremental(abs_path, n)
        self.target_model.gather_kv_incremental(abs_path, n)
        self.num_nodes = n + a
        self.cur = self.num_nodes
        self.mask_iter = self._window()
        if a > 0:
            self.position_ids[:, n:self.num_nodes] = self.position_ids[:, abs_path]
        self.position_ids[:, self.num_nodes:self.num_nodes + self.tree_size] = self.num_nodes + self.depth
        self.trace.append(dict(spec=spec.tolist(), sampled=sampled.tolist(), parents=self.parents.tolist(),
                               accept_path=path.tolist(), accept_length=a, target_token=bonus,
                               num_nodes=self.num_nodes, go_on=go_on))
        return go_on

    def validate_status(self):
        return self.num_nodes <= self.max_length - self.safe_buffer

    def reset(self):
        self.num_nodes = 0
        self.tokens.zero_(); self.position_ids.zero_()
        self.draft_model.clear(); self.target_model.clear()

    def generate_ids(self, input_ids, max_new_tokens):
        """generate() minus tokenizer/timing (static:373-434): returns
        (generated_tokens, avg_accept_tokens)."""
        if len(input_ids) == 0 or max_new_tokens == 0:
            return [], 0
        if not self._prefill(torch.tensor(input_ids, dtype=torch.long)[None]):
            self.reset()
            return [], 0
        steps, go, start = 0, True, self.num_nodes
        while go and (self.num_nodes - start) < max_new_tokens and self.validate_status():
            self.build_tree()
            go = self.verify()
            steps += 1
        out = self.tokens[0, start:self.num_nodes + 1].tolist()
        acc = (self.num_nodes - start + 1) / steps
        self.reset()
        return out, acc


class OracleStaticEngine(_Common):
    """Sequoia growmap engine (static_speculation_engine.py)."""
    MASK_FIRST_EOS = False                                          # static:173 plain argmax

    def __init__(self, draft, target, growmap: dict, eos_tokens, max_length=256, safe_buffer=64,
                 temperature=0.0, topp=0.9, topk=32, repetition_penalty=1.0, uniform_samples=None):
        self._init_common(draft, target, eos_tokens, max_length, safe_buffer, temperature, topp, topk,
                          repetition_penalty)
        L = max_length
        self.level_ids = [torch.tensor(x, dtype=torch.long) for x in growmap["roots"]]
        self.tree_depth = len(self.level_ids)
        self.tree_size = growmap["size"]
        self.branches = growmap["branches"]
        self.tree_mask = torch.tensor(growmap["mask"]) == 1
        self.node_in_path = self.tree_mask.int().sum(dim=-1)
        self.parents = torch.zeros(self.tree_size, dtype=torch.int32)
        for v, succ in enumerate(growmap["Successors"]):
            self.parents[succ] = v
        self.depth = torch.tensor(growmap["depth"], dtype=torch.long)
        # [L, 2L] mask: left half causal, growmap mask pasted at the corner (static:55-57,79)
        self.attn_mask = torch.zeros(L, 2 * L, dtype=torch.bool)
        self.attn_mask[:, :L] = causal(L)
        self.attn_mask[L - self.tree_size:L, L - self.tree_size:L] = self.tree_mask
        # static:115-123 -- children of level i laid out by (parent order, rank)
        self.gather_idx = []
        for i in range(self.tree_depth - 1):
            mb = max(self.branches[i])
            self.gather_idx.append(torch.cat([torch.arange(b) + j * mb for j, b in enumerate(self.branches[i])]))
        self.uniform_samples = uniform_samples

    def _window(self):                                               # static:150
        L, m = self.max_length, self.num_nodes + self.tree_size
        return self.attn_mask[L - m:L, L - m:2 * L - m].contiguous()

    def build_tree(self):                                            # static:257-281
        for step in range(self.tree_depth):
            w = len(self.level_ids[step])
            sl = slice(self.cur, self.cur + w)
            logits = self.draft_model.graph_inference(
                input_ids=self.tokens[:, sl], storage_ids=self.storage_ids[sl],
                position_ids=self.position_ids[:, sl], attention_mask=self.mask_iter[sl])[0]
            self.cur += w
            if step < self.tree_depth - 1:
                new = ops.topk_flatten_gather(logits, max(self.branches[step]), self.gather_idx[step])
                self.tokens[0, self.cur:self.cur + sum(self.branches[step])] = new

    def _stochastic(self, logits):                                   # static:131,310
        """flashinfer.sampling.top_k_top_p_sampling_from_logits(logits / T, uniform_samples, topk, topp) with the ONE
        uniform_samples = rand(3, tree_size) tensor drawn at initialize() and reused by every verify (static:131) --
        the caller passes it in (`uniform_samples`), so a recorded reference run can be replayed draw for draw.
        The sampler itself is the restatement in oracle/ops.py (flashinfer wheel absent: parity unpinned there)."""
        assert self.uniform_samples is not None, "static stochastic verification needs the engine's uniform_samples"
        ids, _ = ops.top_k_top_p_sampling_from_logits(logits / self.temperature, self.uniform_samples, self.topk, self.topp)
        return ids

    def verify(self):                                                # static:282-351
        sl = slice(self.num_nodes, self.cur)
        logits = self.target_model.inference(
            input_ids=self.tokens[:, sl], storage_ids=self.storage_ids[sl],
            position_ids=self.position_ids[:, sl], attention_mask=self.mask_iter[sl])[0]
        return self._commit(self._sample(logits), self.tree_mask, self.node_in_path)


class OracleDynamicEngine(_Common):
    """SpecExec-style beam-grown tree (dynamic_speculation_engine.py)."""
    MASK_FIRST_EOS = True                                            # dynamic:130,163

    def __init__(self, draft, target, eos_tokens, width=16, depth=24, num_beams=24, max_length=256,
                 safe_buffer=64, temperature=0.0, topp=0.9, topk=32, repetition_penalty=1.0, generator=None):
        self._init_common(draft, target, eos_tokens, max_length, safe_buffer, temperature, topp, topk,
                          repetition_penalty)
        L = max_length
        self.tree_width, self.tree_depth, self.num_beams = width, depth, num_beams
        self.tree_size = width * depth + 1
        self.attn_mask = torch.zeros(L, L, dtype=torch.bool)
        c = L - self.tree_size + 1
        self.attn_mask[:c, :c] = causal(c)                          # dynamic:59
        self.tree_score = torch.zeros(self.tree_size)
        self.parents = torch.zeros(self.tree_size, dtype=torch.int32)
        self.depth = torch.tensor([0] + [i + 1 for i in range(depth) for _ in range(width)], dtype=torch.long)
        self.generator = generator

    def _window(self):                                               # dynamic:107
        L, m = self.max_length, self.num_nodes + self.tree_size
        return self.attn_mask[L - m:L, L - m:L].contiguous()

    def build_tree(self):                                            # dynamic:215-248
        n, W, B = self.num_nodes, self.tree_width, self.num_beams
        for step in range(self.tree_depth + 1):
            w = W if step > 0 else 1
            sl = slice(self.cur, self.cur + w)
            logits = self.draft_model.inference(
                input_ids=self.tokens[:, sl], storage_ids=self.storage_ids[sl],
                position_ids=self.position_ids[:, sl], attention_mask=self.mask_iter[sl])[0]
            self.cur += w
            if step == self.tree_depth:
                break
            top, ids = logits.topk(dim=-1, k=B)
            s_new = torch.log(top.softmax(dim=-1) + 1e-4)            # softmax over the B kept logits only
            hist = self.tree_score[self.cur - w - n:self.cur - n]
            score, idx = (hist[:, None] + s_new).reshape(w * B).topk(k=W)
            lo = self.cur - n
            self.tree_score[lo:lo + W] = score
            self.tokens[0, self.cur:self.cur + W] = ids.reshape(w * B)[idx]
            par = idx // B
            self.parents[lo:lo + W] = (par + self.cur - w - n).int()
            self.mask_iter[self.cur:self.cur + W] = self.mask_iter[self.cur - w + par]
            self.mask_iter[self.cur:self.cur + W, self.cur:self.cur + W].fill_diagonal_(True)

    def _stochastic(self, logits):                                   # dynamic:276-281
        logits = ops.keep_topk(logits, self.topk)
        p = ops.top_p_renorm(torch.softmax(logits / self.temperature, dim=-1), self.topp)
        return torch.multinomial(p, num_samples=1, generator=self.generator).squeeze(-1)

    def verify(self):                                                # dynamic:250-327
        n = self.num_nodes
        sl = slice(n, self.cur)
        logits = self.target_model.inference(
            input_ids=self.tokens[:, sl], storage_ids=self.storage_ids[sl],
            position_ids=self.position_ids[:, sl], attention_mask=self.mask_iter[sl])[0]
        sampled = self._sample(logits)
        tree_mask = self.mask_iter[n:self.cur, n:self.cur].clone()
        go = self._commit(sampled, tree_mask, self.depth + 1)
        self.parents.zero_(); self.tree_score.zero_()
        return go
