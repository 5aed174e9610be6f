"""Host-side logic shared by the static (Sequoia) and dynamic (SpecExec) engines.

Keeps the reference engine surface (umbrella/speculation/base.py:9-59 and the
public methods of static_speculation_engine.py / dynamic_speculation_engine.py)
while moving all per-iteration state onto the GPU:

* ``tokens`` (int32), ``num_nodes`` (device scalar) and the tree tables live in
  HBM; every kernel derives positions / KV slots / mask rows from them, so one
  whole iteration -- draft levels, top-k expand, verify forward, arg-max, accept
  scan, KV compaction, state update -- is a single hipGraph replay;
* the host reads back five ints per iteration (one sync instead of the
  reference's ``nonzero()`` + ``.tolist()`` pair, static:320,329).
"""
from __future__ import annotations

import os
import re
import time

import torch

from .. import _lib
from ..logging_config import setup_logger
from ..models import AutoModelLM
from ..utils import TextColors
from .base import BaseEngine
from .speculation_utils import IdTokenizer, is_sentence_complete_regex

logger = setup_logger()


class HipEngine(BaseEngine):
    MASK_FIRST_EOS = False
    DRAFT_KW = {}
    TARGET_KW = {}

    def _common_kwargs(self, kwargs):
        self.max_length = kwargs.pop("max_length", 8192)
        self.stop_distance = kwargs.pop("stop_distance", 32)
        self.safe_buffer = kwargs.pop("safe_buffer", 64)
        self.temperature = kwargs.pop("temperature", 0.0)
        self.topp = kwargs.pop("topp", 0.9)
        self.repetition_penalty = kwargs.pop("repetition_penalty", 1.0)
        self.topk = kwargs.pop("topk", 32)
        self.use_graph = kwargs.pop("hip_graph", True)
        self.graph_scope = "iteration"        # "draft": only the draft tree is captured (layer-streamed targets)
        self.seed = kwargs.pop("seed", 0)
        # optional injected models / tokenizer (tests, synthetic benches)
        self._draft_model = kwargs.pop("draft_model_obj", None)
        self._target_model = kwargs.pop("target_model_obj", None)
        self._tokenizer = kwargs.pop("tokenizer", None)

    # ------------------------------------------------------------------ setup
    def _load_models(self, draft_kw, target_kw):
        cfg = dict(self.config)
        if self._draft_model is None:
            self.draft_model = AutoModelLM.from_pretrained(model_name=self.draft_model_name, batch_size=1,
                                                           max_length=self.max_length, device=self.device,
                                                           dtype=self.dtype, **draft_kw)
            self.draft_model.alloc(**dict(cfg))
        else:
            self.draft_model = self._draft_model
        if self._target_model is None:
            self.target_model = AutoModelLM.from_pretrained(model_name=self.target_model_name, batch_size=1,
                                                            max_length=self.max_length, device=self.device,
                                                            dtype=self.dtype, **target_kw)
            self.target_model.alloc(**dict(cfg))
        else:
            self.target_model = self._target_model
        # roles: the draft may take the <= 4-row GEMV kernels (its logits only steer proposals); the target keeps one
        # kernel path per shape so that a token's logits never depend on the rows sharing its launch
        if self.draft_model is not self.target_model:
            for mdl, on in ((self.target_model, False), (self.draft_model, True)):
                if hasattr(mdl, "use_gemv"):
                    mdl.use_gemv(on)
        self.max_length = self.target_model.max_length
        assert self.draft_model.max_length == self.max_length
        assert self.draft_model.config.vocab_size == self.target_model.config.vocab_size
        self.vocab_size = self.target_model.config.vocab_size
        self.eos_tokens = list(self.target_model.eos_tokens)
        if self._tokenizer is not None:
            self.tokenizer = self._tokenizer
        else:
            try:
                from transformers import AutoTokenizer
                self.tokenizer = AutoTokenizer.from_pretrained(self.target_model_name, local_files_only=True)
            except Exception as e:
                logger.warning(f"NO TOKENIZER for '{self.target_model_name}' ({type(e).__name__}): falling back to the "
                               "id tokenizer -- text is read and written as space-separated token ids")
                self.tokenizer = IdTokenizer()       # no tokenizer files offline: ids <-> "12 7 99" text

    def _alloc_state(self, tree_size, max_path):
        dev = self.device
        self.tree_size, self.max_path = tree_size, max_path
        # one iteration writes KV slots / RoPE positions n .. n + tree_size - 1: the context guard must cover the whole
        # tree, not only `safe_buffer` (the reference fails with a slice-shape error at the same point; here the kernels
        # would write past the caches).  guard == safe_buffer whenever the tree fits in it (all static growmaps).
        self._guard = max(self.safe_buffer, tree_size + 1)
        assert self.max_length > self.safe_buffer + self._guard, (
            f"max_length {self.max_length} cannot hold a {tree_size}-node tree plus safe_buffer {self.safe_buffer}")
        self.tokens = torch.zeros(self.max_length + tree_size + 8, dtype=torch.int32, device=dev)
        self.n_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.sampled = torch.zeros(tree_size, dtype=torch.int32, device=dev)
        self.res = torch.zeros(8, dtype=torch.int32, device=dev)
        self.res_host = torch.zeros(8, dtype=torch.int32).pin_memory()
        self.path = torch.zeros(max(max_path, 1), dtype=torch.int32, device=dev)
        self.eos_dev = torch.tensor(self.eos_tokens or [-1], dtype=torch.int32, device=dev)
        self.num_nodes = 0
        self._graph = None
        # draft root forward with a one-token look-back (see _draft_root)
        self.lookback = os.environ.get("UMB_DRAFT_LOOKBACK", "1") != "0"
        self.root_depth = torch.tensor([-1, 0], dtype=torch.int32, device=dev)
        self.root_mask = torch.zeros(2, self.mask_words, dtype=torch.int64, device=dev)
        self.root_mask[1, 0] = 1
        self.rng_state = torch.tensor([self.seed], dtype=torch.int64, device=dev)    # device-resident: reseed without recapture
        # final workspace sizes before any graph capture: wide prompt chunks, logits only for tree rows
        for mdl, rows in ((self.draft_model, max(self.draft_rows, 2)), (self.target_model, tree_size)):
            rows = max(mdl.CHUNK, rows)
            mdl.reserve(max(mdl.PREFILL_CHUNK, rows), logit_rows=rows)

    # ------------------------------------------------------------------ text API
    def prefill(self, text: str):
        input_ids = self.tokenizer.encode(text, return_tensors="pt").to(self.device)
        return self._prefill(input_ids=input_ids)

    def append(self, text: str):
        input_ids = self.tokenizer.encode(text, return_tensors="pt").to(self.device)
        return self._append(input_ids[:, 1:])          # drop the BOS the tokenizer adds (static:140)

    # ------------------------------------------------------------------ prefix handling
    def _first_token(self, logits_row):
        if self.MASK_FIRST_EOS and self.eos_tokens:                  # dynamic:130,163
            _lib.call("umb_mask_eos", logits_row, self.eos_dev, len(self.eos_tokens))
        first = self.sampled[:1]
        _lib.call("umb_argmax_rows", first, logits_row, 1, self.vocab_size)
        return first

    def _feed(self, lo, hi):
        ids = self.tokens[lo:hi]
        # with the look-back schedule the token before `lo` may still lack draft KV (a deepest-level node accepted in
        # the iteration that ended the previous turn): the draft's prefill starts one slot earlier and re-derives it
        dlo = lo - 1 if (self.lookback and lo > 0) else lo
        self.draft_model.prefill_tokens(self.tokens[dlo:hi], dlo, want_logits=False)
        row = self.target_model.prefill_tokens(ids, lo, want_logits=True)
        first = self._first_token(row)
        self.tokens[hi:hi + 1] = first
        self.num_nodes = hi
        self.n_dev.fill_(hi)
        self.last_bonus = None

    @torch.inference_mode()
    def _prefill(self, input_ids: torch.LongTensor):
        P = input_ids.shape[1]
        if P + self.num_nodes >= self.max_length - self.safe_buffer - self._guard:
            return False
        base = self.num_nodes
        self.tokens[base:base + P] = input_ids[0].to(device=self.device, dtype=torch.int32)
        self._feed(base, base + P)
        return True

    @torch.inference_mode()
    def _append(self, input_ids: torch.LongTensor):
        A = input_ids.shape[1]
        if A + self.num_nodes >= self.max_length - self.safe_buffer - self._guard:
            return False
        n = self.num_nodes
        self.tokens[n + 1:n + 1 + A] = input_ids[0].to(device=self.device, dtype=torch.int32)
        self._feed(n, n + A + 1)        # the pending bonus token becomes context (static:183-185)
        return True

    # ------------------------------------------------------------------ one iteration
    def _draft_root(self):
        """First draft forward of an iteration.  The reference runs one more draft forward per iteration than it
        has levels to expand: the deepest level is fed through the draft only to fill its KV cache, in case one of
        those nodes is accepted (static:257-281 five forwards for four levels, dynamic:218,235).  Here that forward
        is dropped; instead the root forward takes two rows -- the token before the root (slot n-1, the only
        accepted token that can lack draft KV) and the root -- causally.  Re-deriving slot n-1 when it already had
        KV rewrites the same keys (the kernels are batch invariant), and T = 2 costs what T = 1 does in an
        HBM-bound forward: one draft forward less per iteration, same proposals.  Logits row 0 = the root's."""
        d = self.draft_model
        if self.lookback:
            d.forward_tree(self.tokens, self.n_dev, self.root_depth[1:], -1, 2, self.root_mask[1:], self.mask_words,
                           head_from=1)
        else:
            d.forward_tree(self.tokens, self.n_dev, self.depth, 0, 1, self.mask_bits, self.mask_words, head_from=0)

    def _greedy(self):
        return self.temperature < 0.05 and not (self.repetition_penalty > 1.01)

    def _sample(self, dbg=None):
        """One token per tree node from the target logits (static:298-310, dynamic:266-281): arg-max when greedy
        and unpenalised, otherwise umb_sample_rows (penalty -> top-k -> softmax(x/T) -> top-p -> draw), all on
        the device so the stochastic iteration is graph-replayable too."""
        T = self.tree_size
        logits = self.target_model.logits_buffer
        if self._greedy():
            _lib.call("umb_argmax_rows", self.sampled, logits, T, self.vocab_size)
            return
        k = min(int(self.topk), self.vocab_size)
        dk, di, dp = (0, None, None) if dbg is None else (dbg[0].shape[1], dbg[0], dbg[1])
        u = getattr(self, "uniform_samples", None)
        if u is not None:
            # the reference's static-engine draw: ONE [rounds, T] tensor of uniforms reused by every verify (static:131,310)
            _lib.call("umb_sample_rows_uniform", self.sampled, logits, T, self.vocab_size, self.tokens, self.n_dev,
                      float(self.repetition_penalty), float(self.temperature), k, float(self.topp), u, u.shape[0],
                      u.shape[1], dk, di, dp)
            return
        _lib.call("umb_sample_rows", self.sampled, logits, T, self.vocab_size, self.tokens, self.n_dev,
                  float(self.repetition_penalty), float(self.temperature), k, float(self.topp), self.rng_state,
                  dk, di, dp)

    def _commit(self):
        _lib.call("umb_accept_scan", self.sampled, self.parents, self.tokens, self.n_dev, self.tree_size,
                  self.eos_dev, len(self.eos_tokens), self.res, self.path)
        self.draft_model.kv_cache.compact(self.res, self.path, self.max_path)
        self.target_model.kv_cache.compact(self.res, self.path, self.max_path)
        st = getattr(self.draft_model, "chain_status_word", None)
        if st is not None:                    # the persistent chain's give-up word rides along with the accept result
            self.res[7:8].copy_(st)
        pw = getattr(self.target_model, "peer_status_word", None)
        if pw is not None:                    # and the tensor-parallel peer path's (csrc/tp.hip)
            self.res[6:7].copy_(pw)
        self.res_host.copy_(self.res, non_blocking=True)

    def _iteration_launch(self):
        self.build_tree()
        if self.graph_scope == "draft":
            return
        self._verify_forward()
        self._sample()
        self._commit()

    def _iteration_tail(self):
        """What runs eagerly after a draft-only graph: the streamed verify forward (event-ordered copies on the side
        stream cannot live inside the graph), sampling and the commit."""
        self._verify_forward()
        self._sample()
        self._commit()

    def _capture(self):
        """Capture one full iteration into a hipGraph (replayed by step()).  Sampling knobs are launch arguments,
        hence frozen into the graph: update_generation_args drops the graph when they change."""
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream())
        n_save = self.n_dev.clone()
        tok_save = self.tokens.clone()
        with torch.cuda.stream(s):
            self._iteration_launch()          # warm-up outside capture
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._iteration_launch()
        torch.cuda.synchronize()
        self.n_dev.copy_(n_save)
        self.tokens.copy_(tok_save)
        self._graph = g

    @torch.inference_mode()
    def step(self) -> bool:
        """build_tree + verify as one launch; returns continue_generation."""
        if self.num_nodes + self.tree_size > self.max_length:
            raise RuntimeError(f"speculation tree of {self.tree_size} nodes at position {self.num_nodes} exceeds "
                               f"max_length {self.max_length}: check validate_status() before step()")
        if getattr(self, "enable_override", False):
            self._fill_override()
        if self.use_graph:
            if self._graph is None:
                self._capture()
            self._graph.replay()
        else:
            self._iteration_launch()
        if self.graph_scope == "draft":
            self._iteration_tail()
        return self._finish_iteration()

    def _finish_iteration(self) -> bool:
        torch.cuda.current_stream().synchronize()
        keep, bonus, eos, n_new, raw = self.res_host[:5].tolist()
        if self.res_host[6] != 0:
            st = int(self.res_host[6]) & 0xffffffff
            self.target_model.peer.reset_status()
            raise RuntimeError(f"tensor parallel: a peer never published its tile (status {st:#x}); the tokens of this "
                               "iteration are invalid -- the group is out of step or a rank died")
        chain_gave_up = int(self.res_host[7]) & 0xffffffff
        n_old = self.num_nodes
        self.last_accept, self.last_bonus = keep, bonus
        self.num_nodes = n_new
        self.draft_model.kv_cache.kv_offset = n_new
        self.target_model.kv_cache.kv_offset = n_new
        if chain_gave_up:
            self._chain_fallback(chain_gave_up, n_old, n_new)
        return eos == 0

    def _chain_fallback(self, status, n_old, n_new):
        """The draft's persistent chain gave up on a hand-off during this iteration (its launches need every workgroup
        resident, i.e. the whole device).  The iteration's TOKENS are still right: a draft only proposes, the accept scan
        keeps a proposal only where it equals the target's own sample, and the target never read anything the chain wrote.
        What may be wrong is the draft's KV of the rows it processed.  So: take the chain out (the five GEMV launches are
        bit-identical), drop the graph that has it captured, re-derive the draft KV of the committed rows, carry on."""
        d = self.draft_model
        if not getattr(self, "_chain_warned", False):
            logger.warning(f"the draft model's persistent chain gave up on a hand-off (status {status:#x}: its launches need the "
                           "whole device, one process per GPU); continuing on the five GEMV launches (UMB_CHAIN=0 selects them "
                           "from the start)")
            self._chain_warned = True
        d.disable_chain()
        self._graph = None
        self.res[7:8].zero_()
        lo = max(n_old - 1, 0)
        if n_new > lo:
            d.prefill_tokens(self.tokens[lo:n_new], lo, want_logits=False)
            d.kv_cache.kv_offset = n_new

    @torch.inference_mode()
    def verify(self):
        self._verify_forward()
        self._sample()
        self._commit()
        return self._finish_iteration()

    # ------------------------------------------------------------------ generation loops
    def validate_status(self):
        return self.num_nodes <= (self.max_length - self._guard)

    def update_generation_args(self, **generation_args):
        before = (self.temperature, self.topp, self.repetition_penalty, self.topk)
        self.temperature = generation_args.pop("temperature", self.temperature)
        self.topp = generation_args.pop("topp", self.topp)
        self.repetition_penalty = generation_args.pop("repetition_penalty", self.repetition_penalty)
        self.topk = generation_args.pop("topk", self.topk)
        if before != (self.temperature, self.topp, self.repetition_penalty, self.topk):
            self._graph = None

    def manual_seed(self, seed: int):
        self.seed = seed
        self.rng_state.fill_(seed)
        u = getattr(self, "uniform_samples", None)
        if u is not None and getattr(self, "_uniform_arg", None) is None:
            # the static engine's frozen uniforms are a function of the seed (drawn where the reference draws them,
            # static:131): a new seed redraws them IN PLACE, so a captured iteration keeps replaying
            u.copy_(torch.rand(u.shape[0], u.shape[1], generator=torch.Generator().manual_seed(int(seed))))

    @torch.inference_mode()
    def reset(self):
        self.num_nodes = 0
        self.n_dev.zero_()
        self.tokens.zero_()
        self.draft_model.clear()
        self.target_model.clear()

    def _decode_words(self, ids):
        return (self.tokenizer.decode(ids, skip_special_tokens=True, clean_up_tokenization_spaces=False,
                                      spaces_between_special_tokens=False).strip().split(" "))

    def _stop_now(self, words, start, max_new_tokens):
        done = self.num_nodes - start
        return (is_sentence_complete_regex(words[-1]) and done >= max_new_tokens - self.stop_distance) or done >= max_new_tokens

    @torch.inference_mode()
    def speculative_decoding(self, max_new_tokens=128):
        max_new_tokens = max(max_new_tokens, self.stop_distance)
        torch.cuda.synchronize()
        t1 = time.time()
        steps, decode, start, ids, pos, words = 0, True, self.num_nodes, [], 0, [""]
        while decode and self.validate_status():
            begin = self.num_nodes
            decode = self.step()
            steps += 1
            ids.extend(self.tokens[begin:self.num_nodes].tolist())
            words = self._decode_words(ids)
            now = len(words) - 1
            if now > pos:
                print(" ".join(words[pos:now]), end=" ", flush=True)
                pos = now
            if self._stop_now(words, start, max_new_tokens):
                decode = False
        print(" ".join(words[pos:]), flush=True)
        torch.cuda.synchronize()
        t2 = time.time()
        dec_len = self.num_nodes - start + 1
        logger.info(TextColors.colorize("Avg Accept Tokens {:.2f} | TPOT {:.2f} ms ".format(
            dec_len / max(steps, 1), 1000 * (t2 - t1) / dec_len), "magenta"))
        return dec_len, (t2 - t1), steps

    def _empty(self, api_args):
        api_args.update(generated_text="", generated_tokens=[], avg_accept_tokens=0, time_per_output_token=0)
        return api_args

    def _start_request(self, api_args):
        self.update_generation_args(**api_args)
        input_ids = api_args.get("input_ids", None)
        max_new_tokens = api_args.get("max_new_tokens", 128)
        if input_ids is None:
            context = api_args.get("context", None)
            if context is None or len(context) == 0 or max_new_tokens == 0:
                return None
            return self.prefill(context)
        if len(input_ids) == 0 or max_new_tokens == 0:
            return None
        return self._prefill(input_ids=torch.tensor(list(input_ids), dtype=torch.long)[None])

    @torch.inference_mode()
    def generate(self, **api_args):
        ok = self._start_request(api_args)
        if not ok:
            if ok is False:
                self.reset()
            return self._empty(api_args)
        max_new_tokens = api_args.get("max_new_tokens", 128)
        torch.cuda.synchronize()
        t1 = time.time()
        steps, decode, start = 0, True, self.num_nodes
        while decode and (self.num_nodes - start) < max_new_tokens and self.validate_status():
            decode = self.step()
            steps += 1
        torch.cuda.synchronize()
        t2 = time.time()
        dec_len = self.num_nodes - start + 1
        toks = self.tokens[start:self.num_nodes + 1].tolist()
        api_args["generated_text"] = self.tokenizer.decode(toks, skip_special_tokens=True, clean_up_tokenization_spaces=False)
        api_args["generated_tokens"] = toks
        api_args["avg_accept_tokens"] = dec_len / max(steps, 1)
        api_args["time_per_output_token"] = 1000 * (t2 - t1) / dec_len
        self.reset()
        return api_args

    @torch.inference_mode()
    def generate_stream(self, **api_args):
        ok = self._start_request(api_args)
        if ok is None:
            self._empty(api_args)
            return
        if not ok:
            yield "Exceeding reserved allowed context length", "Exceeding reserved allowed context length"
            self.reset()
            return
        max_new_tokens = max(api_args.get("max_new_tokens", 128), self.stop_distance)
        torch.cuda.synchronize()
        t1 = time.time()
        steps, decode, start, ids, pos, text, words = 0, True, self.num_nodes, [], 0, "", [""]

        def perf():
            d = self.num_nodes - start + 1
            return "Output Tokens {} | Avg Accept Tokens {:.2f} | TPOT {:.2f} ms ".format(
                d, d / max(steps, 1), 1000 * (time.time() - t1) / d)

        while decode and self.validate_status():
            begin = self.num_nodes
            decode = self.step()
            steps += 1
            ids.extend(self.tokens[begin:self.num_nodes].tolist())
            words = self._decode_words(ids)
            now = len(words) - 1
            if now > pos:
                text += " ".join(words[pos:now]) + " "
                yield text, perf()
                pos = now
            if self._stop_now(words, start, max_new_tokens):
                decode = False
        tail = " ".join(words[pos:])
        if tail:
            text += tail
        yield text, perf()
        torch.cuda.synchronize()
        logger.info(TextColors.colorize(perf(), "magenta"))
        self.reset()
