"""Llama model runtime on the HIP hot path.

Mirrors the reference's model-runtime face (umbrella/models/base.py:4-32,
umbrella/models/llama.py): ``alloc(**kw)``, ``inference(input_ids, position_ids,
attention_mask, storage_ids) -> fp32 logits [1,T,V]``, ``graph_inference``,
``gather_kv_incremental(indices, offset)``, ``clear()``, attrs ``config`` /
``kv_cache``.  One class covers the reference's five Llama variants:

  Llama / LlamaAwq        resident weights (dense 16-bit or AWQ int4 tiles)
  LlamaOffload / AwqOffload  ``offload=True``: layer slabs in pinned host DRAM streamed through two
                          device slabs on a side stream (event ordered, one hipMemcpyAsync per layer)
  LlamaCudagraph          ``cuda_graph=True``: honours ``exit_layer``; step graphs are captured by the engine

Everything numeric runs in libumbrella_hip.so (see include/umbrella_hip.h).
"""
from __future__ import annotations

import ctypes as C
import math
import os

import torch

from .. import _lib
from .._lib import UmbLayer, UmbLinear, UmbModel, UmbOffload, UmbStep, UmbWorkspace
from ..attn.cache import TreeKVCache
from .base import LLMBase
from .config import KNOWN, LlamaCfg, rope_tables
from .synthetic import LINEARS, linear_shapes, synth_awq_tensors, synth_tensor
from ..logging_config import setup_logger

logger = setup_logger()


def _resolve_hub_snapshot(model_name: str):
    """hub id -> local snapshot directory from the Hugging Face cache (no network), or None."""
    try:
        from huggingface_hub import snapshot_download
        path = snapshot_download(model_name, local_files_only=True)
        return path if os.path.exists(os.path.join(path, "config.json")) else None
    except Exception:
        return None


def _align(n, a=256):
    return (n + a - 1) // a * a


CHAIN_TMAX = 4        # include/umbrella_hip.h UMB_CHAIN_TMAX: the persistent chain's exchange layout is fixed at 4 rows

class PackedLinear:
    """One linear layer in MFMA tile order (dense 16-bit or AWQ int4)."""

    def __init__(self, N, K, awq, w, meta, force_s1=False):
        self.N, self.K, self.awq, self.w, self.meta = N, K, int(awq), w, meta
        R, S, tb, srow = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        _lib.load().umb_gemm_plan2(N, K, self.awq, int(force_s1), C.byref(R), C.byref(S), C.byref(tb), C.byref(srow))
        self.R, self.S, self.tb, self.S_row = R.value, S.value, tb.value, srow.value

    @property
    def Rtb(self) -> int:
        """the R argument of umb_gemm / umb_gemm_fused: n-tiles per wave | n-tiles per block << 8"""
        return self.R | (self.tb << 8)

    @staticmethod
    def packed_bytes(N, K, awq):
        if awq:
            return _align(N * K // 2), _align((N // 16) * (K // 128) * 64)
        return _align(N * K * 2), 0

    @classmethod
    def from_dense(cls, w: torch.Tensor, out=None, force_s1=False, interleave=False, rope=None):
        """w [N, K] fp16/bf16 on the GPU (HF row-major).  interleave: w is a fused [gate; up] stack whose
        rows are stored as (gate_m, up_m) pairs (SiLU*up becomes the GEMM epilogue; implies S == 1).
        rope=(D, n_rope_heads): w is a fused [q|k|v] stack; inside each q/k head rows are stored as RoPE
        partner pairs (m, m + D/2) so rotate-half is lane-local in the GEMM epilogue."""
        N, K = w.shape
        assert N % 16 == 0 and K % 128 == 0, (N, K)
        w = w.contiguous()
        buf = out if out is not None else torch.empty(N * K * 2, dtype=torch.uint8, device=w.device)
        mode, D, rh = (1, 0, 0) if interleave else ((2, rope[0], rope[1]) if rope else (0, 0, 0))
        _lib.call("umb_repack_dense", buf, w, N, K, mode, D, rh, _lib.dtype_code(w.dtype))
        lin = cls(N, K, False, buf, None, force_s1 or interleave)
        lin.interleaved, lin.rope = bool(interleave), rope
        return lin

    @classmethod
    def from_awq(cls, qweight, qzeros, scales, group=128, out_w=None, out_meta=None, interleave=False, rope=None):
        """AutoAWQ GEMM tensors on the GPU: qweight [K, N/8] i32, qzeros [K/G, N/8] i32, scales [K/G, N] fp16."""
        K, N = qweight.shape[0], qweight.shape[1] * 8
        assert N % 16 == 0 and K % 128 == 0 and group == 128, (N, K, group)
        wb, mb = cls.packed_bytes(N, K, True)
        w = out_w if out_w is not None else torch.empty(wb, dtype=torch.uint8, device=qweight.device)
        meta = out_meta if out_meta is not None else torch.empty(mb, dtype=torch.uint8, device=qweight.device)
        _lib.call("umb_awq_repack", w, meta, qweight.contiguous(), qzeros.contiguous(),
                  scales.to(torch.float16).contiguous(), N, K, group, *((1, 0, 0) if interleave else
                                                                         ((2, rope[0], rope[1]) if rope else (0, 0, 0))))
        lin = cls(N, K, True, w, meta, force_s1=interleave)
        lin.interleaved, lin.rope = bool(interleave), rope
        return lin

    def struct(self, w_ptr=None, meta_ptr=None) -> UmbLinear:
        s = UmbLinear()
        s.w = self.w.data_ptr() if w_ptr is None else w_ptr
        s.meta = (self.meta.data_ptr() if self.meta is not None else 0) if meta_ptr is None else meta_ptr
        s.N, s.K, s.awq, s.R, s.S, s.tb, s.S_row = self.N, self.K, self.awq, self.R, self.S, self.tb, self.S_row
        # row-major copy for the GEMV family (resident dense layers only).  The pointer is published only while the model
        # is flagged as a DRAFT (Llama.use_gemv): GEMV and MFMA kernels sum in different orders, so a model whose
        # T = 1 and T = 13 forwards must agree bit for bit (every target: greedy spec == greedy AR) never takes them.
        rows = getattr(self, "w_rows", None)
        s.w_rows = rows.data_ptr() if (rows is not None and w_ptr is None and getattr(self, "gemv_on", False)) else 0
        return s

    def apply_silu(self, x: torch.Tensor) -> torch.Tensor:
        """interleaved [gate; up] linear: x [T, K] -> act [T, N/2] = SiLU(gate) * up in x.dtype."""
        assert getattr(self, "interleaved", False) and self.S == 1
        T = x.shape[0]
        act = torch.empty(T, self.N // 2, dtype=x.dtype, device=x.device)
        _lib.call("umb_gemm", act, x, x.stride(0), self.w, self.meta, T, self.N, self.K, self.awq, 1, self.Rtb, 2,
                  _lib.dtype_code(x.dtype))
        return act

    def apply_ll(self, x: torch.Tensor, round_out=False, fx=None, epi=0, out=None):
        """Low-latency path (T <= 64): x [T, K] 16-bit row-major -> fp32 [T, N] (epi 0), whole-K workgroups."""
        T = x.shape[0]
        if out is None and epi == 0:
            out = torch.empty(T, self.N, dtype=torch.float32, device=x.device)
        if fx is None:
            fx = _lib.UmbGemmLL()
        fx.round_out = int(round_out)
        _lib.call("umb_gemm_ll", out, to_fm(x), self.w, self.meta, T, self.N, self.K, self.awq, epi, fx,
                  _lib.dtype_code(x.dtype))
        return out

    def apply(self, x: torch.Tensor, round_out=False) -> torch.Tensor:
        """x [T, K] 16-bit -> fp32 [T, N] (split-K partials summed in split order)."""
        T = x.shape[0]
        part = torch.empty(self.S, T, self.N, dtype=torch.float32, device=x.device)
        _lib.call("umb_gemm", part, x, x.stride(0), self.w, self.meta, T, self.N, self.K, self.awq, self.S, self.Rtb,
                  int(round_out), _lib.dtype_code(x.dtype))
        out = part[0]
        for s in range(1, self.S):
            out = out + part[s]
        return out


def to_fm(x: torch.Tensor) -> torch.Tensor:
    """row-major [T, K] 16-bit -> FM layout (MFMA B-fragment order, csrc/lowlat.hip); T <= 64, K % 32 == 0."""
    T, K = x.shape
    tt = _lib.load().umb_ll_token_tiles(T)
    out = torch.zeros(tt * 16 * K, dtype=x.dtype, device=x.device)
    _lib.call("umb_to_fm", out, x.contiguous(), T, K, _lib.dtype_code(x.dtype))
    return out


def from_fm(x_fm: torch.Tensor, T: int, K: int) -> torch.Tensor:
    out = torch.empty(T, K, dtype=x_fm.dtype, device=x_fm.device)
    _lib.call("umb_from_fm", out, x_fm, T, K, _lib.dtype_code(x_fm.dtype))
    return out


def ll_plan(N: int, K: int, awq: bool):
    """(R, WN, WK, NW) of the low-latency GEMM for a [N, K] linear (shape-only)."""
    vals = [C.c_int(0) for _ in range(4)]
    _lib.load().umb_ll_plan(N, K, int(awq), *[C.byref(v) for v in vals])
    return tuple(v.value for v in vals)


def pack_mask_bits(mask: torch.Tensor) -> torch.Tensor:
    """bool [T, C] -> int64 [T, ceil(C/64)] little-endian bit words (bit b of word w = column 64w+b)."""
    T, Cn = mask.shape
    W = (Cn + 63) // 64
    m = torch.zeros(T, W * 64, dtype=torch.int64, device=mask.device)
    m[:, :Cn] = mask.to(torch.int64)
    sh = torch.arange(64, device=mask.device, dtype=torch.int64)
    return (m.view(T, W, 64) << sh).sum(dim=-1)     # wraps into the sign bit as intended


class Llama(LLMBase):
    CHUNK = 64          # rows of the default workspace (tree / generic forwards)
    PREFILL_CHUNK = 1024  # prompt tokens per forward when the workspace allows: the matrix-bound verify GEMM streams the
                          # weights once per chunk (70B-AWQ, 2048-token prompt: 4.2 k tok/s at 128, 6.1 k at 256, 6.7 k at
                          # 512, 7.0 k at 1024; 1B: 53 k / 94 k / 155 k / 211 k).  1024 = rows of the attention counters.

    def __init__(self, model_name: str, batch_size: int = 1, max_length: int = 256, device: str = "cuda:0",
                 dtype=torch.float16, offload: bool = False, cuda_graph: bool = False, state_dict=None,
                 config: LlamaCfg | None = None, seed: int = 0, sched: str | None = None) -> None:
        super().__init__()
        assert batch_size == 1, "the hot path is batch 1 (README.md:22)"
        self.model_name, self.batch_size, self.device, self.dtype = model_name, batch_size, device, dtype
        self.max_length = (max_length + 31) // 32 * 32
        self.offload, self.cuda_graph = offload, cuda_graph
        self._state, self._seed = state_dict, seed
        # layer schedule: False = 8 launches / layer with kernel-boundary split-K reduces (fastest measured),
        # True = 5 launches / layer with in-kernel last-arriver reduces (see csrc/model.hip)
        self.fused = os.environ.get("UMB_FUSED", "0") == "1"
        # sched "ll": low-latency schedule for forwards of <= 64 rows -- 5 launches / layer, whole-K workgroups with the
        # layer's elementwise work as GEMM epilogues, activations in MFMA fragment order (csrc/lowlat.hip)
        # "split": 8 launches / layer, split-K GEMMs on the LDS-shared kernel + reduce kernels.  Default "auto": int4 (AWQ)
        # checkpoints and models of hidden size >= 4096 take "split" (70B-AWQ tree verify 2.30 vs 2.58 ms per 16 layers,
        # 8B bf16 T = 31 forward 4.51 vs 5.09 ms, since the shared kernel's weight ring stopped draining -- DESIGN.md);
        # small dense models, where launches dominate, "ll" (1B draft forward 0.77 vs 0.84 ms)
        self.sched = sched or os.environ.get("UMB_SCHED", "auto")
        self._tp = None                         # UmbTP descriptor: set by tensor_parallel.TensorParallelLlama on a shard
        if config is None and not os.path.isdir(model_name) and state_dict is None:
            local = _resolve_hub_snapshot(model_name)             # HF cache, offline
            if local is not None:
                self.model_name = model_name = local
        if config is not None:
            self.config = config
        elif os.path.isdir(model_name):
            self.config = LlamaCfg.from_dir(model_name)
        elif model_name in KNOWN:
            self.config = KNOWN[model_name]
        else:
            raise ValueError(f"Model type '{model_name}' is not supported. Supported types: {list(KNOWN.keys())} "
                             "or a local directory with config.json")
        if self.sched == "auto":
            self.sched = "split" if (self.config.awq or self.config.hidden_size >= 4096) else "ll"
        c = self.config
        if c.attention_bias:
            self.fused = False                    # projection bias: default / low-latency schedules only
        self.hidden_size, self.num_heads, self.head_dim = c.hidden_size, c.num_attention_heads, c.head_dim
        self.num_key_value_heads = c.num_key_value_heads
        self.eos_tokens = list(c.eos_token_id)
        self.ws_tokens = 0
        self.logit_rows = 0

    # ------------------------------------------------------------------ weights
    def _tensor_source(self):
        """name -> device tensor fetcher: given state dict, local safetensors, or seeded synthetic."""
        c, dev = self.config, self.device
        if self._state is not None:
            sd = self._state
            return lambda name, shape, kind: sd[name].to(dev)
        if os.path.isdir(self.model_name):
            from safetensors import safe_open
            files = [os.path.join(self.model_name, f) for f in sorted(os.listdir(self.model_name)) if f.endswith(".safetensors")]
            index = {}
            for f in files:
                with safe_open(f, "pt") as h:
                    for k in h.keys():
                        index[k] = f

            def fetch(name, shape, kind):
                if name == "lm_head.weight" and name not in index:
                    name = "model.embed_tokens.weight"
                with safe_open(index[name], "pt") as h:
                    return h.get_tensor(name).to(dev)
            return fetch
        if os.environ.get("UMBRELLA_SYNTHETIC", "0") != "1":
            raise FileNotFoundError(
                f"no checkpoint for '{self.model_name}': not a local directory and not in the Hugging Face cache "
                "(HF_HOME / HF_HUB_CACHE).  Set UMBRELLA_SYNTHETIC=1 to run on seeded random weights of the same "
                "shapes (benchmarks / tests only -- the output is meaningless text).")
        logger.warning(f"SYNTHETIC WEIGHTS: '{self.model_name}' is initialised with seeded random tensors "
                       "(UMBRELLA_SYNTHETIC=1); generated tokens carry no meaning")
        gen = torch.Generator(device=dev).manual_seed(self._seed)
        # GPT-2 style scaled init: projections that write into the residual stream get std / sqrt(2L),
        # which keeps a random-init 80-layer stack from chaotically amplifying 1-ulp differences
        resid_scale = 1.0 / math.sqrt(2.0 * c.num_hidden_layers)

        def synth(name, shape, kind):
            if kind == "norm":
                return torch.ones(shape, dtype=self.dtype, device=dev)
            if kind == "bias":
                return synth_tensor(shape, 0.1, self.dtype, dev, gen)
            if kind == "embed":
                return synth_tensor(shape, 1.0, self.dtype, dev, gen)
            std = 0.02 if kind == "linear" else 0.05
            if name.endswith("o_proj.weight") or name.endswith("down_proj.weight"):
                std *= resid_scale
            return synth_tensor(shape, std, self.dtype, dev, gen)
        synth.gen, synth.resid_scale = gen, resid_scale
        return synth

    def _load_linear_group(self, fetch, prefix, names, slab, cursor):
        """Fuse `names` (HF linear names) along N, pack into `slab` at `cursor`; returns (PackedLinear, cursor)."""
        c = self.config
        shapes = linear_shapes(c)
        N = sum(shapes[n][0] for n in names)
        K = shapes[names[0]][1]
        il = names[0] == "mlp.gate_proj"          # fused [gate; up]: interleave rows, SiLU*up in the GEMM epilogue
        rope = (c.head_dim, c.num_attention_heads + c.num_key_value_heads) if names[0] == "self_attn.q_proj" else None
        wb, mb = PackedLinear.packed_bytes(N, K, c.awq)
        w_view = slab[cursor:cursor + (N * K // 2 if c.awq else N * K * 2)]
        meta_view = slab[cursor + wb:cursor + wb + (N // 16) * (K // 128) * 64] if c.awq else None
        if c.awq:
            parts = []
            for n in names:
                base = prefix + n
                if getattr(fetch, "gen", None) is not None:
                    std = 0.02 * (fetch.resid_scale if n in ("self_attn.o_proj", "mlp.down_proj") else 1.0)
                    parts.append(synth_awq_tensors(shapes[n][0], K, c.awq_group, self.device, fetch.gen, std))
                else:
                    parts.append((fetch(base + ".qweight", None, "q"), fetch(base + ".qzeros", None, "q"),
                                  fetch(base + ".scales", None, "q")))
            qw = torch.cat([p[0] for p in parts], dim=1)
            qz = torch.cat([p[1] for p in parts], dim=1)
            sc = torch.cat([p[2] for p in parts], dim=1)
            lin = PackedLinear.from_awq(qw, qz, sc, c.awq_group, out_w=w_view, out_meta=meta_view, interleave=il, rope=rope)
        else:
            w = torch.cat([fetch(prefix + n + ".weight", shapes[n], "linear").to(self.dtype) for n in names], dim=0)
            lin = PackedLinear.from_dense(w, out=w_view, interleave=il, rope=rope)
            # forwards of <= 4 rows (the draft's tree levels) run on the row-streaming GEMV kernels (csrc/gemv.hip), which
            # read a plain row-major copy of the weights with the packed layouts' row order.  Low-latency models only
            # (small dense models: +1x their weight bytes), K in {2048, 8192}, resident layers; UMB_GEMV=0 disables.
            if (self.sched == "ll" and not self.fused and not self.offload and K in (2048, 8192)
                    and os.environ.get("UMB_GEMV", "1") != "0"):
                lin.w_rows = torch.empty_like(w)
                _lib.call("umb_repack_rows", lin.w_rows, w.contiguous(), N, K, 1 if il else (2 if rope else 0),
                          rope[0] if rope else 0, rope[1] if rope else 0)
        if self.fused:
            lin.R, lin.tb, lin.S_row = 1, 0, 0      # the in-kernel split epilogues own one n-tile per wave (ws.fused == 1
                                                   # whenever self.fused is set, whatever `sched` says: see reserve())
        lin.off_w, lin.off_meta = cursor, (cursor + wb if c.awq else None)
        return lin, cursor + wb + mb

    def alloc(self, **kwargs):
        c, dev, dt = self.config, self.device, self.dtype
        _lib.load()
        exit_layer = kwargs.pop("exit_layer", -1)
        self.num_cache_layers = kwargs.get("num_cache_layers", 0)
        Lfull = c.num_hidden_layers
        if self.cuda_graph and exit_layer and exit_layer > 0:          # llama.py:421,450-451
            Lfull = min(Lfull, exit_layer)
        # layer_range=(lo, hi): this process holds one pipeline stage (layers lo..hi-1 of the model)
        lo, hi = kwargs.pop("layer_range", None) or (0, Lfull)
        self.layer_lo, self.layer_hi, self.is_first, self.is_last = lo, hi, lo == 0, hi == Lfull
        L = hi - lo
        self.num_layers = L
        fetch = self._tensor_source()
        reseed = getattr(fetch, "gen", None)
        H, V = c.hidden_size, c.vocab_size
        if reseed is not None:
            reseed.manual_seed(self._seed * 1000003)
        # checkpoints may carry more rows than config.vocab_size (Qwen2.5 7B+: 152064 rows, vocabulary 151936): the
        # logits buffer, arg-max, top-k and sampling all work on V columns, so both matrices are cut to V rows
        Ve = getattr(c, "embed_rows", 0) or V      # tensor-parallel shard: whole table, V / P head rows
        self.embed_tokens = fetch("model.embed_tokens.weight", (Ve, H), "embed")[:Ve].to(dt).contiguous() \
            if (self.is_first or (self.is_last and c.tie_word_embeddings)) else None
        if self.is_last:
            if reseed is not None:
                # synthetic weights: the head has a stream of its own.  Drawn from the running stream it came AFTER the
                # embedding on a single GPU but FIRST on the last stage of a layer-sharded target (which holds no
                # embedding) -- an untied last stage then got the embedding's values as its head, i.e. another model
                # than the one GPU runs (found with the tiny untied pair: the sharded run repeated its input token)
                reseed.manual_seed(self._seed * 1000003 + 500009)
            head_w = self.embed_tokens if c.tie_word_embeddings else fetch("lm_head.weight", (V, H), "head")[:V].to(dt)
            assert head_w.shape == (V, H), (tuple(head_w.shape), V, H)
            self.lm_head = PackedLinear.from_dense(head_w, force_s1=True)
            assert self.lm_head.N == V
            # a tied head's rows exist already as a plain row-major table: the <= 4-row forwards of a DRAFT stream them through
            # the engine of csrc/chain.hip (umb_head_stream; published by use_gemv like the layers' row copies)
            if c.tie_word_embeddings and self.embed_tokens is not None and tuple(self.embed_tokens.shape) == (V, H):
                self.lm_head.w_rows = self.embed_tokens
            del head_w
            self.norm_weight = fetch("model.norm.weight", (H,), "norm").to(dt).contiguous()
        else:
            self.lm_head, self.norm_weight = None, None
        self.cos_cache, self.sin_cache = (t.to(dev).contiguous() for t in rope_tables(c, self.max_length, dt))
        self.kv_cache = TreeKVCache(L, c.num_key_value_heads, c.head_dim, self.max_length, dev, dt)

        groups = (("qkv", ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj")), ("o", ("self_attn.o_proj",)),
                  ("gu", ("mlp.gate_proj", "mlp.up_proj")), ("down", ("mlp.down_proj",)))
        shapes = linear_shapes(c)
        slab_bytes = 0
        for _, names in groups:
            wb, mb = PackedLinear.packed_bytes(sum(shapes[n][0] for n in names), shapes[names[0]][1], c.awq)
            slab_bytes += wb + mb
        self.slab_bytes = slab_bytes
        # host side of the offload path: ONE arena on the GPU's NUMA node, pinned with one registration (models/host_arena.py;
        # a torch pin_memory() tensor per layer when the arena cannot be had)
        self._host_arena = None
        if self.offload and L > self.num_cache_layers:
            from . import host_arena
            slot = (slab_bytes + host_arena.ALIGN - 1) // host_arena.ALIGN * host_arena.ALIGN
            self._host_arena = host_arena.take((L - self.num_cache_layers) * slot, dev)
        self.layers, self.slabs, self.host_slabs, self.norms, self.qkv_biases = [], [], [], [], []
        self._layer_structs = (UmbLayer * L)()
        stream_any = False
        for i in range(L):
            p = f"model.layers.{lo + i}."
            if reseed is not None:                   # per-layer seed: a stage's weights do not depend on the split
                reseed.manual_seed(self._seed * 1000003 + lo + i + 1)
            slab = torch.empty(slab_bytes, dtype=torch.uint8, device=dev)
            cursor, lins = 0, {}
            for key, names in groups:
                lins[key], cursor = self._load_linear_group(fetch, p, names, slab, cursor)
            n1 = fetch(p + "input_layernorm.weight", (H,), "norm").to(dt).contiguous()
            n2 = fetch(p + "post_attention_layernorm.weight", (H,), "norm").to(dt).contiguous()
            self.norms.append((n1, n2))
            qb = None
            if c.attention_bias:
                qb = torch.cat([fetch(p + f"self_attn.{n}.bias", (shapes[f"self_attn.{n}"][0],), "bias").to(dt).reshape(-1)
                                for n in ("q_proj", "k_proj", "v_proj")]).contiguous()
            self.qkv_biases.append(qb)
            streamed = self.offload and i >= self.num_cache_layers
            ls = self._layer_structs[i]
            for key in ("qkv", "o", "gu", "down"):
                ln = lins[key]
                if streamed:      # offsets (+1 so 0 stays NULL) relative to the streamed slab base
                    setattr(ls, key, ln.struct(w_ptr=ln.off_w + 1, meta_ptr=(ln.off_meta + 1) if ln.off_meta is not None else 0))
                else:
                    setattr(ls, key, ln.struct())
            ls.norm1, ls.norm2 = n1.data_ptr(), n2.data_ptr()
            ls.qkv_bias = qb.data_ptr() if qb is not None else None
            if streamed:
                host = self._host_arena.alloc(slab_bytes) if self._host_arena is not None else None
                if host is None:
                    host = torch.empty(slab_bytes, dtype=torch.uint8, pin_memory=True)
                host.copy_(slab)
                self.host_slabs.append(host)
                self.slabs.append(None)
                for ln in lins.values():
                    ln.w = ln.meta = None
                stream_any = True
                del slab
            else:
                self.host_slabs.append(None)
                self.slabs.append(slab)
            self.layers.append(lins)
        self._plans = {k: (self.layers[0][k].N, self.layers[0][k].K, self.layers[0][k].S) for k in self.layers[0]}
        self._m = UmbModel()
        m = self._m
        m.dtype, m.L, m.H, m.I, m.Hq, m.Hkv, m.D, m.V, m.Lmax = (_lib.dtype_code(dt), L, H, c.intermediate_size,
                                                               c.num_attention_heads, c.num_key_value_heads,
                                                               c.head_dim, Ve, self.max_length)   # V: rows of the embedding table
        m.eps, m.attn_scale = c.rms_norm_eps, 1.0 / math.sqrt(c.head_dim)
        m.embed = self.embed_tokens.data_ptr() if (self.embed_tokens is not None and self.is_first) else 0
        if self.is_last:
            m.lm_head, m.final_norm = self.lm_head.struct(), self.norm_weight.data_ptr()
        m.rope_cos, m.rope_sin = self.cos_cache.data_ptr(), self.sin_cache.data_ptr()
        m.k_cache, m.vt_cache = self.kv_cache.k.data_ptr(), self.kv_cache.vt.data_ptr()
        m.layers = C.cast(self._layer_structs, C.POINTER(UmbLayer))
        self._off = None
        if stream_any:
            # Device slab ring.  Two slabs (the reference's count, llama.py:160-167) suffice when every layer streams: the
            # link is the bottleneck and the compute of one layer hides inside the copy of the next.  With a device-resident
            # prefix (num_cache_layers) the link would sit idle while those layers compute once both slabs are full, so the
            # ring is as deep as that prefix is long in link time (70B-AWQ, 40 resident layers: 22 ms of compute = 2.8 slabs
            # of 444 MB at 56 GB/s); UMB_OFFLOAD_SLABS overrides.
            n_streamed = sum(1 for h in self.host_slabs if h is not None)
            ns = 2 if self.num_cache_layers <= 0 else _lib.MAX_SLABS
            ns = int(os.environ.get("UMB_OFFLOAD_SLABS", ns))
            ns = max(2, min(_lib.MAX_SLABS, ns, max(2, n_streamed)))
            self.n_slabs = ns
            self._dev_slabs = [torch.empty(slab_bytes, dtype=torch.uint8, device=dev) for _ in range(ns)]
            self.load_stream = torch.cuda.Stream(device=dev)
            self._events = [torch.cuda.Event() for _ in range(2 * ns)]
            for e in self._events:
                e.record()                 # materialise the hipEvent_t handles
            arr = (C.c_void_p * L)(*[(h.data_ptr() if h is not None else None) for h in self.host_slabs])
            self._host_arr = arr
            off = UmbOffload()
            off.host_slabs = C.cast(arr, C.POINTER(C.c_void_p))
            off.slab_bytes = slab_bytes
            off.n_slabs = ns
            for i in range(ns):
                off.dev_slab[i] = self._dev_slabs[i].data_ptr()
                off.ev_copied[i] = self._events[i].cuda_event
                off.ev_free[i] = self._events[ns + i].cuda_event
            off.copy_stream = self.load_stream.cuda_stream
            # cross-forward prefetch state: which layers the previous forward left streaming (UMB_OFFLOAD_PREFETCH=0:
            # the reference-free baseline that refetches them behind the draft's kernels)
            self._pf_state = (C.c_int32 * _lib.MAX_SLABS)(*([-1] * _lib.MAX_SLABS))
            if os.environ.get("UMB_OFFLOAD_PREFETCH", "1") != "0":
                off.prefetched = C.cast(self._pf_state, C.POINTER(C.c_int32))
            self._off = off
        self.reserve(self.CHUNK)
        torch.cuda.synchronize()

    # ------------------------------------------------------------------ workspace
    def use_gemv(self, on: bool):
        """Draft role switch (engines call it; default off).  on: forwards of <= 4 rows run on the row-streaming GEMV
        kernels (csrc/gemv.hip) wherever alloc() kept row-major weight copies; off: every T <= 64 forward of this
        model takes ONE shape-only kernel path, which is what makes a token's logits independent of how many rows
        share its launch (the batch invariance greedy spec == greedy AR rests on).  Changes launch arguments: call
        before any graph capture."""
        for i, lins in enumerate(self.layers):
            for key, ln in lins.items():
                ln.gemv_on = bool(on)
                rows = getattr(ln, "w_rows", None)
                streamed = self.host_slabs[i] is not None
                getattr(self._layer_structs[i], key).w_rows = rows.data_ptr() if (on and rows is not None and not streamed) else 0
        if getattr(self, "lm_head", None) is not None:
            self.lm_head.gemv_on = bool(on)
            rows = getattr(self.lm_head, "w_rows", None)
            self._m.lm_head.w_rows = rows.data_ptr() if (on and rows is not None and self._tp is None) else 0
        self.gemv = bool(on)
        self._chain_setup()

    def _chain_setup(self):
        """Persistent chain (csrc/chain.hip): with the draft role on and a covered shape (umb_chain_ok: 1B-class dense
        models, <= 3 rows, no q/k/v bias, a 256-CU device) the <= 3-row forwards run as tree attention + ONE persistent
        launch per layer.  The launch needs every workgroup resident, i.e. this process alone on the device:
        UMB_CHAIN=0 keeps the five GEMV launches.  A launch that cannot complete (a shared device) gives up after 20 ms and
        sets the status word; the engines then call disable_chain(), repair the draft's KV rows and carry on with the
        GEMV launches (engine_common.HipEngine._chain_fallback)."""
        c = self.config
        lib = _lib.load()
        on = (getattr(self, "gemv", False) and os.environ.get("UMB_CHAIN", "1") != "0" and self._tp is None
              and not getattr(self, "_chain_disabled", False)
              and self._off is None and hasattr(self, "_ws")
              and all(getattr(ln, "w_rows", None) is not None for lins in self.layers for ln in lins.values()))
        has_bias = int(any(getattr(st, "qkv_bias", None) for st in self._layer_structs))
        nqkv = c.q_dim + 2 * c.num_key_value_heads * c.head_dim
        if on and lib.umb_chain_ok(1, c.hidden_size, c.intermediate_size, nqkv, c.head_dim, has_bias):
            if getattr(self, "_chain_xchg", None) is None:
                n = lib.umb_chain_xchg_bytes(CHAIN_TMAX, c.hidden_size, c.intermediate_size)
                self._chain_xchg = torch.zeros(n, dtype=torch.uint8, device=self.device)
                _lib.check(lib.umb_chain_xchg_init(self._chain_xchg.data_ptr(), CHAIN_TMAX, c.hidden_size, c.intermediate_size,
                                                   _lib.stream_ptr()), "umb_chain_xchg_init")
            self._ws.chain_xchg = self._chain_xchg.data_ptr()
            off = CHAIN_TMAX * (c.hidden_size // 2 + c.intermediate_size // 2 + c.hidden_size // 2) * 8 + 64
            self.chain_status_word = self._chain_xchg[off:off + 4].view(torch.int32)     # device view (engines copy it out)
        elif hasattr(self, "_ws"):
            self._ws.chain_xchg = 0
            self.chain_status_word = None
        self.chain = bool(hasattr(self, "_ws") and self._ws.chain_xchg)

    def disable_chain(self):
        """Take the persistent chain out of this model's forwards (the five GEMV launches run instead -- bit-identical by
        tests/test_chain.py) and clear its give-up word.  Graphs captured with the chain in them must be dropped by the
        caller.  reset_chain() brings it back."""
        if hasattr(self, "_ws"):
            self._ws.chain_xchg = 0
        self.chain = False
        self._chain_disabled = True
        if getattr(self, "chain_status_word", None) is not None:
            self.chain_status_word.zero_()
        self.chain_status_word = None

    def reset_chain(self):
        """Re-initialise the exchange (a launch that gave up leaves stale granules behind) and re-enable the chain where the
        shape is covered; a no-op for models without it."""
        self._chain_disabled = False
        if getattr(self, "_chain_xchg", None) is not None:
            c = self.config
            self._chain_xchg.zero_()
            _lib.check(_lib.load().umb_chain_xchg_init(self._chain_xchg.data_ptr(), CHAIN_TMAX, c.hidden_size, c.intermediate_size,
                                                       _lib.stream_ptr()), "umb_chain_xchg_init")
        self._chain_setup()

    def chain_status(self) -> int:
        """Sticky give-up word of the persistent chain's bounded hand-off spins (0: every launch so far completed its
        hand-offs; 0xDEADxxxx: workgroup xxxx timed out -- the device was shared, results after that are invalid)."""
        if getattr(self, "_chain_xchg", None) is None:
            return 0
        c = self.config
        out = C.c_uint32(0)
        _lib.check(_lib.load().umb_chain_status(self._chain_xchg.data_ptr(), CHAIN_TMAX, c.hidden_size, c.intermediate_size,
                                                C.byref(out), _lib.stream_ptr()), "umb_chain_status")
        return int(out.value)

    def reserve(self, tokens: int, logit_rows: int | None = None):
        """Size the activation workspace for forwards of up to `tokens` rows (`logit_rows` of which may go through the
        lm_head; default all).  Re-allocates: call before any graph capture."""
        logit_rows = min(tokens, logit_rows or tokens)
        if tokens <= self.ws_tokens and logit_rows <= self.logit_rows:
            return
        tokens, logit_rows = max(tokens, self.ws_tokens), max(logit_rows, self.logit_rows)
        c, dev, dt = self.config, self.device, self.dtype
        tokens = (tokens + 15) // 16 * 16           # fragment-order activations come in 16-row tiles
        T = tokens
        self.logit_rows = logit_rows
        H, I, QD, V = c.hidden_size, c.intermediate_size, c.q_dim, c.vocab_size
        self.ws_tokens = T
        w = self._bufs = {}
        w["h"] = torch.zeros(T, H, dtype=dt, device=dev)
        w["xn"] = torch.zeros(T, H, dtype=dt, device=dev)
        w["q"] = torch.zeros(T, QD, dtype=dt, device=dev)
        w["attn"] = torch.zeros(T, QD, dtype=dt, device=dev)
        w["act"] = torch.zeros(T, I, dtype=dt, device=dev)
        part = max(S * T * N for (N, K, S) in self._plans.values()) + 64
        w["partial"] = torch.empty(part, dtype=torch.float32, device=dev)
        self.attn_chunk = max(128, (self.max_length // 16 + 31) // 32 * 32)
        self.attn_splits = (self.max_length + self.attn_chunk - 1) // self.attn_chunk
        w["po"] = torch.empty(self.attn_splits * T * QD, dtype=torch.float32, device=dev)
        w["ml"] = torch.empty(self.attn_splits * T * c.num_attention_heads * 2, dtype=torch.float32, device=dev)
        w["pos"] = torch.zeros(T, dtype=torch.int32, device=dev)
        w["slot"] = torch.zeros(T, dtype=torch.int32, device=dev)
        w["prefix"] = torch.zeros(1, dtype=torch.int32, device=dev)
        w["logits"] = torch.empty(logit_rows if self.is_last else 1, V if self.is_last else 8, dtype=torch.float32,
                                  device=dev)
        w["hw"] = torch.zeros(T, H, dtype=dt, device=dev)
        self.ssq_stride = max(H // 16, 256)         # low-latency schedule: one sum of squares per 16-column tile;
                                                    # GEMV schedule: one per workgroup (<= 256)
        w["ssq"] = torch.zeros(T, self.ssq_stride, dtype=torch.float32, device=dev)
        maxn = max(N for (N, K, S) in self._plans.values())
        if not hasattr(self, "_counters"):          # self-resetting arrival counters (zero between launches)
            self._counters = torch.zeros(maxn // 64 + 64, dtype=torch.int32, device=dev)
            g = c.num_attention_heads // c.num_key_value_heads
            self._attn_counters = torch.zeros(c.num_key_value_heads * ((1024 * g + 15) // 16) + 64, dtype=torch.int32,
                                              device=dev)      # one per (kv head, 16-row query tile), T <= 1024
        ws = self._ws = UmbWorkspace()
        ws.h, ws.xn, ws.q, ws.attn, ws.act = (w[k].data_ptr() for k in ("h", "xn", "q", "attn", "act"))
        ws.partial, ws.attn_po, ws.attn_ml = w["partial"].data_ptr(), w["po"].data_ptr(), w["ml"].data_ptr()
        ws.pos, ws.slot, ws.prefix, ws.logits = (w[k].data_ptr() for k in ("pos", "slot", "prefix", "logits"))
        ws.hw, ws.ssq = w["hw"].data_ptr(), w["ssq"].data_ptr()
        ws.counters, ws.attn_counters = self._counters.data_ptr(), self._attn_counters.data_ptr()
        ws.Tmax, ws.attn_chunk, ws.attn_splits, ws.ssq_stride = T, self.attn_chunk, self.attn_splits, self.ssq_stride
        ws.fused = 2 if (self.sched == "ll" and not self.fused) else int(self.fused)
        # schedule 0 with the RMSNorm deferred (model.hip layer_split_defer): many-blocks-per-row residual reduces; UMB_DEFER_NORM=0:
        # the one-block-per-row reduce that normalises in place (A/B; tensor-parallel shards always take that one)
        ws.defer_norm = int(ws.fused != 1 and os.environ.get("UMB_DEFER_NORM", "1") != "0")
        if getattr(self, "chain", False):
            self._chain_setup()

    @property
    def logits_buffer(self) -> torch.Tensor:
        return self._bufs["logits"]

    @property
    def hidden_buffer(self) -> torch.Tensor:
        return self._bufs["h"]

    # ------------------------------------------------------------------ forward
    def _run(self, step: UmbStep):
        lib = _lib.load()
        st = _lib.stream_ptr()
        if self._tp is not None:
            assert self._off is None, "tensor-parallel shards are device resident"
            rc = lib.umb_model_forward_tp(C.byref(self._m), C.byref(self._ws), C.byref(step), C.byref(self._tp), st)
        elif self._off is not None:
            rc = lib.umb_model_forward_offload(C.byref(self._m), C.byref(self._ws), C.byref(step), C.byref(self._off), st)
        else:
            rc = lib.umb_model_forward(C.byref(self._m), C.byref(self._ws), C.byref(step), st)
        _lib.check(rc, "umb_model_forward")

    def forward_tree(self, tokens_all, n_ptr, depth, tree_off, T, mask_bits, mask_words, head_from=0,
                     layer_range=None, skip_embed=False):
        """Tree-mode step: rows are tree nodes [tree_off, tree_off+T) of the engine's token buffer; all
        run-time indices derive from the device scalar *n_ptr (graph-capturable)."""
        s = UmbStep()
        s.T, s.tree_off = T, tree_off
        s.tokens_all, s.n_ptr, s.depth = tokens_all.data_ptr(), n_ptr.data_ptr(), depth.data_ptr()
        s.mask_bits = mask_bits.data_ptr() + tree_off * mask_words * 8
        s.mask_words, s.n_mask_keys = mask_words, tree_off + T
        s.head_from = head_from if self.is_last else T
        s.layer_begin, s.layer_end = layer_range or (0, self.num_layers)
        s.skip_embed = int(skip_embed or not self.is_first)
        self._run(s)

    def forward_explicit(self, tokens, positions, slots, prefix_len, mask_bits=None, mask_words=0, n_mask_keys=None,
                         head_from=0, layer_range=None, skip_embed=False):
        """Explicit-mode step: int32 device arrays tokens/positions/slots [T], prefix_len int32[1]."""
        T = tokens.shape[0]
        s = UmbStep()
        s.T = T
        s.tokens, s.positions, s.slots = tokens.data_ptr(), positions.data_ptr(), slots.data_ptr()
        s.prefix_len = prefix_len.data_ptr()
        s.mask_bits = mask_bits.data_ptr() if mask_bits is not None else 0
        s.mask_words = mask_words
        s.n_mask_keys = T if n_mask_keys is None else n_mask_keys
        s.head_from = head_from if self.is_last else T
        s.layer_begin, s.layer_end = layer_range or (0, self.num_layers)
        s.skip_embed = int(skip_embed or not self.is_first)
        self._keep = (tokens, positions, slots, prefix_len, mask_bits)
        self._run(s)

    @torch.inference_mode()
    def prefill_tokens(self, ids: torch.Tensor, start: int, want_logits=True):
        """Causal forward over ids (int32 [P], device) placed at slots/positions start.. ; returns the
        fp32 logits row of the last token (view into the workspace) if requested."""
        P = ids.shape[0]
        dev = self.device
        out = None
        chunk = max(self.CHUNK, min(self.PREFILL_CHUNK, self.ws_tokens))
        for lo in range(0, P, chunk):
            hi = min(P, lo + chunk)
            T = hi - lo
            pos = torch.arange(start + lo, start + hi, dtype=torch.int32, device=dev)
            pre = torch.tensor([start + lo], dtype=torch.int32, device=dev)
            last = hi == P and want_logits
            self.forward_explicit(ids[lo:hi].contiguous(), pos, pos, pre, head_from=(T - 1 if last else T))
            if last:
                out = self._bufs["logits"][0]
        self.kv_cache.kv_offset = start + P
        return out

    @torch.inference_mode()
    def inference(self, input_ids: torch.LongTensor, position_ids: torch.LongTensor, attention_mask: torch.Tensor,
                  storage_ids: torch.LongTensor):
        """Reference face (umbrella/models/llama.py:117-134).  attention_mask: bool [T, >= kv] (True = attend);
        columns are cache slots.  Rows may only attend slots that were written before or by this call."""
        dev = self.device
        ids = input_ids.reshape(-1).to(device=dev, dtype=torch.int32)
        pos = position_ids.reshape(-1).to(device=dev, dtype=torch.int32)
        slots = storage_ids.reshape(-1).to(device=dev, dtype=torch.int32)
        mask = attention_mask.to(dev)
        T = ids.shape[0]
        allrows = mask.all(dim=0).to(torch.int32)
        prefix = int(allrows.cumprod(0).sum().item())                    # leading columns every row attends
        anyc = mask.any(dim=0).nonzero()
        kv_end = int(anyc[-1].item()) + 1 if anyc.numel() else prefix
        nmk = max(kv_end - prefix, 1)
        bits = pack_mask_bits(mask[:, prefix:prefix + nmk])
        W = bits.shape[1]
        pre = torch.tensor([prefix], dtype=torch.int32, device=dev)
        V = self.config.vocab_size
        out = torch.empty(T, V, dtype=torch.float32, device=dev)
        for lo in range(0, T, self.logit_rows):
            hi = min(T, lo + self.logit_rows)
            self.forward_explicit(ids[lo:hi].contiguous(), pos[lo:hi].contiguous(), slots[lo:hi].contiguous(), pre,
                                  mask_bits=bits[lo:hi].contiguous(), mask_words=W, n_mask_keys=nmk, head_from=0)
            out[lo:hi] = self._bufs["logits"][:hi - lo]
        self.kv_cache.kv_offset = max(self.kv_cache.kv_offset, int(slots.max().item()) + 1)
        return out[None]

    def graph_inference(self, input_ids, storage_ids, position_ids=None, attention_mask=None):
        return self.inference(input_ids, position_ids, attention_mask, storage_ids)

    def gather_kv_incremental(self, indices: torch.LongTensor, offset: int):
        self.kv_cache.gather_kv_incremental(indices, offset)

    def clear(self):
        self.kv_cache.clear()

    def host_placement(self, copies: int = 8) -> dict:
        """Where the streamed slabs live on the host and what the link delivers from there: arena / NUMA node
        (models/host_arena.py) and the rate of `copies` single-slab host-to-device copies on the side stream (GB/s: min,
        median, max) -- the figures the bench line reports next to the offload step time."""
        if self._off is None:
            return {}
        info = self._host_arena.describe() if self._host_arena is not None else \
            {"arena": "one torch pin_memory() tensor per layer", "numa_node": None}
        slabs = [h for h in self.host_slabs if h is not None]
        rates = []
        with torch.cuda.stream(self.load_stream):
            for i in range(copies + 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self._dev_slabs[i % len(self._dev_slabs)].copy_(slabs[i % len(slabs)], non_blocking=True)
                e1.record()
                e1.synchronize()
                if i:
                    rates.append(self.slab_bytes / e0.elapsed_time(e1) / 1e6)
        rates.sort()
        info["copy_GBs"] = {"min": round(rates[0], 1), "median": round(rates[len(rates) // 2], 1), "max": round(rates[-1], 1)}
        if self._pf_state is not None:
            for i in range(len(self._pf_state)):
                self._pf_state[i] = -1                       # the probe overwrote the device slabs: nothing is prefetched
        return info

    # bytes read from HBM by one forward over all layers + lm_head (algorithmic, for the roofline)
    def weight_bytes(self) -> int:
        c = self.config
        per_layer = 0
        for lins in self.layers[:1]:
            for ln in lins.values():
                per_layer += (ln.N * ln.K // 2 + (ln.N // 16) * (ln.K // 128) * 64) if ln.awq else ln.N * ln.K * 2
        return per_layer * self.num_layers + (self.lm_head.N * self.lm_head.K * 2 if self.lm_head is not None else 0)
