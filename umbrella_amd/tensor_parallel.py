"""Tensor-parallel verify across the GPUs of one node (the reference has no distributed code).

Layer sharding (parallel.py) adds capacity; the only route to a faster single request beyond one GPU's HBM is to
split every layer: rank r of P holds
    q / k / v rows of its Hq/P query and Hkv/P key-value heads   (column split: no communication)
    the o_proj columns of those heads                            (row split: fp32 partial sums -> all-reduce)
    gate / up rows [r I/P, (r+1) I/P)                            (column split)
    the down_proj columns of that slice                          (row split: fp32 partial sums -> all-reduce)
    lm_head rows [r V/P, (r+1) V/P)                              (column split: logits all-gathered along the vocabulary)
with norms, the residual stream and the embedding replicated.

The layer chain is the SAME native one a single GPU runs (csrc/model.hip, the 8-launch schedule, any T: tree verify,
wide dynamic trees, 1024-token prompt chunks): `umb_model_forward_tp` calls back for an all-reduce behind the two
row-split GEMMs of every layer, and this module answers with RCCL (`torch.distributed` "nccl") on the launch stream --
so a whole iteration, collectives included, still replays as ONE hipGraph per rank; at world 1 no collective is issued
and the path is the plain engine's.  xGMI is point to point: the collective carries exactly one [T, H] fp32 tile (T H 4
bytes: 426 KB at T = 13, H = 8192 -- the split-K slabs are summed first, csrc/model.hip: tp_allreduce), latency bound,
160 of them per 70B verify.
The engines are the ordinary ones (static or dynamic, greedy or stochastic): the target they see is
`TensorParallelLlama`, whose `logits_buffer` holds the all-gathered [T, V] logits, so sampling, accept scan and KV
compaction are unchanged; every rank runs the same engine (SPMD) with the small draft replicated -- deterministic
kernels and a shared seed make all ranks grow and accept the same tree without a control channel.
Tests: sharding algebra on CPU (gloo, world 2); two processes sharing one GPU over host-staged gloo reproduce the
single-process engine's tokens (tests/test_tensor_parallel.py); RCCL itself is exercised at world 1 on the test box,
including its capture into the iteration graph.
"""
from __future__ import annotations

import copy
import os

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
    c.embed_rows = cfg.vocab_size               # the embedding table stays whole (replicated); only the head is split
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


# ------------------------------------------------------------------ communicator
class TPComm:
    """The tensor-parallel group: RCCL ("nccl") with one GPU per rank; gloo (host staged) where ranks share a GPU (tests)
    or live on the CPU.  Collectives are issued on the current stream, so under hipGraph capture they become graph nodes."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.live = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.live else 0
        self.world = dist.get_world_size(group) if self.live else 1
        self.backend = dist.get_backend(group) if self.live else "none"
        self.staged = self.backend == "gloo"

    def all_reduce(self, t: torch.Tensor):
        if not self.live:
            return
        if self.staged and t.is_cuda:
            h = t.cpu()
            self.dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, group=self.group)

    def all_gather_columns(self, local: torch.Tensor, full: torch.Tensor, scratch: torch.Tensor | None = None):
        """local [rows, Vl] of every rank -> full [rows, world * Vl] (rank r's columns at [r Vl, (r + 1) Vl))"""
        rows, Vl = local.shape
        if self.world == 1:
            full.copy_(local)
            return
        if self.staged and local.is_cuda:
            h = local.cpu()
            parts = [torch.empty_like(h) for _ in range(self.world)]
            self.dist.all_gather(parts, h, group=self.group)
            full.copy_(torch.cat(parts, dim=1))
            return
        buf = scratch[:self.world * rows * Vl].view(self.world, rows, Vl) if scratch is not None else \
            torch.empty(self.world, rows, Vl, dtype=local.dtype, device=local.device)
        self.dist.all_gather_into_tensor(buf.view(-1), local.reshape(-1), group=self.group)
        full.view(rows, self.world, Vl).copy_(buf.transpose(0, 1))


# ------------------------------------------------------------------ direct peer all-reduce (csrc/tp.hip)
class PeerExchangeUnavailable(RuntimeError):
    """raised by PeerExchange on EVERY rank of the group when any rank failed to allocate or map (a collective verdict)"""


class PeerExchange:
    """The exchange buffers of the direct-xGMI all-reduce: every rank allocates [2][cap] floats + one flag line
    (fine-grained device memory where the runtime allows), hands its interprocess handle round the group once, maps the
    peers' buffers and fills an UmbTPPeer descriptor the native layer chain reads.  Setup only: the data path is two
    kernel launches per all-reduce in the rank's own stream (umb_tp_publish / umb_tp_reduce_residual_norm)."""
    FLAG_BYTES = 256

    def __init__(self, comm: "TPComm", cap_floats: int, device, require_fine_grained: bool = False):
        import ctypes as C
        dist = comm.dist
        lib = _lib.load()
        self.lib, self.world, self.rank = lib, comm.world, comm.rank
        assert 2 <= self.world <= _lib.TP_MAX_RANKS
        self.cap = (int(cap_floats) + 63) // 64 * 64
        nbytes = 2 * self.cap * 4 + self.FLAG_BYTES
        # Every step that can fail on ONE rank (allocation, mapping a peer's handle) is followed by a group-wide
        # agreement: a rank that raised on its own used to leave the others inside all_gather_object / the closing barrier,
        # or on the peer path against a rank already on the collective hook.  Every rank reaches every collective below
        # whatever happened locally; the path is enabled only if all ranks succeeded, otherwise everybody cleans up and
        # raises the same PeerExchangeUnavailable (the caller then takes the hook on ALL ranks).
        self.base, self.fine_grained, self.peer_ptrs, self._opened = 0, False, [0] * self.world, []
        with torch.cuda.device(device):
            handle = (C.c_ubyte * 64)()
            err = None
            try:
                base, fg = C.c_void_p(0), C.c_int(0)
                _lib.check(lib.umb_tp_xchg_alloc(nbytes, C.byref(base), handle, C.byref(fg)), "umb_tp_xchg_alloc")
                self.base, self.fine_grained = base.value, bool(fg.value)
            except Exception as e:
                err = e
            handles = [None] * self.world
            dist.all_gather_object(handles, (self.rank, bytes(handle), os.getpid(), err is None), group=comm.group)
            if all(h[3] for h in handles):
                try:
                    for r, hb, pid, _ok in handles:
                        if r == self.rank:
                            self.peer_ptrs[r] = self.base
                            continue
                        p = C.c_void_p(0)
                        buf = (C.c_ubyte * 64).from_buffer_copy(hb)
                        _lib.check(lib.umb_tp_xchg_open(buf, C.byref(p)), f"umb_tp_xchg_open(rank {r})")
                        self.peer_ptrs[r] = p.value
                        self._opened.append(p.value)
                except Exception as e:
                    err = e
            elif err is None:
                err = RuntimeError("a peer rank could not allocate its exchange buffer")
            oks = [None] * self.world
            pr = torch.cuda.get_device_properties(torch.device(device))
            where = (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            dist.all_gather_object(oks, (err is None, self.fine_grained, where), group=comm.group)
            if not all(o[0] for o in oks):
                self.close()
                raise PeerExchangeUnavailable(str(err) if err is not None else "a peer rank could not map the exchange buffers")
            # Ordinary (coarse-grained) device memory read ACROSS devices has never run on hardware (every box of this pool has one
            # GPU): in "auto" the group takes the collective hook there.  Ranks sharing one device (the tests) read through one
            # L2 and keep the path; UMB_TP_ALLREDUCE=peer keeps it anywhere.  Decided from gathered facts: the same on every rank.
            if require_fine_grained and len({o[2] for o in oks}) > 1 and not all(o[1] for o in oks):
                self.close()
                raise PeerExchangeUnavailable("exchange buffers in ordinary device memory on more than one device: "
                                              "not validated, UMB_TP_ALLREDUCE=peer forces it")
        self.words = torch.zeros(64, dtype=torch.int32, device=device)       # epoch | arrive | status, 64 bytes apart
        d = _lib.UmbTPPeer()
        d.rank, d.world, d.cap, d.spin_limit = self.rank, self.world, self.cap, int(os.environ.get("UMB_TP_SPIN", "0"))
        for r in range(self.world):
            d.slot[r] = self.peer_ptrs[r] + self.FLAG_BYTES
            d.flag[r] = self.peer_ptrs[r]
        d.epoch, d.arrive, d.status = self.words.data_ptr(), self.words.data_ptr() + 64, self.words.data_ptr() + 128
        self.desc = d
        dist.barrier(group=comm.group)                # every mapping exists before anyone publishes

    def status(self) -> int:
        return int(self.words[32].item()) & 0xffffffff

    def close(self):
        for p in getattr(self, "_opened", []):
            self.lib.umb_tp_xchg_close(p)
        self._opened = []
        if getattr(self, "base", 0):
            self.lib.umb_tp_xchg_free(self.base)
            self.base = 0

    def reset_status(self):
        self.words[32] = 0


# ------------------------------------------------------------------ the sharded target
class TensorParallelLlama:
    """This rank's shard of one Llama target behind the model-runtime face the engines use.  `shard` is a `Llama` built on
    `local_config` (1/P of the heads, the MLP width and the vocabulary; whole embedding table) whose forward runs
    `umb_model_forward_tp`; this wrapper owns the all-reduce hook and the vocabulary all-gather of the logits."""

    PEER_MAX_ROWS = 64                   # tiles of up to this many rows take the direct peer all-reduce (csrc/tp.hip)

    def __init__(self, cfg: LlamaCfg, shard, comm: TPComm, force_hook: bool = False):
        import ctypes as C
        self.config, self.m, self.comm = cfg, shard, comm
        self.world, self.rank = comm.world, comm.rank
        self.device, self.dtype, self.max_length = shard.device, shard.dtype, shard.max_length
        self.eos_tokens = list(cfg.eos_token_id)
        self.num_layers, self.kv_cache = shard.num_layers, shard.kv_cache
        self.CHUNK, self.PREFILL_CHUNK = shard.CHUNK, shard.PREFILL_CHUNK
        self._off = None
        self.sched = "split"
        self._err = None
        # which transport answers the layer chain's all-reduce hook (reported on the bench line)
        self.allreduce_path = ("gloo, host staged (tests: ranks sharing a GPU)" if comm.staged else
                               f"{comm.backend}: torch.distributed all_reduce (RCCL) of one [T, H] fp32 tile on the launch "
                               "stream, captured into the iteration hipGraph") if comm.live else "none"

        def hook(ctx, buf, count, stream):
            try:
                part = self.m._bufs["partial"]
                assert buf == part.data_ptr() and count <= part.numel(), "all-reduce outside the split-K partial buffer"
                self.comm.all_reduce(part[:count])
                return 0
            except Exception as e:                      # never unwind through the C frame
                self._err = e
                return 1
        self._hook = _lib.ALLREDUCE_FN(hook)            # keep the trampoline alive as long as the model
        tp = _lib.UmbTP()
        # force_hook (tests): issue the collectives even in a 1-rank group (RCCL all-reduce over one rank = identity)
        tp.rank, tp.world = self.rank, (max(self.world, 2) if force_hook else self.world)
        tp.allreduce, tp.ctx = self._hook, None
        # Direct peer reads for the small tiles (tree verify: T <= 64 rows): UMB_TP_ALLREDUCE = auto | peer | hook.  "auto"
        # takes the peer path wherever the exchange buffers can be mapped (one GPU per rank over RCCL, or ranks sharing a
        # GPU in the tests) and falls back to the hook with a warning; tiles above PEER_MAX_ROWS rows always take the hook.
        self.peer = None
        mode = os.environ.get("UMB_TP_ALLREDUCE", "auto")
        if comm.live and comm.world > 1 and mode != "hook" and str(self.device).startswith("cuda"):
            try:
                self.peer = PeerExchange(comm, self.PEER_MAX_ROWS * cfg.hidden_size, self.device, require_fine_grained=(mode == "auto"))
                self._peer_ptr = C.pointer(self.peer.desc)       # kept: reading `tp.peer` back yields a VIEW of the field, not a copy
                tp.peer = self._peer_ptr
                tp.peer_max_floats = self.PEER_MAX_ROWS * cfg.hidden_size
                self.allreduce_path = (f"peer: direct reads of the P [T, H] fp32 tiles through hipIpc-mapped exchange buffers "
                                       f"({'fine-grained' if self.peer.fine_grained else 'ordinary'} device memory), summed in "
                                       f"rank order inside the residual / norm kernel, for tiles of <= {self.PEER_MAX_ROWS} rows; "
                                       "larger tiles: " + self.allreduce_path)
            except Exception as e:
                if mode == "peer":
                    raise
                import warnings
                warnings.warn(f"tensor parallel: peer exchange unavailable ({type(e).__name__}: {e}); all-reduce through the hook")
        shard._tp = tp
        self._alloc_gather()

    @classmethod
    def build(cls, cfg: LlamaCfg, source, comm: TPComm, max_length, device, dtype, name="tp", seed=0, force_hook=False):
        """source: a full state dict (sliced here with shard_state_dict), or any mapping name -> THIS rank's slice
        (LazySyntheticShard; a checkpoint reader wrapped with shard_tensor)."""
        from .models.llama import Llama
        lc = local_config(cfg, comm.world)
        sd = dict(shard_state_dict(source, cfg, comm.rank, comm.world)) if isinstance(source, dict) else source
        m = Llama(f"{name}-tp{comm.rank}of{comm.world}", max_length=max_length, device=device, dtype=dtype, state_dict=sd,
                  config=lc, seed=seed, sched="split")
        m.fused = False          # umb_model_forward_tp runs the 8-launch schedule only: UMB_FUSED=1 in the environment must not reach a shard
        m.alloc()
        return cls(cfg, m, comm, force_hook=force_hook)

    # ---- buffers
    def _alloc_gather(self):
        rows = self.m.logit_rows
        V, Vl = self.config.vocab_size, self.m.config.vocab_size
        assert V == Vl * self.world
        self._full = torch.empty(rows, V, dtype=torch.float32, device=self.device)
        self._scratch = torch.empty(self.world * rows * Vl, dtype=torch.float32, device=self.device) \
            if (self.world > 1 and not self.comm.staged) else None

    def reserve(self, tokens, logit_rows=None):
        before = (self.m.ws_tokens, self.m.logit_rows)
        self.m.reserve(tokens, logit_rows)
        if (self.m.ws_tokens, self.m.logit_rows) != before:
            self._alloc_gather()

    @property
    def logits_buffer(self) -> torch.Tensor:
        return self._full

    @property
    def ws_tokens(self):
        return self.m.ws_tokens

    @property
    def logit_rows(self):
        return self.m.logit_rows

    def _raise_hook_error(self):
        """A Python exception inside the all-reduce hook reaches the C chain as rc = 1 (the forward aborts with -5 on this
        rank while its peers wait inside the collective): surface the cause at once and take the group down with it
        instead of leaving the peers to hang until their own timeout."""
        if self._err is None:
            return
        e, self._err = self._err, None
        try:
            if self.comm.live and self.world > 1:
                import torch.distributed as dist
                dist.destroy_process_group(self.comm.group) if self.comm.group is not None else dist.destroy_process_group()
        except Exception:
            pass
        raise e

    def _guarded(self, fn, *a, **kw):
        try:
            fn(*a, **kw)
        except RuntimeError:
            self._raise_hook_error()                         # the hook's own exception, not "umb_model_forward failed: -5"
            raise
        self._raise_hook_error()

    def close(self, engines=()):
        """Ordered release.  A collective captured into a hipGraph keeps referring to its communicator's
        resources: the graphs go FIRST, then the stream is drained, then the peers' exchange buffers are unmapped and this
        rank's freed -- only after that may the caller destroy the process group (shutdown_tensor_parallel does all of it).
        `engines`: every engine that ran over this target (their captured iteration graphs are dropped here)."""
        import gc
        for e in engines:
            if getattr(e, "_graph", None) is not None:
                e._graph = None
        gc.collect()                                    # torch frees a CUDAGraph's exec on collection, not on rebinding
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if self.peer is not None:
            import ctypes as C
            self.m._tp.peer = C.POINTER(_lib.UmbTPPeer)()
            self.peer.close()
            self.peer = None

    def peer_self_check(self, rows: int = 13, tol: float = 0.02, rounds: int = 3) -> bool:
        """The direct peer all-reduce has only ever run with the ranks on ONE device (where every rank shares an L2); on a
        real node its cross-device visibility is decided by the hardware.  So before it is trusted: one T-row forward through
        the peer path and the same forward through the collective hook (RCCL) -- both deterministic, both summing the P tiles --
        must give the same logits up to fp32 summation order on EVERY rank; otherwise the peer path is switched off for this
        model (collectively) and `allreduce_path` says so.  Costs two small forwards at load time."""
        if self.peer is None or self.world < 2:
            return True
        import ctypes as C
        dev = self.device
        g = torch.Generator().manual_seed(97)
        pos = torch.arange(rows, dtype=torch.int32, device=dev)
        pre = torch.zeros(1, dtype=torch.int32, device=dev)
        tp = self.m._tp
        saved = self._peer_ptr
        diff = scale = 0.0
        # several rounds with different rows: a stale read touches few elements and not every call; every
        # forward is 2 L exchanges back to back, so both slots (epoch parities) are exercised many times per round
        for _ in range(max(1, rounds)):
            ids = torch.randint(3, max(4, self.config.vocab_size - 1), (rows,), generator=g).int().to(dev)
            outs = []
            for use_peer in (True, False):
                tp.peer = saved if use_peer else C.POINTER(_lib.UmbTPPeer)()
                self.m.clear()
                self.forward_explicit(ids, pos, pos, pre, head_from=0)
                torch.cuda.synchronize()
                outs.append(self.logits_buffer[:rows].clone())
            scale = max(scale, float(outs[1].abs().max()))
            diff = max(diff, float((outs[0] - outs[1]).abs().max()))
        self.m.clear()
        self.last_self_check = {"max_abs_diff": diff, "scale": scale, "rounds": rounds}
        bad = not (diff <= tol * max(scale, 1.0)) or self.peer.status() != 0
        flag = torch.tensor([1.0 if bad else 0.0], device="cpu" if self.comm.staged else dev)
        self.comm.dist.all_reduce(flag, group=self.comm.group)          # any rank's doubt switches every rank off
        ok = float(flag[0]) == 0.0
        if ok:
            tp.peer = saved
        else:
            tp.peer = C.POINTER(_lib.UmbTPPeer)()
            self.allreduce_path = "hook only (the peer all-reduce FAILED its self-check against the collective on this node): " + \
                self.allreduce_path.split("larger tiles: ")[-1]
            self.peer_disabled = True
        return ok

    @property
    def peer_status_word(self):
        """device view of the peer path's sticky give-up word (None without a peer exchange): the engines copy it out with
        every iteration's accept result, so a reduce that gave up is raised at the next host sync, not at the next clear()"""
        if self.peer is None or getattr(self, "peer_disabled", False):
            return None
        return self.peer.words[32:33]

    def _check_peer(self):
        if self.peer is not None and not getattr(self, "peer_disabled", False):
            st = self.peer.status()
            if st:
                self.peer.reset_status()             # sticky on the device, not across the exception
                raise RuntimeError(f"tensor parallel: a peer never published its tile (status {st:#x}: row block {st & 0xffff}); "
                                   "the group is out of step or a rank died")

    def _gather(self, rows):
        if rows > 0:
            self.comm.all_gather_columns(self.m._bufs["logits"][:rows], self._full[:rows], self._scratch)

    # ---- the model-runtime face
    def forward_tree(self, tokens_all, n_ptr, depth, tree_off, T, mask_bits, mask_words, head_from=0, **kw):
        self._guarded(self.m.forward_tree, tokens_all, n_ptr, depth, tree_off, T, mask_bits, mask_words, head_from=head_from, **kw)
        self._gather(T - head_from)

    def forward_explicit(self, tokens, positions, slots, prefix_len, mask_bits=None, mask_words=0, n_mask_keys=None,
                         head_from=0, **kw):
        self._guarded(self.m.forward_explicit, tokens, positions, slots, prefix_len, mask_bits=mask_bits,
                      mask_words=mask_words, n_mask_keys=n_mask_keys, head_from=head_from, **kw)
        self._gather(tokens.shape[0] - head_from)

    @torch.inference_mode()
    def prefill_tokens(self, ids, start, want_logits=True):
        """causal forward over ids placed at slots / positions start.. (Llama.prefill_tokens, sharded): the fp32 logits
        row of the last token, all V columns, when asked"""
        P = ids.shape[0]
        out = None
        chunk = max(self.CHUNK, min(self.PREFILL_CHUNK, self.m.ws_tokens))
        for lo in range(0, P, chunk):
            hi = min(P, lo + chunk)
            T = hi - lo
            pos = torch.arange(start + lo, start + hi, dtype=torch.int32, device=self.device)
            pre = torch.tensor([start + lo], dtype=torch.int32, device=self.device)
            last = hi == P and want_logits
            self.forward_explicit(ids[lo:hi].contiguous(), pos, pos, pre, head_from=(T - 1 if last else T))
            if last:
                out = self._full[0]
        self.kv_cache.kv_offset = start + P
        self._check_peer()                       # once per prompt and per clear(): never inside a captured iteration
        return out

    def gather_kv_incremental(self, indices, offset):
        self.kv_cache.gather_kv_incremental(indices, offset)

    def clear(self):
        try:
            self._check_peer()
        finally:
            self.m.clear()                           # the caches are cleared whatever the peer path reported

    def weight_bytes(self):
        return self.m.weight_bytes()


def build_tp_engine(device: str, dtype=torch.float16, seed: int = 0, source=None, comm: TPComm | None = None,
                    force_hook: bool = False, **config):
    """Reference-style engine config (``engine``, ``model``, ``draft_model``, ``growmap`` | ``growmap_path`` | width /
    num_beams / depth, sampling knobs ...) -> the ordinary static / dynamic engine over a tensor-parallel target.
    SPMD: every rank of the group calls this and drives its engine identically.  `source`: name -> tensor mapping of
    the target's checkpoint (default: seeded synthetic tensors of the configured shapes, UMBRELLA_SYNTHETIC=1)."""
    from .models.auto_model import AutoModelLM
    from .models.config import KNOWN
    from .speculation.auto_engine import AutoEngine
    comm = comm or TPComm()
    target = config["model"]
    cfg = KNOWN[target] if target in KNOWN else LlamaCfg.from_dir(target)
    max_length = config.get("max_length", 8192)
    if source is None:
        source = LazySyntheticShard(cfg, comm.rank, comm.world, device, dtype, seed=seed)
    tp = TensorParallelLlama.build(cfg, source, comm, max_length, device, dtype, name=target, seed=seed, force_hook=force_hook)
    tp.peer_self_check()
    for k in ("offload", "num_cache_layers"):                    # single-GPU placement knobs of the reference
        config.pop(k, None)
    if config.get("engine", "dynamic") == "dynamic":
        config["offload"] = False
    if comm.staged:
        config["hip_graph"] = False                              # a host-staged collective cannot live in a hipGraph
    eng = AutoEngine.from_config(device, dtype=dtype, seed=seed, target_model_obj=tp, **config)
    return eng


def allreduce_probe(device, numel, reps=50):
    """One isolated all-reduce of the [T, H] fp32 tile a row-split GEMM hands over, on the launch stream, timed with
    events behind a warm-up; -> microseconds per call (None in a 1-rank group or on a host-staged backend)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() < 2 or dist.get_backend() == "gloo":
        return None
    buf = torch.zeros(numel, dtype=torch.float32, device=device)
    for _ in range(5):
        dist.all_reduce(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    dist.barrier()
    e0.record()
    for _ in range(reps):
        dist.all_reduce(buf)
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / reps, 2)


def tp_measure(args, wl, dtype, device, rank, world):
    """ONE request, every layer of the target split over the ranks of the default process group (static 3x4, greedy).
    SPMD -- every rank runs this; all return the result dict.  Nothing is printed, the group stays up.
    The acceptance knob is the headline's (same acc vector, same seed: speculation/steering.py; every rank steers with
    the same recorded continuation, so the ranks keep taking identical decisions), `allreduce_us` = one isolated
    all-reduce of the [T, H] fp32 tile, `devices` = the physical device identities gathered over the group."""
    import torch.distributed as dist

    import __graft_entry__ as ge
    from .models.config import KNOWN
    from .sequoia_utils import DEFAULT_ACC, generate_sequoia_tree
    from .speculation.steering import device_census, steered_measure
    ge.build()
    cfg = KNOWN[wl["target"]]
    census = device_census(device, dist if dist.is_initialized() else None)
    gm = generate_sequoia_tree(3, 4)
    ar_us = allreduce_probe(device, gm["size"] * cfg.hidden_size)
    eng = build_tp_engine(device, dtype=dtype, seed=args.seed, engine="static", model=wl["target"], draft_model=wl["draft"],
                          growmap=gm, max_length=args.max_length, exit_layer=16)
    eng.initialize()
    g = torch.Generator().manual_seed(1234)
    prompt = torch.randint(3, wl.get("vocab_hi", 128000), (1, args.prompt_len), generator=g)
    acc = list(DEFAULT_ACC)
    barrier = dist.barrier if world > 1 else None
    r = steered_measure(eng, prompt, acc, args.seed, args.warmup, args.steps, len(gm["roots"]), barrier=barrier)
    tgt = eng.target_model
    r.update({"n_ranks_rccl": dist.get_world_size() if dist.is_initialized() else 1,
              "backend": dist.get_backend() if dist.is_initialized() else "none", "devices": census["devices"],
              "n_distinct_devices": census["n_distinct"],
              "allreduces_per_verify": 2 * cfg.num_hidden_layers if world > 1 else 0,
              "allreduce_bytes": eng.tree_size * cfg.hidden_size * 4, "allreduce_us": ar_us,
              "allreduce_path": getattr(tgt, "allreduce_path", "none") if world > 1 else "none (1 rank: the hook is not called)",
              "peer_self_check": getattr(tgt, "last_self_check", None),
              "iteration_in_one_hipgraph": bool(eng.use_graph and eng.graph_scope == "iteration"), "acc": acc,
              "parallelism": f"tp{world}: heads / MLP width / vocabulary split, 2 all-reduces of the [T, H] fp32 partial "
                             "sums per layer inside the native layer chain; draft replicated", "tree": "3x4", "scaling": "strong"})
    # the measurement is done: graphs first, then the peers' buffers (the caller destroys the group afterwards)
    shutdown_tensor_parallel(tgt if hasattr(tgt, "close") else None, [eng], destroy_group=False)
    return r


def run_tp_bench(args, wl, dtype, device, rank, world):
    """bench.py --parallel tp: the tensor-parallel engine alone, one JSON line from rank 0."""
    import json

    import torch.distributed as dist
    r = tp_measure(args, wl, dtype, device, rank, world)
    out = None
    if rank == 0:
        out = {"metric": "tokens/s @ bs=1 (speculative decoding)", "value": r["tokens_per_s"], "unit": "tokens/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": wl["dtype"],
               "data": "synthetic: random-init weights, random prompt; acceptance set by the controllable-acceptance draft "
                       "(the headline's acc vector and seed)",
               "config": {"workload": wl["desc"], "parallelism": r["parallelism"], "tree": "3x4", "prompt_len": args.prompt_len},
               "accept_len": r["accept_len"], "value_raw_draft": r["tokens_per_s_raw_draft"], "tp": r}
        print(json.dumps(out), flush=True)
    shutdown_tensor_parallel(destroy_group=True)               # also the 1-rank group bench.py opens for the RCCL smoke run
    return out


def shutdown_tensor_parallel(target=None, engines=(), destroy_group: bool = True):
    """Tear a tensor-parallel setup down in the one order that is safe with RCCL collectives captured in hipGraphs:
    graphs -> stream drain -> peer buffers -> (barrier) -> process group.  Destroying the communicator while a live graph
    still holds its kernels aborted the process now and then (round 5: 2 of 8 runs, worked around with os._exit)."""
    import gc

    import torch.distributed as dist
    if target is not None:
        target.close(engines)
    else:
        for e in engines:
            if getattr(e, "_graph", None) is not None:
                e._graph = None
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    if destroy_group and dist.is_available() and dist.is_initialized():
        try:
            dist.barrier()
        except Exception:
            pass
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        dist.destroy_process_group()
