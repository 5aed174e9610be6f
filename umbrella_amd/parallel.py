"""Layer-sharded verify across the GPUs of one node (BASELINE config 5; new functionality -- the
reference has no distributed code at all).

One process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU
tests.  Rank 0 owns the draft model, the engine state and pipeline stage 0; rank r owns decoder layers
[lo_r, hi_r) of the target and their KV slice.  A batch-1 request is a strictly sequential chain of layers,
so this buys capacity, not speed: per verify the [T, H] activation makes world-1 point-to-point hops
(213 KB at T = 13 -- latency bound, one direct xGMI link per hop; no ring collective is involved), the last
stage returns the T sampled ids, and rank 0 broadcasts the accept result so every stage compacts its own
KV slice.  Token ids are identical to the single-GPU result: same kernels, same order.

Protocol.  Stage ranks follow a fixed op schedule; rank 0 speaks only at mode changes:
    plan  (int64[8] broadcast)  [OP_PREFILL, P, start, want_head]   every stage then walks the same ceil(P / chunk)
                                                                    causal chunks: recv -> forward -> send
                                [OP_DECODE]                         enter the decode loop
                                [OP_RESET] / [OP_STOP]
                                [OP_DECODE, sampling knobs]         (greedy flag, temperature, top-p, penalty, top-k, seed)
    decode loop, per iteration: (dynamic trees: ONE broadcast of the beam-grown mask rows) recv [T, H] -> forward_tree
                                (hipGraph) -> send; last stage: arg-max, or umb_sample_rows over its own copy of the token
                                history -> T ids to rank 0; then ONE broadcast int32[8 + 2 max_path + 1] = accept result +
                                path + the tokens committed by this iteration + `cont` flag: KV compaction, history
                                update, and cont = 0 leaves the loop (back to waiting for a plan).
    A prompt's token ids follow their OP_PREFILL plan in one broadcast (the last stage penalises repetitions over them).
Rank 0 defers the commit of iteration i until it knows what follows (the next step(): cont = 1; reset / new
prompt / shutdown: cont = 0), so the decode loop costs exactly one collective per iteration besides the hops, and a
stage rank reads back one word per iteration (the flag).  Activations are received straight into the stage model's
hidden buffer; prompt chunks travel in PREFILL_CHUNK-row messages.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

OP_STOP, OP_DECODE, OP_PREFILL, OP_RESET = 0, 1, 2, 4
OP_TREE, OP_CHUNK, OP_COMMIT = OP_DECODE, OP_PREFILL, 3          # legacy names
COMMIT_CONT = 5                                                   # word of the commit message that carries `cont`


def split_layers(num_layers: int, world: int):
    """Contiguous, as even as possible: [(lo, hi)] per rank."""
    base, rem = divmod(num_layers, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


class PipelineComm:
    """Control + activation plumbing, independent of what a stage computes (CPU-testable with gloo).
    host_staging: point-to-point payloads go through pinned host buffers (gloo moves CPU tensors only); used by the
    two-process test that runs real HIP stages on one GPU.  With RCCL the device buffers are sent as they are."""

    def __init__(self, rank: int, world: int, device, hidden: int, dtype, max_tokens: int, max_path: int,
                 host_staging: bool = False):
        self.rank, self.world, self.device = rank, world, device
        self.host_staging = host_staging and str(device) != "cpu"
        cdev = "cpu" if self.host_staging else device
        self.ctrl = torch.zeros(8, dtype=torch.int64, device=cdev)
        self.ids = torch.zeros(max_tokens, dtype=torch.int32, device=device)
        self.max_path = max(max_path, 1)
        n_commit = 8 + 2 * self.max_path + 1                  # accept result | path | committed tokens (kept + bonus)
        self.commit = torch.zeros(n_commit, dtype=torch.int32, device=cdev)
        self.commit_dev = torch.zeros(n_commit, dtype=torch.int32, device=device)
        self.h = torch.zeros(max_tokens, hidden, dtype=dtype, device=device)      # only used when no model buffer is given
        if self.host_staging:
            self._h_host = torch.zeros(max_tokens, hidden, dtype=dtype).pin_memory()
            self._ids_host = torch.zeros(max_tokens, dtype=torch.int32).pin_memory()

    @property
    def first(self):
        return self.rank == 0

    @property
    def last(self):
        return self.rank == self.world - 1

    def plan(self, *vals):
        """rank 0: broadcast a plan word; other ranks: receive it.  Returns the list of ints."""
        if self.first:
            self.ctrl.zero_()
            self.ctrl[:len(vals)] = torch.tensor(vals, dtype=torch.int64)
        dist.broadcast(self.ctrl, src=0)
        return self.ctrl.tolist()

    command = plan                                                # legacy name

    def bcast(self, t: torch.Tensor):
        """rank 0's `t` -> every rank's `t` (device tensor; host staged under gloo)"""
        if self.world == 1:
            return t
        if self.host_staging:
            h = t.cpu() if self.first else torch.empty(t.shape, dtype=t.dtype)
            dist.broadcast(h, src=0)
            if not self.first:
                t.copy_(h)
        else:
            dist.broadcast(t, src=0)
        return t

    def _recv(self, buf, host, src):
        if self.host_staging:
            dist.recv(host[:buf.shape[0]], src=src)
            # blocking: the single pinned staging buffer is overwritten by the next recv (a multi-chunk prompt on the last
            # stage has no other synchronisation point between two chunks)
            buf.copy_(host[:buf.shape[0]], non_blocking=False)
        else:
            dist.recv(buf, src=src)

    def _send(self, buf, host, dst):
        if self.host_staging:
            host[:buf.shape[0]].copy_(buf)                        # synchronises with the producing kernels
            dist.send(host[:buf.shape[0]], dst=dst)
        else:
            dist.send(buf, dst=dst)

    def recv_activations(self, T, into=None):
        """ranks > 0: the previous stage's [T, H] rows, received straight into `into` (the stage model's hidden buffer)."""
        buf = (self.h if into is None else into)[:T]
        if not self.first:
            self._recv(buf, getattr(self, "_h_host", None), self.rank - 1)
        return buf

    def send_activations(self, h):
        if not self.last:
            self._send(h if h.is_contiguous() else h.contiguous(), getattr(self, "_h_host", None), self.rank + 1)

    def return_ids(self, ids=None, n=1):
        """last stage -> rank 0 (no-op when they coincide)."""
        if self.world == 1:
            return ids
        if self.last:
            self._send(ids[:n].contiguous(), getattr(self, "_ids_host", None), 0)
            return ids
        if self.first:
            self._recv(self.ids[:n], getattr(self, "_ids_host", None), self.world - 1)
            return self.ids[:n]
        return None

    def share_commit(self, res=None, path=None, cont=1, newtok=None):
        """rank 0: accept result (int32[8]) + path + the tokens this iteration committed (positions n_old .. n_new: the
        kept nodes and the bonus token) + continue flag -> every stage.  Returns (res, path, newtok, cont): views of the
        device copy (what the compaction kernel reads)."""
        mp = self.max_path
        if self.first:
            self.commit_dev[:8] = res[:8]
            self.commit_dev[8:8 + mp].zero_()
            self.commit_dev[8:8 + min(mp, path.numel())] = path[:mp]
            if newtok is not None:
                self.commit_dev[8 + mp:8 + mp + newtok.numel()] = newtok
            self.commit_dev[COMMIT_CONT] = cont
            if self.host_staging:
                self.commit.copy_(self.commit_dev)
        src = self.commit if self.host_staging else self.commit_dev
        if self.world > 1:
            dist.broadcast(src, src=0)
        if self.host_staging and not self.first:
            self.commit_dev.copy_(self.commit)
        if self.first:
            flag = cont
        else:
            self.commit_host = src[:8].tolist()       # ONE read-back per iteration on a stage rank: kept, n_new, cont ...
            flag = self.commit_host[COMMIT_CONT]
        return self.commit_dev[:8], self.commit_dev[8:8 + mp], self.commit_dev[8 + mp:], flag


class PipelinedTarget:
    """Drop-in for the engine's ``target_model`` on rank 0: every forward runs stage 0 locally and drives the
    other stages through PipelineComm; ``sampled`` ids come back from the last stage."""

    def __init__(self, stage_model, comm: PipelineComm, engine):
        self.m, self.comm, self.eng = stage_model, comm, engine
        self.config, self.max_length, self.eos_tokens = stage_model.config, stage_model.max_length, stage_model.eos_tokens
        self.kv_cache, self.CHUNK, self.num_layers = stage_model.kv_cache, stage_model.CHUNK, stage_model.num_layers
        self.PREFILL_CHUNK = stage_model.PREFILL_CHUNK
        self._off = None
        self._tree_graph = None

    def reserve(self, tokens, logit_rows=None):
        self.m.reserve(tokens, logit_rows)

    @property
    def logits_buffer(self):                          # single-stage group: the head is on this rank
        return self.m.logits_buffer

    def clear(self):
        self.eng._leave_decode()
        self.comm.plan(OP_RESET)
        self.m.clear()

    def weight_bytes(self):
        return self.m.weight_bytes()

    def _stage0_tree(self, tokens_all, n_dev, depth, T, mask_bits, mask_words):
        """stage 0's share of the verify, replayed as a hipGraph (all run-time state is read from device memory)."""
        run = lambda: self.m.forward_tree(tokens_all, n_dev, depth, 0, T, mask_bits, mask_words, head_from=0)
        if not self.eng.use_graph:
            return run()
        if self._tree_graph is None:
            self._tree_graph = _capture(run, self.m.device)
        self._tree_graph.replay()

    # tree-mode verify: returns sampled ids [T] on rank 0
    def verify_tree(self, tokens_all, n_dev, depth, T, mask_bits, mask_words):
        self._stage0_tree(tokens_all, n_dev, depth, T, mask_bits, mask_words)
        self.comm.send_activations(self.m.hidden_buffer[:T])
        return self.comm.return_ids(n=T)

    # causal chunks (prefill / append): returns the arg-max id of the last row (int32[1]) on rank 0
    def prefill_tokens(self, ids, start, want_logits=True):
        P = ids.shape[0]
        out = None
        self.eng._leave_decode()
        chunk = prefill_chunk(self.m)
        self.comm.plan(OP_PREFILL, P, start, int(want_logits), chunk)
        self.comm.bcast(ids.contiguous())                   # the last stage samples over the token history
        for lo in range(0, P, chunk):
            hi = min(P, lo + chunk)
            T = hi - lo
            last = hi == P and want_logits
            pos = torch.arange(start + lo, start + hi, dtype=torch.int32, device=self.m.device)
            pre = torch.tensor([start + lo], dtype=torch.int32, device=self.m.device)
            local_head = last and self.m.is_last            # single-stage group: the head is here
            self.m.forward_explicit(ids[lo:hi].contiguous(), pos, pos, pre, head_from=(T - 1 if local_head else T))
            self.comm.send_activations(self.m.hidden_buffer[:T])
            if local_head:
                out = self.eng._first_token(self.m.logits_buffer[0]).clone()
            elif last:
                out = self.comm.return_ids(n=1)
        self.kv_cache.kv_offset = start + P
        return out


def prefill_chunk(model) -> int:
    """rows per prompt message: the stage model's prompt chunk (1024 when the workspace allows)"""
    return max(model.CHUNK, min(model.PREFILL_CHUNK, model.ws_tokens))


def _capture(run, device):
    s = torch.cuda.Stream(device=device)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run()                                              # warm-up outside capture
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    torch.cuda.synchronize()
    return g


def _f2i(x: float) -> int:
    import struct
    return struct.unpack("<i", struct.pack("<f", float(x)))[0]


def _i2f(i: int) -> float:
    import struct
    return struct.unpack("<f", struct.pack("<i", int(i)))[0]


def stage_worker(model, comm: PipelineComm, tables, mask_first_eos=False, use_graph=True):
    """Event loop of ranks > 0.  `tables` = dict(depth, mask_bits, mask_words, n_dev, eos_dev, n_eos, max_path, tree_size,
    dynamic).  The last stage turns its logits into one token id per tree node -- arg-max, or umb_sample_rows (repetition
    penalty over its own copy of the token history, top-k / top-p, temperature, a draw keyed by (seed, num_nodes, row):
    the same kernel, arguments and seed as the single-GPU engine, so the ids are the same) -- and returns T ints."""
    from . import _lib
    dev = model.device
    T = tables["tree_size"]
    sampled = torch.zeros(max(comm.ids.shape[0], T), dtype=torch.int32, device=dev)
    tokens = torch.zeros(model.max_length + T + 8, dtype=torch.int32, device=dev)        # the committed token history
    n_dev = tables["n_dev"]
    V = model.config.vocab_size
    rng_state = torch.zeros(1, dtype=torch.int64, device=dev)
    knobs = dict(greedy=1, temperature=0.0, topp=0.9, penalty=1.0, topk=32)
    tree_graph, graph_knobs = None, None

    def tree_forward():
        model.forward_tree(tokens, n_dev, tables["depth"], 0, T, tables["mask_bits"], tables["mask_words"], head_from=0)
        if comm.last:
            if knobs["greedy"]:
                _lib.call("umb_argmax_rows", sampled, model.logits_buffer, T, V)
            else:
                _lib.call("umb_sample_rows", sampled, model.logits_buffer, T, V, tokens, n_dev, float(knobs["penalty"]),
                          float(knobs["temperature"]), min(int(knobs["topk"]), V), float(knobs["topp"]), rng_state, 0, None, None)

    while True:
        c = comm.plan()
        op = c[0]
        if op == OP_STOP:
            return
        if op == OP_RESET:
            model.clear()
            n_dev.zero_()
            tokens.zero_()
        elif op == OP_PREFILL:
            P, start, want, chunk = c[1], c[2], c[3], c[4]
            comm.bcast(tokens[start:start + P])
            for lo in range(0, P, chunk):
                hi = min(P, lo + chunk)
                rows = hi - lo
                last = hi == P and want
                comm.recv_activations(rows, into=model.hidden_buffer)
                pos = torch.arange(start + lo, start + hi, dtype=torch.int32, device=dev)
                pre = torch.tensor([start + lo], dtype=torch.int32, device=dev)
                model.forward_explicit(tokens[start + lo:start + hi], pos, pos, pre, head_from=(rows - 1 if last else rows))
                comm.send_activations(model.hidden_buffer[:rows])
                if comm.last and last:
                    row = model.logits_buffer[0]
                    if mask_first_eos and tables["n_eos"]:
                        _lib.call("umb_mask_eos", row, tables["eos_dev"], tables["n_eos"])
                    _lib.call("umb_argmax_rows", sampled[:1], row, 1, V)
                    tokens[start + P:start + P + 1] = sampled[:1]       # the root of the first tree: part of the history
                    comm.return_ids(sampled, 1)
            n_dev.fill_(start + P)
        elif op == OP_DECODE:
            knobs = dict(greedy=c[1], temperature=_i2f(c[2]), topp=_i2f(c[3]), penalty=_i2f(c[4]), topk=c[5])
            rng_state.fill_(c[6])
            if graph_knobs != tuple(sorted(knobs.items())):             # sampling knobs are launch arguments of the graph
                tree_graph, graph_knobs = None, tuple(sorted(knobs.items()))
            cont = 1
            while cont:
                if tables["dynamic"]:
                    comm.bcast(tables["mask_bits"])                     # this iteration's beam-grown ancestor rows
                comm.recv_activations(T, into=model.hidden_buffer)
                if use_graph:
                    if tree_graph is None:
                        torch.cuda.current_stream().synchronize()
                        keep = model.hidden_buffer[:T].clone()
                        tree_graph = _capture(tree_forward, dev)
                        model.hidden_buffer[:T].copy_(keep)         # the warm-up pass overwrote the received rows
                    tree_graph.replay()
                else:
                    tree_forward()
                comm.send_activations(model.hidden_buffer[:T])
                if comm.last:
                    comm.return_ids(sampled, T)
                res, path, newtok, cont = comm.share_commit()
                model.kv_cache.compact(res, path.contiguous(), tables["max_path"])
                # history: positions n_old .. n_new <- kept tokens + bonus (a no-op commit keeps nothing)
                keep_n, n_new = comm.commit_host[0], comm.commit_host[3]
                if keep_n > 0:
                    tokens[n_new - keep_n:n_new + 1] = newtok[:keep_n + 1]
                n_dev.copy_(res[3:4])


def build_pipelined_engine(device: str, dtype=torch.float16, seed: int = 0, **config):
    """Layer-sharded engine from a reference-style config (``engine: static`` with ``growmap_path`` | ``growmap``, or
    ``engine: dynamic`` with ``width`` / ``num_beams`` / ``depth`` -- the reference's 70B engine; ``model``,
    ``draft_model``, ``max_length``, sampling knobs ...): every rank of the default process group holds a contiguous
    slice of the target's layers (`split_layers`); rank 0 also holds the draft and gets the engine back, the other
    ranks serve forwards inside this call until rank 0 sends OP_STOP (`shutdown_pipeline`) and then return None.
    Launch with torch.distributed.run, one process per GPU; the process group must already be initialised."""
    import json as _json

    from .models.config import KNOWN, LlamaCfg
    from .models.llama import Llama, pack_mask_bits
    from .speculation.static_speculation_engine import resolve_growmap_path
    kind = config.pop("engine", "static")
    assert kind in ("static", "dynamic"), kind
    rank, world = dist.get_rank(), dist.get_world_size()
    target, draft = config.pop("model"), config.pop("draft_model")
    max_length = config.get("max_length", 8192)
    cfg = KNOWN[target] if target in KNOWN else LlamaCfg.from_dir(target)
    lo, hi = split_layers(cfg.num_hidden_layers, world)[rank]
    # `seed` is the engine's (sampling) seed, as in the single-GPU engines; synthetic weights are drawn from
    # `weights_seed` (default 0 = what AutoModelLM.from_pretrained uses), so a sharded run sees the single-GPU model
    stage = Llama(target, max_length=max_length, device=device, dtype=dtype, seed=config.pop("weights_seed", 0))
    stage.alloc(layer_range=(lo, hi))
    if kind == "static":
        gm = config.pop("growmap", None)
        if gm is None:
            with open(resolve_growmap_path(config.pop("growmap_path"))) as f:
                gm = _json.load(f)
        else:
            config.pop("growmap_path", None)
        T, max_path = gm["size"], len(gm["roots"])
        depth = torch.tensor(gm["depth"], dtype=torch.int32, device=device)
        mask_bits = pack_mask_bits((torch.tensor(gm["mask"]) == 1).to(device)).contiguous()
    else:
        W, Dp = config.get("width", 16), config.get("depth", 24)
        T, max_path = W * Dp + 1, Dp + 1
        depth = torch.tensor([0] + [i + 1 for i in range(Dp) for _ in range(W)], dtype=torch.int32, device=device)
        mask_bits = torch.zeros(T, (T + 63) // 64, dtype=torch.int64, device=device)
    stage.reserve(max(stage.PREFILL_CHUNK, T), logit_rows=max(stage.CHUNK, T))
    host_staging = dist.get_backend() == "gloo"
    comm = PipelineComm(rank, world, device, cfg.hidden_size, dtype, max(stage.PREFILL_CHUNK, T), max_path,
                        host_staging=host_staging)
    if rank != 0:
        tables = dict(depth=depth, mask_bits=mask_bits, n_dev=torch.zeros(1, dtype=torch.int32, device=device),
                      eos_dev=torch.tensor(list(cfg.eos_token_id), dtype=torch.int32, device=device),
                      n_eos=len(cfg.eos_token_id), max_path=max_path, tree_size=T, dynamic=kind == "dynamic")
        tables["mask_words"] = mask_bits.shape[1]
        stage_worker(stage, comm, tables, mask_first_eos=kind == "dynamic", use_graph=config.get("hip_graph", True))
        return None
    for k in ("offload", "cuda_graph", "num_cache_layers"):      # single-GPU placement knobs of the reference
        config.pop(k, None)
    if kind == "static":
        eng = PipelinedStaticEngine(draft, target, dtype=dtype, device=device, growmap=gm, seed=seed,
                                    stage_model=stage, comm=comm, **config)
    else:
        eng = PipelinedDynamicEngine(draft, target, dtype=dtype, device=device, seed=seed, offload=False,
                                     stage_model=stage, comm=comm, **config)
    eng.initialize()
    return eng


def shutdown_pipeline(eng):
    """Rank 0: release the stage workers blocked in build_pipelined_engine."""
    eng._leave_decode()
    eng._comm.plan(OP_STOP)


def hop_probe(device, nbytes, reps=30):
    """One isolated hop of the verify pipeline: a [T, H] 16-bit activation sent rank 0 -> rank 1 and back, `reps` round
    trips behind one warm-up; -> microseconds per one-way hop (None in a 1-rank group).  Collective over ranks 0 and 1."""
    import time
    world, rank = dist.get_world_size(), dist.get_rank()
    if world < 2:
        return None
    staged = dist.get_backend() == "gloo"
    buf = torch.zeros(max(nbytes // 2, 1), dtype=torch.float16, device="cpu" if staged else device)
    dist.barrier()
    if rank > 1:
        dist.barrier()
        return None
    us = None
    for timed in (False, True):
        if not staged:
            torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(reps if timed else 3):
            if rank == 0:
                dist.send(buf, 1); dist.recv(buf, 1)
            else:
                dist.recv(buf, 0); dist.send(buf, 0)
        if not staged:
            torch.cuda.synchronize()
        us = (time.time() - t0) / reps / 2 * 1e6
    dist.barrier()
    return round(us, 2)


def pp_measure(args, wl, dtype, device, rank, world):
    """One request, the target's layers sharded over the ranks of the default process group (static 3x4, greedy; BASELINE
    config 5): rank 0 drives and returns the result dict, the other ranks serve stage forwards inside this call and
    return None.  Collective: every rank of the group must call it.  Nothing is printed, the group stays up.
    The acceptance knob is the headline's (same acc vector, same seed: speculation/steering.py), so `ms_per_step` /
    `accept_len` compare 1 : 1 with the N = 1 line; `hop_us` = one isolated [T, H] send/recv hop, `devices` = the
    physical device identities gathered over the group."""
    import __graft_entry__ as ge
    from .models.config import KNOWN
    from .sequoia_utils import DEFAULT_ACC, generate_sequoia_tree
    from .speculation.steering import device_census, steered_measure
    ge.build()
    census = device_census(device, dist)
    gm = generate_sequoia_tree(3, 4)
    hop_bytes = gm["size"] * KNOWN[wl["target"]].hidden_size * 2
    hop_us = hop_probe(device, hop_bytes)
    eng = build_pipelined_engine(device, dtype=dtype, seed=args.seed, model=wl["target"], draft_model=wl["draft"],
                                 growmap=gm, max_length=args.max_length, exit_layer=16)
    if eng is None:
        return None
    g = torch.Generator().manual_seed(1234)
    prompt = torch.randint(3, wl.get("vocab_hi", 128000), (1, args.prompt_len), generator=g)
    acc = list(DEFAULT_ACC)
    r = steered_measure(eng, prompt, acc, args.seed, args.warmup, args.steps, len(gm["roots"]))
    layers = [hi - lo for lo, hi in split_layers(eng._stage_model.config.num_hidden_layers, world)]
    shutdown_pipeline(eng)
    r.update({"n_ranks_rccl": world, "backend": dist.get_backend(), "devices": census["devices"],
              "n_distinct_devices": census["n_distinct"], "layers_per_rank": layers, "hops_per_verify": world - 1,
              "hop_bytes": hop_bytes, "hop_us": hop_us, "acc": acc,
              "parallelism": f"pp{world}: target layers sharded, send/recv of the [T, H] activation per hop, one commit "
                             "broadcast per iteration; draft + engine state on rank 0", "tree": "3x4", "scaling": "strong"})
    return r


def run_pp_bench(args, wl, dtype, device, rank, world):
    """bench.py --parallel pp: the layer-sharded engine alone, one JSON line from rank 0."""
    import json
    r = pp_measure(args, wl, dtype, device, rank, world)
    out = None
    if r is not None:
        out = {"metric": "tokens/s @ bs=1 (speculative decoding)", "value": r["tokens_per_s"], "unit": "tokens/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": wl["dtype"],
               "data": "synthetic: random-init weights, random prompt; acceptance set by the controllable-acceptance draft "
                       "(the headline's acc vector and seed)",
               "config": {"workload": wl["desc"], "parallelism": r["parallelism"], "tree": "3x4", "prompt_len": args.prompt_len},
               "accept_len": r["accept_len"], "value_raw_draft": r["tokens_per_s_raw_draft"], "pp": r}
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return out


from .speculation.dynamic_speculation_engine import DynamicSpeculationEngine as _Dynamic  # noqa: E402
from .speculation.static_speculation_engine import StaticSpeculationEngine as _Static  # noqa: E402


class _PipelinedMixin:
    """An engine whose target is a PipelinedTarget (rank 0 of a layer-sharded group).  The draft tree replays as a
    hipGraph, stage 0's share of the verify as another; hops, accept scan and the commit broadcast are eager.  The last
    stage samples (greedy or stochastic) and returns T token ids instead of [T, V] logits."""
    DYNAMIC_MASK = False

    def _pp_init(self, stage_model, comm):
        # the static engine's reference draw (ONE frozen rand(3, T), static:131,310) is a single-GPU mode: here the LAST stage
        # samples, with its own counter-based draws (umb_sample_rows, same seed / knobs on every stage).  An explicit
        # uniform_samples tensor cannot be honoured and is refused; the default simply switches to the fresh-draw sampler.
        assert getattr(self, "_uniform_arg", None) is None, \
            "uniform_samples (the reference's frozen uniforms) is a single-GPU static-engine mode: the last pipeline stage draws its own"
        self.reference_sampler = False
        self._stage_model, self._comm = stage_model, comm
        self._in_decode, self._pending = False, False
        self._decode_knobs = None

    def initialize(self):
        self._target_model = PipelinedTarget(self._stage_model, self._comm, self)
        super().initialize()
        self.graph_scope = "draft"                   # cross-rank hops cannot live inside the graph

    def _knobs(self):
        return (int(self._greedy()), _f2i(self.temperature), _f2i(self.topp), _f2i(self.repetition_penalty),
                int(self.topk), int(self.seed))

    def update_generation_args(self, **generation_args):
        super().update_generation_args(**generation_args)
        if self._in_decode and self._knobs() != self._decode_knobs:
            self._leave_decode()                     # the stages learn the new knobs with the next OP_DECODE plan

    def manual_seed(self, seed: int):
        super().manual_seed(seed)
        if self._in_decode:
            self._leave_decode()

    # ---- decode-mode bookkeeping: the commit of iteration i travels when rank 0 knows what follows it
    def _commit_tokens(self):
        """the tokens iteration i committed: positions n_old .. n_new (kept nodes + bonus), read after the accept scan"""
        keep, n_new = int(self.res_host[0]), int(self.res_host[3])
        return self.tokens[n_new - keep:n_new + 1] if keep > 0 else self.tokens[:0]

    def _flush_commit(self, cont):
        if self._pending:
            torch.cuda.current_stream().synchronize()            # res_host holds this iteration's accept result
            self._comm.share_commit(self.res, self.path, cont=cont, newtok=self._commit_tokens())
            self._pending = False

    def _enter_decode(self):
        if not self._in_decode:
            self._decode_knobs = self._knobs()
            self._comm.plan(OP_DECODE, *self._decode_knobs)
            self._in_decode = True
        else:
            self._flush_commit(cont=1)

    def _leave_decode(self):
        if self._in_decode:
            if not self._pending:
                # decode mode was entered but the iteration died before its commit was formed (a kernel or transport
                # error inside verify_tree): the stages still wait for one.  Release them with a no-op commit -- keep 0
                # tokens, num_nodes unchanged, cont = 0 -- so reset() / a new prompt / shutdown work afterwards.
                self.res.zero_()
                self.res[3] = int(self.num_nodes)
                self.res_host.copy_(self.res)
                self._pending = True
            self._flush_commit(cont=0)
            self._in_decode = False

    def reset(self):
        self._leave_decode()
        super().reset()

    def _feed(self, lo, hi):
        ids = self.tokens[lo:hi]
        dlo = lo - 1 if (self.lookback and lo > 0) else lo          # see HipEngine._feed
        self.draft_model.prefill_tokens(self.tokens[dlo:hi], dlo, want_logits=False)
        first = self.target_model.prefill_tokens(ids, lo, want_logits=True)
        self.tokens[hi:hi + 1] = first
        self.num_nodes = hi
        self.n_dev.fill_(hi)
        self.last_bonus = None

    def _verify_forward(self):
        self._enter_decode()
        if self.DYNAMIC_MASK:
            self._comm.bcast(self.mask_bits)                         # the stages' copy of this iteration's ancestor rows
        ids = self.target_model.verify_tree(self.tokens, self.n_dev, self.depth, self.tree_size, self.mask_bits,
                                            self.mask_words)
        self._remote_sampled = ids

    def _iteration_tail(self):
        from . import _lib
        self._verify_forward()
        if self._comm.world > 1:
            self.sampled.copy_(self._remote_sampled)
        else:
            self._sample()                                           # single-stage group: the logits are here
        _lib.call("umb_accept_scan", self.sampled, self.parents, self.tokens, self.n_dev, self.tree_size,
                  self.eos_dev, len(self.eos_tokens), self.res, self.path)
        self._pending = True                          # broadcast with the next iteration's `cont` (or on leaving)
        self.draft_model.kv_cache.compact(self.res, self.path, self.max_path)
        self._stage_model.kv_cache.compact(self.res, self.path, self.max_path)
        st = getattr(self.draft_model, "chain_status_word", None)
        if st is not None:                            # the persistent chain's give-up word (HipEngine._commit / _chain_fallback)
            self.res[7:8].copy_(st)
        self.res_host.copy_(self.res, non_blocking=True)

    def verify(self):
        raise NotImplementedError("use step() on the pipelined engine")


class PipelinedStaticEngine(_PipelinedMixin, _Static):
    def __init__(self, *a, stage_model=None, comm=None, **kw):
        super().__init__(*a, **kw)
        self._pp_init(stage_model, comm)


class PipelinedDynamicEngine(_PipelinedMixin, _Dynamic):
    """The reference's 70B engine (dynamic_speculation_engine.py:215-327: beam-grown tree, stochastic verification)
    over a layer-sharded target: the beam expansion stays on rank 0 with the draft; each iteration's ancestor-mask rows
    are broadcast to the stages ahead of the activations."""
    DYNAMIC_MASK = True

    def __init__(self, *a, stage_model=None, comm=None, **kw):
        super().__init__(*a, **kw)
        self._pp_init(stage_model, comm)
