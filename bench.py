#!/usr/bin/env python
"""Headline benchmark: tokens/s at batch 1, speculative decoding on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one speculative iteration of the hot path (draft-tree expand + target verify +
accept scan + KV compaction) on synthetic input.  Default workload = the configuration
BASELINE.json's metric is quoted on that fits one GPU: Llama-3.1-70B-Instruct-AWQ-INT4 target +
Llama-3.2-1B draft, static Sequoia 3x4 tree (T = 13), greedy, max_length 2048 -- the reference's
configs/greedy_config_48gb.json pairing -- with random-init weights of the exact shapes
(no checkpoints offline) and a random 128-token prompt already resident in HBM.

Random draft/target pairs accept ~0 drafted tokens, so acceptance is an INPUT here: the
controllable-acceptance draft (StaticSpeculationEngine.set_oracle_draft) places the target's
own greedy token at the rank-r child with probability acc[r] (default: the reference's
DEFAULT_ACC vector, umbrella/sequoia_utils.py:7).  Every draft / verify kernel still runs;
`accept_len` is reported next to the value, and `value_raw_draft` is the same loop with the
knob off (accept ~1.0).

N > 1 (launched by torch.distributed.run, one rank per GPU): independent requests, one engine
replica per GPU, no data-path collective ("scaling": "weak"); the timed region is bracketed by
barriers and the max over ranks is used.  `--parallel pp` instead shards the target's layers
across the ranks (RCCL send/recv of the [T, H] activations, BASELINE config 5).
"""
from __future__ import annotations

import argparse
import json
import os

os.environ.setdefault("UMBRELLA_SYNTHETIC", "1")   # benchmarks run on seeded random weights of the exact shapes (no checkpoints offline)
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "70b-awq+1b": dict(target="hugging-quants/Meta-Llama-3.1-70B-Instruct-AWQ-INT4",
                       draft="meta-llama/Llama-3.2-1B-Instruct", dtype="fp16", tree="3x4",
                       desc="Llama-3.1-70B-Instruct-AWQ-INT4 target + Llama-3.2-1B draft, static Sequoia 3x4 (T=13), greedy, "
                            "on-device (configs/greedy_config_48gb.json pairing)"),
    "8b+1b": dict(target="meta-llama/Llama-3.1-8B-Instruct", draft="meta-llama/Llama-3.2-1B-Instruct", dtype="bf16",
                  tree="5x6", desc="Llama-3.1-8B-Instruct target + Llama-3.2-1B draft, bf16, static Sequoia 5x6 (T=31), greedy"),
    "1b+1b": dict(target="meta-llama/Llama-3.2-1B-Instruct", draft="meta-llama/Llama-3.2-1B-Instruct", dtype="fp16",
                  tree="3x4", desc="Llama-3.2-1B target + Llama-3.2-1B draft, fp16, static Sequoia 3x4, greedy"),
}
ACC_5x6 = [0.5, 0.2, 0.12, 0.08, 0.05, 0.03]
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    return rank, world, local


def build_engine(wl, device, dtype, max_length, seed, pp=None):
    from umbrella_amd.sequoia_utils import DEFAULT_ACC, generate_sequoia_tree
    from umbrella_amd.speculation.static_speculation_engine import StaticSpeculationEngine
    if wl["tree"] == "3x4":
        gm, acc = generate_sequoia_tree(3, 4), DEFAULT_ACC
    elif wl["tree"] == "mi355x-T16d3":
        # growmap re-tuned for the measured MI355X cost ratios (scripts/tune_growmap.py, profiles/r02_growmap_tuning_*.json):
        # the 15 most probable nodes of depth <= 3 under the same acceptance vector -- one draft forward fewer, a full
        # 16-row token tile in the verify
        import json as _json
        with open(os.path.join(ROOT, "umbrella_amd", "trees", "mi355x_70b_awq_1b-T16d3.json")) as f:
            gm, acc = _json.load(f), DEFAULT_ACC
    else:
        gm, acc = generate_sequoia_tree(5, 6, ACC_5x6), ACC_5x6
    eng = StaticSpeculationEngine(wl["draft"], wl["target"], dtype=dtype, device=device, growmap=gm,
                                  max_length=max_length, exit_layer=16, seed=seed)
    eng.initialize()
    return eng, gm, acc


def run_steps(eng, steps):
    toks0 = eng.num_nodes
    for _ in range(steps):
        eng.step()
    return eng.num_nodes - toks0


def kernel_roofline(eng, reps=40):
    """Live HIP-event timing of the target's four layer linears at the verify shape, each through the kernel (and split
    count / epilogue) the model's schedule actually launches, rotating over the layers so weights come from HBM, not
    the 256 MB Infinity Cache.  Events are recorded on the stream the kernels are launched on (torch's current stream)."""
    from umbrella_amd import _lib
    from umbrella_amd.models.llama import ll_plan, to_fm
    m = eng.target_model
    T = eng.tree_size
    ll = getattr(m, "sched", "") == "ll" and not m.fused and T <= 64
    dt = _lib.dtype_code(m.dtype)
    res = {}
    L = m.num_layers
    lib = _lib.load()
    for key in ("qkv", "o", "gu", "down"):
        lin0 = m.layers[0][key]
        N, K = lin0.N, lin0.K
        x = torch.randn(T, K, device=m.device).to(m.dtype)
        xfm = to_fm(x)
        part = m._bufs["partial"]
        per_launch = (N * K // 2 + (N // 16) * (K // 128) * 64) if lin0.awq else N * K * 2
        h = torch.zeros(T, N, dtype=m.dtype, device=m.device)
        hw = torch.zeros(64 * N, dtype=m.dtype, device=m.device)
        nw = torch.ones(N, dtype=m.dtype, device=m.device)
        ssq = torch.ones(T, max(N // 16, K // 16, 4), dtype=torch.float32, device=m.device)
        act = torch.zeros(64 * max(N // 2, 8), dtype=m.dtype, device=m.device)
        info = {"family": "split-K", "R": lin0.R, "S": lin0.S}
        if ll and key == "gu" and lin0.S == 1 and N // (64 * lin0.R) >= 256:
            # low-latency schedule, large-N gate/up: LDS-shared kernel, FM in / FM out, 1/rms from the sums of squares
            fs = _lib.UmbGemmFused()
            fs.ssq_in, fs.ssq_groups, fs.pad0, fs.ssq_dim, fs.eps, fs.pad1 = ssq.data_ptr(), K // 32, ssq.shape[1], float(K), 1e-5, 3
            launch = lambda ln: _lib.call("umb_gemm_fused", act, xfm, K, ln.w, ln.meta, T, N, K, ln.awq, 1, ln.R, 2, fs, dt)
            info = {"family": "split-K kernel (S=1, FM buffers)", "R": lin0.R, "S": 1}
        elif ll:
            R, WN, WK, NW = ll_plan(N, K, bool(lin0.awq))
            fx = _lib.UmbGemmLL()
            if key in ("o", "down"):
                fx.h, fx.hw, fx.norm_w, fx.ssq_out, fx.ssq_out_stride = h.data_ptr(), hw.data_ptr(), nw.data_ptr(), ssq.data_ptr(), ssq.shape[1]
                epi, out = 4, None
            elif key == "gu":
                fx.ssq_in, fx.ssq_groups, fx.ssq_in_stride, fx.ssq_dim, fx.eps = ssq.data_ptr(), K // 32, ssq.shape[1], float(K), 1e-5
                epi, out = 2, act
            else:
                epi, out = 0, part                      # qkv: the GEMM itself (RoPE / KV append epilogue writes only kilobytes)
            launch = lambda ln: _lib.call("umb_gemm_ll", out, xfm, ln.w, ln.meta, T, N, K, ln.awq, epi, fx, dt)
            info = {"family": "low-latency", "R": R, "WN": WN, "WK": WK}
        else:
            S = lin0.S
            if key in ("o", "down"):                    # model.hip eff_s(): split count of the row-reduced linears
                cap = max(K // 1792, 4)
                if S > cap and (N // (64 * max(lin0.R, 1))) * cap >= 256:
                    S = cap
            epi = 2 if key == "gu" else 0
            out = act if key == "gu" else part
            launch = lambda ln: _lib.call("umb_gemm", out, x, x.stride(0), ln.w, ln.meta, T, N, K, ln.awq, S, ln.R, epi, dt)
            info = {"family": "split-K", "R": lin0.R, "S": S}
        for i in range(4):
            launch(m.layers[i % L][key])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(reps):
            launch(m.layers[(i + 4) % L][key])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000.0 / reps
        res[key] = dict(N=N, K=K, bytes=per_launch, us=us, gbs=per_launch / us / 1e3, **info)
    return res


def _oracle_state(cfg, layers, alias):
    """fp32 random state dict for the oracle; with alias=True the `layers` decoder layers share one set of tensors
    (full-depth arithmetic and memory traffic -- a 70B layer is 3.4 GB in fp32, far beyond any cache -- at one
    layer's RAM)."""
    from umbrella_amd.models.synthetic import linear_shapes

    def rnd(n, k):      # timing only: a materialised rank-1 random matrix costs one pass over memory instead of a Gaussian draw per element
        return (torch.randn(n, 1) * 0.14) * (torch.randn(1, k) * 0.14)
    sd = {"model.embed_tokens.weight": rnd(cfg.vocab_size, cfg.hidden_size),
          "model.norm.weight": torch.ones(cfg.hidden_size)}
    if not cfg.tie_word_embeddings:
        sd["lm_head.weight"] = rnd(cfg.vocab_size, cfg.hidden_size)
    first = {}
    for i in range(layers):
        p = f"model.layers.{i}."
        for ln, (n, k) in linear_shapes(cfg).items():
            key = ln + ".weight"
            if not alias or i == 0:
                first[key] = rnd(n, k)
            sd[p + key] = first[key]
        sd[p + "input_layernorm.weight"] = torch.ones(cfg.hidden_size)
        sd[p + "post_attention_layernorm.weight"] = torch.ones(cfg.hidden_size)
    return sd


def cpu_baseline(wl, gm, accept_len, threads, budget_s=30.0):
    """Oracle ("port": torch CPU fp32, AWQ dequantised at load as a CPU port would) MEASURED end to end on the host
    cores on a bounded sample of the same workload: one whole static-tree iteration -- every draft forward of the
    iteration on the full 16-layer draft and the T-row verify through ALL target layers + lm_head -- at a 128-token
    context.  The 80 target layers alias one layer's tensors (RAM: a 70B model is 280 GB in fp32), which changes neither
    the arithmetic nor the memory traffic (3.4 GB per layer, no cache holds it).  tokens/s = accept_len / iteration."""
    import copy
    from oracle.model import OracleLlama
    from umbrella_amd.models.config import KNOWN, rope_inv_freq
    torch.set_num_threads(threads)
    T = gm["size"]
    P = 128

    def build(name, alias):
        cfg = copy.copy(KNOWN[name])
        sd = _oracle_state(cfg, cfg.num_hidden_layers, alias)
        inv, sc = rope_inv_freq(cfg)
        return OracleLlama(cfg, sd, inv, sc, max_length=256, dtype=torch.float32)

    def forward(m, rows):
        ids = torch.randint(3, 1000, (1, rows))
        pos = torch.arange(P, P + rows)[None]
        mk = torch.ones(rows, 256, dtype=torch.bool)
        m.kv_cache.kv_offset = P
        t0 = time.time()
        m.inference(ids, pos, mk, torch.arange(P, P + rows))
        return time.time() - t0

    widths = [len(x) for x in gm["roots"]]
    draft = build(wl["draft"], alias=False)
    forward(draft, 1)                                      # warm the thread pool / allocator
    t_draft = sum(forward(draft, w) for w in widths)       # the reference schedule: one draft forward per level
    del draft
    target = build(wl["target"], alias=True)
    t_target = forward(target, T)
    layers = target.num_layers
    del target
    it = t_draft + t_target
    return {"value": round(accept_len / it, 4), "unit": "tokens/s", "cores": threads, "kind": "port",
            "iteration_s": round(it, 3), "draft_s": round(t_draft, 3), "verify_s": round(t_target, 3),
            "sample": f"1 whole static iteration measured end to end with the oracle (torch CPU fp32): {len(widths)} draft "
                      f"forwards (rows {widths}) on the full draft + the {T}-row verify through all {layers} target layers "
                      f"and the lm_head, context 128; target layers alias one layer's fp32 tensors (RAM), same arithmetic "
                      f"and traffic; tokens/s at the GPU run's accept_len {accept_len:.2f}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="70b-awq+1b", choices=sorted(WORKLOADS))
    ap.add_argument("--prompt-len", type=int, default=128)
    ap.add_argument("--max-length", type=int, default=2048)
    ap.add_argument("--tree", default=None, choices=[None, "3x4", "5x6", "mi355x-T16d3"],
                    help="growmap override (default: the workload's reference tree); mi355x-T16d3 = the re-tuned tree, a "
                         "second line next to the headline, never the headline itself")
    ap.add_argument("--parallel", default="replicas", choices=["replicas", "pp", "tp"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()

    rank, world, local = dist_env()
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    device = f"cuda:{local}"
    torch.cuda.set_device(local)
    wl = dict(WORKLOADS[args.workload])
    if args.tree:
        wl["tree"] = args.tree
        wl["desc"] = wl["desc"] + f" [growmap override: {args.tree}]"
    dtype = torch.float16 if wl["dtype"] == "fp16" else torch.bfloat16

    if args.parallel == "pp" and world > 1:
        from umbrella_amd.parallel import run_pp_bench
        return run_pp_bench(args, wl, dtype, device, rank, world)
    if args.parallel == "tp":
        if world == 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(device))
        from umbrella_amd.tensor_parallel import run_tp_bench
        return run_tp_bench(args, wl, dtype, device, rank, world)

    import __graft_entry__ as ge
    ge.build()
    eng, gm, acc = build_engine(wl, device, dtype, args.max_length, args.seed + rank)
    g = torch.Generator().manual_seed(1234 + rank)
    prompt = torch.randint(3, 128000, (1, args.prompt_len), generator=g)           # examples/bench.py:31

    # ---- untimed: the target's own greedy continuation (ground truth for the acceptance knob)
    need = (args.warmup + args.steps) * len(gm["roots"]) + 16
    assert eng._prefill(prompt)
    start = eng.num_nodes
    torch.cuda.synchronize()
    t0 = time.time()
    raw_steps = 0
    while eng.num_nodes - start < need and eng.validate_status():
        eng.step()
        raw_steps += 1
    torch.cuda.synchronize()
    raw_dt = time.time() - t0
    raw_tokens = eng.num_nodes - start
    truth = eng.tokens[start:eng.num_nodes + 1].tolist()
    raw_tps = raw_tokens / raw_dt
    raw_accept = raw_tokens / max(raw_steps, 1)

    # A token verified as a depth-d tree node sums its attention in a different order than the same token
    # verified as a root, so on random-init weights a 16-bit near-tie can flip the target's arg-max about
    # once per ~50 tokens and leave the recorded continuation.  The kernels are deterministic, so the
    # continuation is re-recorded under the knob's own execution pattern until it is a fixed point
    # (each pass reproduces the previous one bit-for-bit up to its first divergence).
    passes = 0
    max_passes = max(8, min(40, (args.warmup + args.steps + 15) // 16))   # ~1 near-tie flip per 50 tokens: long runs need more
    for passes in range(1, max_passes + 1):
        eng.reset()
        assert eng._prefill(prompt)
        eng.set_oracle_draft(truth, start, acc, seed=args.seed)
        run_steps(eng, args.warmup + args.steps)
        div = eng.diverged
        while eng.num_nodes - start < need and eng.validate_status():
            eng.step()
        new_truth = eng.tokens[start:eng.num_nodes + 1].tolist()
        if div == 0:
            break
        truth = new_truth

    # ---- timed: acceptance knob on
    eng.reset()
    assert eng._prefill(prompt)
    eng.set_oracle_draft(truth, start, acc, seed=args.seed)
    run_steps(eng, args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    tokens = run_steps(eng, args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.time() - t0
    if world > 1:
        tt = torch.tensor([dt, float(tokens)], dtype=torch.float64, device=device)
        mx = tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        dt, tokens = float(mx[0]), float(tt[1])
    accept_len = tokens / (args.steps * world)
    value = tokens / dt

    out = None
    if rank == 0:
        m, d = eng.target_model, eng.draft_model
        levels = [len(x) for x in gm["roots"]]
        n_mid = start + tokens / world / 2
        kv_t = m.num_layers * 2 * m.config.num_key_value_heads * m.config.head_dim * 2
        kv_d = d.num_layers * 2 * d.config.num_key_value_heads * d.config.head_dim * 2
        # draft forwards actually executed per iteration: one per expanded level (the reference's extra KV-fill
        # forward over the deepest level is folded into the next root forward, engine_common._draft_root), each
        # with its lm_head; without the look-back the reference count (len(levels), last one without lm_head)
        if getattr(eng, "lookback", False):
            n_fwd = len(levels) - 1
            draft_fwd = d.weight_bytes()
        else:
            n_fwd = len(levels)
            draft_fwd = d.weight_bytes() - (d.lm_head.N * d.lm_head.K * 2) / len(levels)
        bytes_iter = n_fwd * draft_fwd + m.weight_bytes() + n_mid * (kv_t + n_fwd * kv_d)
        iter_ms = dt / args.steps * 1e3
        out = {"metric": "tokens/s @ bs=1 (speculative decoding)", "value": round(value, 2), "unit": "tokens/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(iter_ms, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": wl["dtype"],
               "data": "synthetic: random-init weights of the exact shapes, random prompt ids; acceptance set by the "
                       "controllable-acceptance draft (acc vector below), all draft/verify kernels execute",
               "config": {"workload": wl["desc"], "engine": "static", "tree": wl["tree"], "tree_size": eng.tree_size,
                          "prompt_len": args.prompt_len, "max_length": args.max_length, "acc": acc,
                          "parallelism": "1 engine per GPU (replicas)" if world > 1 else "single GPU"},
               "accept_len": round(accept_len, 3), "value_raw_draft": round(raw_tps, 2),
               "accept_len_raw_draft": round(raw_accept, 3), "oracle_draft_divergence": getattr(eng, "diverged", 0), "oracle_draft_passes": passes,
               "schedule": getattr(m, "sched", "split"),
               "draft_forwards_per_iter": n_fwd, "iter_bytes_GB": round(bytes_iter / 1e9, 3),
               "iter_hbm_frac": round(bytes_iter / (iter_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if not args.no_roofline:
            kr = kernel_roofline(eng)
            dom = max(kr.values(), key=lambda r: r["bytes"])
            tot_b, tot_us = sum(r["bytes"] for r in kr.values()), sum(r["us"] for r in kr.values())
            # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the figure is
            # the committed result of separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/README.md); it is
            # only reported while the kernel source it was taken from is unchanged (sha256 of csrc/gemm.hip recorded there)
            traffic, tsrc = None, None
            import hashlib
            with open(os.path.join(ROOT, "umbrella_amd", "csrc", "gemm.hip"), "rb") as f:
                src_hash = hashlib.sha256(f.read()).hexdigest()[:16]
            for pmc_name in ("r02_pmc_gemm70b_traffic.json",):
                pmc = os.path.join(ROOT, "profiles", pmc_name)
                if m.config.awq and dom["N"] == 57344 and dom["K"] == 8192 and os.path.exists(pmc):
                    with open(pmc) as f:
                        rec = json.load(f)
                    if rec.get("gemm_hip_sha256_16") == src_hash:      # a figure taken on other kernel source is not reported
                        traffic = rec["gate_up_traffic_bytes"]
                        tsrc = f"static: profiles/{pmc_name} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on this gemm.hip)"
                        break
            out["roofline"] = {"bound": "hbm", "kernel": f"{dom['family']} {'int4' if m.config.awq else 'dense'} gate_up "
                                                       f"N={dom['N']} K={dom['K']} T={eng.tree_size}",
                               "achieved": round(dom["gbs"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(dom["gbs"] / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": tsrc,
                               "avg_launch_us": round(dom["us"], 2), "bytes_per_launch": dom["bytes"],
                               "layer_gemms": {k: {kk: (round(vv, 2) if isinstance(vv, float) else vv) for kk, vv in v.items()
                                                   if kk not in ("N", "K", "bytes")} for k, v in kr.items()},
                               "layer_gemms_GBs": round(tot_b / tot_us / 1e3, 1)}
        if not args.no_cpu_baseline and world == 1:          # the CPU leg runs on rank 0 at N = 1 only
            try:
                out["cpu_baseline"] = cpu_baseline(wl, gm, accept_len, torch.get_num_threads())
            except Exception as e:                                        # never lose the GPU line to the baseline leg
                out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
