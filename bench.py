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
    """Live HIP-event timing of the dominant kernel: the target's int4 (or bf16) skinny GEMMs at the
    verify shape, rotating over the layers so weights come from HBM, not the 256 MB Infinity Cache."""
    from umbrella_amd import _lib
    m = eng.target_model
    T = eng.tree_size
    res = {}
    x_by_k = {}
    for key in ("qkv", "o", "gu", "down"):
        lin0 = m.layers[0][key]
        x = x_by_k.setdefault(lin0.K, torch.randn(T, lin0.K, device=m.device).to(m.dtype))
        part = m._bufs["partial"]
        per_launch = (lin0.N * lin0.K // 2 + (lin0.N // 16) * (lin0.K // 128) * 64) if lin0.awq else lin0.N * lin0.K * 2
        L = m.num_layers

        def launch(i):
            ln = m.layers[i % L][key]
            _lib.call("umb_gemm", part, x, x.stride(0), ln.w, ln.meta, T, ln.N, ln.K, ln.awq, ln.S, ln.R, 0,
                      _lib.dtype_code(m.dtype))
        for i in range(4):
            launch(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(reps):
            launch(i + 4)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000.0 / reps
        res[key] = dict(N=lin0.N, K=lin0.K, R=lin0.R, S=lin0.S, bytes=per_launch, us=us, gbs=per_launch / us / 1e3)
    return res


def cpu_baseline(wl, gm, accept_len, threads):
    """Oracle ("port") timed on host cores on a bounded sample of the same workload: one static-tree
    iteration with the target truncated to 1 of its layers and the draft to 2 (per-layer cost extrapolated
    linearly; embedding + lm_head timed in full).  AWQ linears are dequantised once at load (what a CPU port
    would do) and all arithmetic is fp32 on the torch CPU backend."""
    import copy
    from oracle.model import OracleLlama
    from umbrella_amd.models.config import KNOWN, rope_inv_freq
    from umbrella_amd.models.synthetic import linear_shapes
    torch.set_num_threads(threads)
    T = gm["size"]

    def time_model(name, layers, rows_list):
        cfg = copy.copy(KNOWN[name])
        full_layers = cfg.num_hidden_layers
        cfg.num_hidden_layers = layers
        sd = {"model.embed_tokens.weight": torch.randn(cfg.vocab_size, cfg.hidden_size) * 0.02,
              "model.norm.weight": torch.ones(cfg.hidden_size)}
        if not cfg.tie_word_embeddings:
            sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
        for i in range(layers):
            p = f"model.layers.{i}."
            for ln, (n, k) in linear_shapes(cfg).items():
                sd[p + ln + ".weight"] = torch.randn(n, k) * 0.02
            sd[p + "input_layernorm.weight"] = torch.ones(cfg.hidden_size)
            sd[p + "post_attention_layernorm.weight"] = torch.ones(cfg.hidden_size)
        inv, sc = rope_inv_freq(cfg)
        m = OracleLlama(cfg, sd, inv, sc, max_length=256, dtype=torch.float32)
        P = 128                                   # prefix already "in the cache" (zeros: timing only)
        total = 0.0
        for rows in rows_list:
            ids = torch.randint(3, 1000, (1, rows))
            pos = torch.arange(P, P + rows)[None]
            mk = torch.ones(rows, 256, dtype=torch.bool)
            best_all, best_head = 1e9, 1e9
            for _ in range(2):                    # second pass = warm caches / thread pool
                m.kv_cache.kv_offset = P
                t0 = time.time()
                m.inference(ids, pos, mk, torch.arange(P, P + rows))
                best_all = min(best_all, time.time() - t0)
                nl, m.num_layers = m.num_layers, 0
                m.kv_cache.kv_offset = P
                t0 = time.time()
                m.inference(ids, pos, mk, torch.arange(P, P + rows))
                best_head = min(best_head, time.time() - t0)
                m.num_layers = nl
            total += (best_all - best_head) / layers * full_layers + best_head
        return total

    widths = [len(x) for x in gm["roots"]]
    t_draft = time_model(wl["draft"], 2, widths)
    t_target = time_model(wl["target"], 1, [T])
    it = t_draft + t_target
    return {"value": round(accept_len / it, 4), "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": f"oracle (torch CPU fp32, AWQ dequantised at load) on 1 static {len(widths)-1}-level iteration: draft "
                      f"truncated to 2 of its layers, target to 1 of its layers, per-layer time extrapolated linearly, "
                      f"embedding + lm_head timed in full; extrapolated iteration {it:.2f} s at the same accept_len {accept_len:.2f}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="70b-awq+1b", choices=sorted(WORKLOADS))
    ap.add_argument("--prompt-len", type=int, default=128)
    ap.add_argument("--max-length", type=int, default=2048)
    ap.add_argument("--parallel", default="replicas", choices=["replicas", "pp"])
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
    wl = WORKLOADS[args.workload]
    dtype = torch.float16 if wl["dtype"] == "fp16" else torch.bfloat16

    if args.parallel == "pp" and world > 1:
        from umbrella_amd.parallel import run_pp_bench
        return run_pp_bench(args, wl, dtype, device, rank, world)

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
               "draft_forwards_per_iter": n_fwd, "iter_bytes_GB": round(bytes_iter / 1e9, 3),
               "iter_hbm_frac": round(bytes_iter / (iter_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if not args.no_roofline:
            kr = kernel_roofline(eng)
            dom = max(kr.values(), key=lambda r: r["bytes"])
            tot_b, tot_us = sum(r["bytes"] for r in kr.values()), sum(r["us"] for r in kr.values())
            traffic, tsrc = None, None
            pmc = os.path.join(ROOT, "profiles", "r01_pmc_gemm70b_traffic.json")
            if m.config.awq and dom["N"] == 57344 and dom["K"] == 8192 and os.path.exists(pmc):
                with open(pmc) as f:                      # PMC pass is a separate rocprofv3 run (see profiles/README.md)
                    traffic, tsrc = json.load(f)["gate_up_traffic_bytes"], "profiles/r01_pmc_gemm70b_traffic.json"
            out["roofline"] = {"bound": "hbm", "kernel": f"skinny_gemm_kernel<{'AWQ' if m.config.awq else 'dense'}, TT=1, R={dom['R']}> "
                                                       f"gate_up N={dom['N']} K={dom['K']} T={eng.tree_size}",
                               "achieved": round(dom["gbs"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(dom["gbs"] / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": tsrc,
                               "avg_launch_us": round(dom["us"], 2), "bytes_per_launch": dom["bytes"],
                               "layer_gemms": {k: {"us": round(v["us"], 2), "GBs": round(v["gbs"], 1), "R": v["R"], "S": v["S"]}
                                               for k, v in kr.items()},
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
