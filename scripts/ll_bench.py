"""A/B micro-benchmark: split-K family (umb_gemm + its reduce kernel) vs low-latency family (umb_gemm_ll) per layer
shape, weights rotated over enough copies that every launch streams from HBM; then whole forwards of a model under
hipGraph replay with both schedules (UMB_SCHED).  Usage: python scripts/ll_bench.py [1b] [70b] [8b] [fwd1b] [fwd70b]"""
import os

os.environ.setdefault("UMBRELLA_SYNTHETIC", "1")   # benchmarks run on seeded random weights of the exact shapes (no checkpoints offline)
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from umbrella_amd import _lib
from umbrella_amd.models.llama import PackedLinear, ll_plan, to_fm
from umbrella_amd.models.synthetic import synth_awq_tensors

dev = "cuda:0"
gen = torch.Generator(device=dev).manual_seed(0)
SHAPES = {"70b": (torch.float16, 13, [("qkv", 10240, 8192, 1, 0), ("o", 8192, 8192, 1, 0), ("gu", 57344, 8192, 1, 1), ("down", 8192, 28672, 1, 0)]),
          "1b": (torch.float16, 3, [("qkv", 3072, 2048, 0, 0), ("o", 2048, 2048, 0, 0), ("gu", 16384, 2048, 0, 1), ("down", 2048, 8192, 0, 0),
                                    ("head", 128256, 2048, 0, 0)]),
          # balance experiment: 4096 n-tiles = 512 blocks = exactly 2 per CU (the real gate/up has 3584 = 448 blocks)
          "bal": (torch.float16, 13, [("gu", 57344, 8192, 1, 1), ("gu", 65536, 8192, 1, 1), ("gu", 49152, 8192, 1, 1)]),
          # fixed cost vs streaming slope: the same N at three K
          "kscan": (torch.float16, 13, [("gu", 57344, 512, 1, 1), ("gu", 57344, 1024, 1, 1), ("gu", 57344, 2048, 1, 1), ("gu", 57344, 4096, 1, 1), ("gu", 57344, 8192, 1, 1), ("gu", 57344, 16384, 1, 1)]),
          "8bawq": (torch.float16, 32, [("qkv", 6144, 4096, 1, 0), ("o", 4096, 4096, 1, 0), ("gu", 28672, 4096, 1, 1), ("down", 4096, 14336, 1, 0)]),
          "8b": (torch.bfloat16, 31, [("qkv", 6144, 4096, 0, 0), ("o", 4096, 4096, 0, 0), ("gu", 28672, 4096, 0, 1), ("down", 4096, 14336, 0, 0)])}


def timeit(fn, reps=40, warm=4):
    for i in range(warm):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(reps):
        fn(i + warm)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def bench_shapes(model):
    dtype, T, shapes = SHAPES[model]
    T = int(os.environ.get("T", T))
    dt = _lib.dtype_code(dtype)
    tot = [0.0, 0.0, 0.0]
    for name, N, K, awq, il in shapes:
        if os.environ.get("ONLY") and name not in os.environ["ONLY"].split(","):
            continue
        per = N * K // 2 + (N // 16) * (K // 128) * 64 if awq else N * K * 2
        ncopy = max(2, int(700e6 // per) + 1)
        lins = []
        for _ in range(ncopy):
            if awq:
                lins.append(PackedLinear.from_awq(*synth_awq_tensors(N, K, 128, dev, gen), interleave=bool(il)))
            else:
                lins.append(PackedLinear.from_dense(torch.randn(N, K, device=dev, dtype=dtype) * 0.02, interleave=bool(il),
                                                    force_s1=(name == "head")))
        ln = lins[0]
        x = torch.randn(T, K, device=dev).to(dtype)
        xfm = to_fm(x)
        part = torch.empty(max(ln.S * T * N, 1), dtype=torch.float32, device=dev)
        h = torch.zeros(T, N, dtype=dtype, device=dev)
        xn = torch.zeros(T, N, dtype=dtype, device=dev)
        nw = torch.ones(N, dtype=dtype, device=dev)
        epi_old = 2 if il else (1 if name == "head" else 0)
        S_eff = ln.S
        if name in ("o", "down") and T <= 64:           # what model.hip's eff_s() does for the row-reduced GEMMs
            cap = max(K // 1792, 4)
            if ln.S_row > 0:
                S_eff = ln.S_row
            elif ln.S > cap and (N // (64 * ln.R)) * cap >= 256:
                S_eff = cap

        R_old = int(os.environ.get("R_OLD", ln.R)) | (int(os.environ.get("TB_OLD", ln.tb)) << 8)
        if os.environ.get("S_OLD") and epi_old == 0:
            S_eff = int(os.environ["S_OLD"])
            part = torch.empty(max(S_eff * T * N, 1), dtype=torch.float32, device=dev)

        def old(i):
            l = lins[i % ncopy]
            _lib.call("umb_gemm", part, x, K, l.w, l.meta, T, N, K, l.awq, S_eff if epi_old == 0 else l.S, R_old, epi_old, dt)
            if name in ("o", "down"):
                _lib.call("umb_reduce_residual_norm", part, S_eff, T, N, h, h, xn, nw, 1e-5, dt)

        # low-latency: the layer's own epilogue (logits for head / qkv, SiLU for gate-up, residual for o / down)
        fx = _lib.UmbGemmLL()
        out_ll = torch.empty(T * N, dtype=torch.float32, device=dev)
        R, WN, WK, NW = ll_plan(N, K, bool(awq))
        ssq = torch.zeros(T, max(N // 16, 4), dtype=torch.float32, device=dev)
        hw = torch.zeros(64 * N, dtype=dtype, device=dev)
        epi_new = 2 if il else (4 if name in ("o", "down") else 0)
        if epi_new == 4:
            fx.h, fx.hw, fx.norm_w, fx.ssq_out, fx.ssq_out_stride = h.data_ptr(), hw.data_ptr(), nw.data_ptr(), ssq.data_ptr(), ssq.shape[1]

        def new(i):
            l = lins[i % ncopy]
            _lib.call("umb_gemm_ll", out_ll, xfm, l.w, l.meta, T, N, K, l.awq, epi_new, fx, dt)
        us_old, us_new = timeit(old), timeit(new)
        tot[0] += per; tot[1] += us_old; tot[2] += us_new
        print(f"{model} {name:5s} N={N:6d} K={K:5d} awq={awq} T={T}: split S={S_eff} {us_old:7.2f} us {per/us_old/1e3:6.0f} GB/s | "
              f"ll R={R} WN={WN} WK={WK} {us_new:7.2f} us {per/us_new/1e3:6.0f} GB/s", flush=True)
        del lins
        torch.cuda.empty_cache()
    print(f"{model} layer GEMMs: split {tot[1]:.1f} us ({tot[0]/tot[1]/1e3:.0f} GB/s) | ll {tot[2]:.1f} us ({tot[0]/tot[2]/1e3:.0f} GB/s)", flush=True)


def bench_forward(name, layers, T, dtype):
    """graph-replayed T-row tree forward of a (possibly truncated) model, both schedules"""
    import copy
    from umbrella_amd.models.config import KNOWN
    from umbrella_amd.models.llama import Llama
    res = {}
    for sched in os.environ.get("SCHEDS", "split,ll").split(","):
        os.environ["UMB_SCHED"] = sched
        cfg = copy.copy(KNOWN[name])
        cfg.num_hidden_layers = layers
        m = Llama(name, max_length=2048, device=dev, dtype=dtype, config=cfg)
        m.alloc()
        m.use_gemv(os.environ.get("GEMV", "1") != "0")         # draft role (<= 4-row forwards on the GEMV family where it exists)
        ids = torch.randint(3, 128000, (128 + T,), dtype=torch.int32, device=dev)
        m.prefill_tokens(ids[:128], 0)
        step = ids[128:].contiguous()
        pos = torch.arange(128, 128 + T, dtype=torch.int32, device=dev)
        pre = torch.tensor([128], dtype=torch.int32, device=dev)
        run = lambda: m.forward_explicit(step, pos, pos, pre, head_from=0)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.time()
        n = 50
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        res[sched] = (time.time() - t0) / n * 1e3
        wb = m.weight_bytes()
        del m, g
        torch.cuda.empty_cache()
    best = min(res.values())
    print(f"forward {name} L={layers} T={T}: " + " | ".join(f"{k} {v:.3f} ms" for k, v in res.items()) +
          f" | weights {wb/1e9:.2f} GB -> {wb/best/1e6:.0f} GB/s (best)  [UMB_PF_MB={os.environ.get('UMB_PF_MB', 'default')}]", flush=True)


def bench_stream():
    """read-only streaming rate of this box (umb_stream_read): one 2 GiB pass, and gate/up-sized (235 MB) launches
    rotated over 4 regions so each launch misses the 256 MB Infinity Cache like a layer's weights do"""
    buf = torch.empty(2 << 30, dtype=torch.uint8, device=dev)
    buf.random_(0, 255)
    sink = torch.zeros(1, dtype=torch.int32, device=dev)
    us = timeit(lambda i: _lib.call("umb_stream_read", buf, buf.numel(), sink), reps=10, warm=2)
    print(f"stream read 2 GiB: {us:.1f} us -> {buf.numel()/us/1e3:.0f} GB/s", flush=True)
    for mb in (int(x) for x in os.environ.get("STREAM_MB", "33,67,235,470").split(",")):
        n = mb * 1000 * 1000 // 16 * 16
        k = min(buf.numel() // n, 8)
        us = timeit(lambda i: _lib.call("umb_stream_read", buf[(i % k) * n:], n, sink), reps=40, warm=4)
        print(f"stream read {mb} MB launches (rotating {k} regions): {us:.1f} us -> {n/us/1e3:.0f} GB/s", flush=True)
        us = timeit(lambda i: _lib.call("umb_stream_read", buf, n, sink), reps=40, warm=4)
        print(f"stream read {mb} MB launches (SAME region, Infinity Cache resident if it fits): {us:.1f} us -> {n/us/1e3:.0f} GB/s", flush=True)


def bench_graphscan():
    """In-graph cost of one dependent launch vs K: 24 launches of the 70B gate/up GEMM (each reading the previous one's
    output buffer as nothing -- the dependency is the stream order) captured in one hipGraph, weights rotated; the same
    for umb_stream_read over the same bytes.  Fits time = fixed + bytes / rate for both."""
    dtype, T, N = torch.float16, 13, 57344
    dt = _lib.dtype_code(dtype)
    sink = torch.zeros(1, dtype=torch.int32, device=dev)
    rows = []
    for K in (1024, 2048, 4096, 8192):
        per = N * K // 2 + (N // 16) * (K // 128) * 64
        ncopy = max(3, int(600e6 // per) + 1)
        lins = [PackedLinear.from_awq(*synth_awq_tensors(N, K, 128, dev, gen), interleave=True) for _ in range(ncopy)]
        x = torch.randn(T, K, device=dev).to(dtype)
        act = torch.zeros(T, N // 2, dtype=dtype, device=dev)
        nl = 24

        def gemms():
            for i in range(nl):
                l = lins[i % ncopy]
                _lib.call("umb_gemm", act, x, K, l.w, l.meta, T, N, K, l.awq, 1, l.Rtb, 2, dt)

        def reads():
            for i in range(nl):
                l = lins[i % ncopy]
                _lib.call("umb_stream_read", l.w, l.w.numel() * l.w.element_size(), sink)

        res = []
        for fn in (gemms, reads):
            s_ = torch.cuda.Stream()
            with torch.cuda.stream(s_):
                fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(20):
                g.replay()
            torch.cuda.synchronize()
            res.append((time.time() - t0) / 20 / nl * 1e6)
        rows.append((K, per, res[0], res[1]))
        print(f"graph K={K:5d} {per/1e6:6.1f} MB: gemm {res[0]:6.2f} us/launch | stream read {res[1]:6.2f} us/launch", flush=True)
        del lins
        torch.cuda.empty_cache()
    (k0, b0, g0, r0), (k1, b1, g1, r1) = rows[1], rows[3]
    for name, a, b in (("gemm", g0, g1), ("read", r0, r1)):
        slope = (b - a) / (b1 - b0)
        print(f"{name}: fixed {a - slope * b0:5.2f} us + bytes / {1 / slope / 1e6:5.2f} TB/s", flush=True)


for what in sys.argv[1:] or ["1b", "70b", "fwd1b", "fwd70b"]:
    if what == "graphscan":
        bench_graphscan()
    elif what == "stream":
        bench_stream()
    elif what in SHAPES:
        bench_shapes(what)
    elif what == "fwd1b":
        for T in (int(v) for v in os.environ.get("T1B", "1,3").split(",")):
            bench_forward("meta-llama/Llama-3.2-1B-Instruct", 16, T, torch.float16)
    elif what == "fwd70b":
        bench_forward("hugging-quants/Meta-Llama-3.1-70B-Instruct-AWQ-INT4", 16, int(os.environ.get("T70", 13)), torch.float16)
    elif what == "fwd8bawq":
        for T in (int(v) for v in os.environ.get("T8B", "1,2,32").split(",")):
            bench_forward("hugging-quants/Meta-Llama-3.1-8B-Instruct-AWQ-INT4", int(os.environ.get("L8B", 32)), T, torch.float16)
    elif what == "fwd8b":
        for T in (int(v) for v in os.environ.get("T8B", "31").split(",")):
            bench_forward("meta-llama/Llama-3.1-8B-Instruct", 32, T, torch.bfloat16)
