"""Sweep the split-K count S of the skinny GEMM for the layer shapes of the BASELINE models (T = 13)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from umbrella_amd import _lib
from umbrella_amd.models.llama import PackedLinear
from umbrella_amd.models.synthetic import synth_awq_tensors
dev = "cuda:0"; T = int(os.environ.get("T", "13"))
gen = torch.Generator(device=dev).manual_seed(0)
SHAPES = [("70b qkv", 10240, 8192, 1), ("70b o", 8192, 8192, 1), ("70b down", 8192, 28672, 1),
          ("8b-awq gu", 28672, 4096, 1), ("8b-awq qkv", 6144, 4096, 1), ("8b-awq o", 4096, 4096, 1), ("8b-awq down", 4096, 14336, 1),
          ("8b qkv", 6144, 4096, 0), ("8b o", 4096, 4096, 0), ("8b down", 4096, 14336, 0),
          ("1b qkv", 3072, 2048, 0), ("1b o", 2048, 2048, 0), ("1b down", 2048, 8192, 0)]
only = sys.argv[1:] 
for name, N, K, awq in SHAPES:
    if only and not any(o in name for o in only):
        continue
    dtype = torch.float16 if awq else torch.bfloat16
    per = N * K // 2 + (N // 16) * (K // 128) * 64 if awq else N * K * 2
    lins = []
    for _ in range(max(3, int(600e6 // per) + 1)):
        if awq:
            qw, qz, sc = synth_awq_tensors(N, K, 128, dev, gen)
            lins.append(PackedLinear.from_awq(qw, qz, sc))
        else:
            lins.append(PackedLinear.from_dense(torch.randn(N, K, device=dev, dtype=dtype) * 0.02))
    x = torch.randn(T, K, device=dev).to(dtype)
    out = torch.empty(16 * T * N, dtype=torch.float32, device=dev)
    l0 = lins[0]
    res = []
    KB = K // 128
    for S in range(1, 17):
        if KB // S < (4 if awq else 2):
            break
        def launch(i):
            l = lins[i % len(lins)]
            _lib.call("umb_gemm", out, x, K, l.w, l.meta, T, N, K, l.awq, S, l.Rtb, 0, _lib.dtype_code(dtype))
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            launch(0); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for i in range(24):
                    launch(i)
                    _lib.call("umb_bench_launch", 2, 1, None)
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(3):
                e0.record(s); g.replay(); e1.record(s); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / 24 - 3.1)
        res.append((S, best))
    nblk = (N // 16) // (4 * l0.R)
    bs = min(res, key=lambda r: r[1])
    print(f"{name:12s} N={N} K={K} R={l0.R} nblk={nblk} plan S={l0.S}: " + " ".join(f"{S}:{u:.1f}" for S, u in res) + f"  best S={bs[0]} ({bs[1]:.1f} us, {per / bs[1] / 1e3:.0f} GB/s)")
    del lins; torch.cuda.empty_cache()
