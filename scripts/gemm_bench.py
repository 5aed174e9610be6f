"""Micro-benchmark of umb_gemm at the verify shapes (rotating weight copies so reads come from HBM)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from umbrella_amd import _lib
from umbrella_amd.models.llama import PackedLinear
from umbrella_amd.models.synthetic import synth_awq_tensors

dev = "cuda:0"
T = int(os.environ.get("T", "13"))
dtype = torch.float16 if os.environ.get("DT", "fp16") == "fp16" else torch.bfloat16
gen = torch.Generator(device=dev).manual_seed(0)
SHAPES = {"70b": [("qkv", 10240, 8192, 1, 0), ("o", 8192, 8192, 1, 0), ("gu", 57344, 8192, 1, 1), ("down", 8192, 28672, 1, 0)],
          "70b-dense": [("qkv", 10240, 8192, 0, 0), ("o", 8192, 8192, 0, 0), ("gu", 57344, 8192, 0, 1), ("down", 8192, 28672, 0, 0)],
          "1b": [("qkv", 3072, 2048, 0, 0), ("o", 2048, 2048, 0, 0), ("gu", 16384, 2048, 0, 1), ("down", 2048, 8192, 0, 0),
                 ("head", 128256, 2048, 0, 0)]}
for model in sys.argv[1:] or ["70b", "1b"]:
    tot_b = tot_us = 0
    for name, N, K, awq, il in SHAPES[model]:
        per = N * K // 2 + (N // 16) * (K // 128) * 64 if awq else N * K * 2
        ncopy = max(2, int(600e6 // per) + 1)
        lins = []
        for _ in range(ncopy):
            if awq:
                qw, qz, sc = synth_awq_tensors(N, K, 128, dev, gen)
                lins.append(PackedLinear.from_awq(qw, qz, sc, interleave=bool(il)))
            else:
                w = torch.randn(N, K, device=dev, dtype=dtype) * 0.02
                lins.append(PackedLinear.from_dense(w, interleave=bool(il), force_s1=(name == "head")))
        ln = lins[0]
        x = torch.randn(T, K, device=dev).to(dtype)
        out = torch.empty(max(ln.S * T * N, 1), dtype=torch.float32, device=dev)
        epi = 2 if il else (1 if name == "head" else 0)

        def launch(i):
            l = lins[i % ncopy]
            _lib.call("umb_gemm", out, x, K, l.w, l.meta, T, N, K, l.awq, l.S, l.Rtb, epi, _lib.dtype_code(dtype))
        for i in range(3):
            launch(i)
        reps = 30
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for i in range(reps):
            launch(i + 3)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        tot_b += per; tot_us += us
        print(f"{model} {name:5s} N={N:6d} K={K:5d} awq={awq} R={ln.R} S={ln.S} T={T}: {us:7.2f} us  {per/us/1e3:7.1f} GB/s")
        del lins
        torch.cuda.empty_cache()
    print(f"{model} layer total: {tot_us:.1f} us, {tot_b/tot_us/1e3:.1f} GB/s")
