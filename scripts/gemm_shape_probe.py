"""Probe: int4 skinny GEMM bandwidth vs output width N (block-count quantisation over the 256 CUs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from umbrella_amd import _lib
from umbrella_amd.models.llama import PackedLinear
from umbrella_amd.models.synthetic import synth_awq_tensors
dev = "cuda:0"; T = 13; dtype = torch.float16; K = 8192
gen = torch.Generator(device=dev).manual_seed(0)
for N in [int(a) for a in sys.argv[1:]] or [49152, 57344, 65536]:
    per = N * K // 2 + (N // 16) * (K // 128) * 64
    lins = []
    for _ in range(max(2, int(700e6 // per) + 1)):
        qw, qz, sc = synth_awq_tensors(N, K, 128, dev, gen)
        lins.append(PackedLinear.from_awq(qw, qz, sc, interleave=True))
    x = torch.randn(T, K, device=dev).to(dtype)
    out = torch.empty(T * N, dtype=torch.float32, device=dev)
    l0 = lins[0]
    def launch(i):
        l = lins[i % len(lins)]
        _lib.call("umb_gemm", out, x, K, l.w, l.meta, T, N, K, 1, l.S, l.Rtb, 2, _lib.dtype_code(dtype))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        launch(0); launch(1); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(30):
                launch(i)
                _lib.call("umb_bench_launch", 2, 1, None)          # a gap like the small kernels between GEMMs
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s); g.replay(); e1.record(s); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30 - 3.1
    nblk = (N // 16) // (4 * l0.R)
    print(f"N={N:6d} R={l0.R} S={l0.S} blocks={nblk * l0.S:5d} ({nblk * l0.S / 256:.2f}/CU): {us:7.2f} us  {per / us / 1e3:7.1f} GB/s")
    del lins; torch.cuda.empty_cache()
