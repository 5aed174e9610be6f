"""How the row-reduce kernels' time steps with the row count (one 1024-thread block per row, one block per CU):
T = 256 vs 257 (S = 8) and 768 vs 769 (S = 2) -- is the single extra block a whole second / fourth round?
Graph of 20 dependent launches, wall / 20."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from umbrella_amd import _lib as L

dev = "cuda:0"
N = 8192


def bench(T, S, reps=20):
    part = torch.randn(S, T, N, device=dev, dtype=torch.float32)
    res = torch.randn(T, N, device=dev, dtype=torch.float16)
    h = torch.empty_like(res); xn = torch.empty_like(res)
    w = torch.ones(N, device=dev, dtype=torch.float16)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        def go():
            L.check(L.load().umb_reduce_residual_norm(L.ptr(part), S, T, N, L.ptr(res), L.ptr(h), L.ptr(xn), L.ptr(w), 1e-5,
                                                      L.dtype_code(torch.float16), L.stream_ptr()))
        go(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                go()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10):
            g.replay()
        e1.record(st)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / (10 * reps)


for T, S in ((13, 4), (13, 8), (128, 8), (255, 8), (256, 8), (257, 8), (258, 8), (320, 8), (384, 8), (512, 8), (513, 8),
             (768, 2), (769, 2), (770, 2), (1024, 2), (256, 2), (257, 2)):
    print(f"T={T:5d} S={S}  {bench(T, S):7.2f} us", flush=True)
