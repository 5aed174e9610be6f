"""Dispatch-cadence probe: per-kernel cost of dependent no-op launches, eager vs hipGraph replay."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from umbrella_amd import _lib
p = torch.zeros(4, dtype=torch.int32, device="cuda:0")
N = 2000
for blocks in (1, 256, 2048):
    for _ in range(2):
        _lib.call("umb_bench_launch", N, blocks, p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time(); e0.record(); _lib.call("umb_bench_launch", N, blocks, p); e1.record(); th = time.time() - t0
    torch.cuda.synchronize()
    eager = e0.elapsed_time(e1) * 1e3 / N
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        _lib.call("umb_bench_launch", 8, blocks, p)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        _lib.call("umb_bench_launch", N, blocks, p)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"blocks={blocks}: eager {eager:.2f} us/kernel (host issue {th*1e6/N:.2f} us), graph {e0.elapsed_time(e1)*1e3/N:.2f} us/kernel")
