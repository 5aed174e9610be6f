"""Does a weight slab that was touched a few microseconds earlier (default cache policy) stream faster than a cold one?
Prices a 'prefetch the next layer into the Infinity Cache' helper before building one.  HIP events around ONE
umb_stream_read launch (non-temporal 16-byte loads, the GEMV / GEMM weight-stream policy), median of 15."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from umbrella_amd import _lib

dev = "cuda:0"
big = torch.empty(3 << 30, dtype=torch.uint8, device=dev); big.random_(0, 255)
flush = big[(2 << 30):].view(torch.int32)          # 1 GiB: evicts L2 and the 256 MiB Infinity Cache
sink = torch.zeros(1, dtype=torch.int32, device=dev)


def timed(n, prep):
    ts = []
    for _ in range(15):
        flush.sum()                                 # cold start for every sample
        prep()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.call("umb_stream_read", big, n, sink)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for mb in (8.4, 12.6, 33.5, 67, 122, 244):
    n = int(mb * 1e6) // 4096 * 4096
    x32 = big[:n].view(torch.int32)
    cold = timed(n, lambda: None)
    warm_def = timed(n, lambda: x32.sum())                                  # default-policy pass over X first
    warm_nt = timed(n, lambda: _lib.call("umb_stream_read", big, n, sink))  # non-temporal pass over X first
    half = timed(n, lambda: x32[: x32.numel() // 2].sum())                  # first half touched only
    print(f"{mb:6.1f} MB: cold {cold:6.2f} us ({n/cold/1e6:5.2f} TB/s) | after default-policy pass {warm_def:6.2f} ({n/warm_def/1e6:5.2f}) | "
          f"after nt pass {warm_nt:6.2f} ({n/warm_nt/1e6:5.2f}) | first half warmed {half:6.2f}", flush=True)
