"""umb_reduce_qkv_rope alone (70B heads), graph of 20 launches: heads per block (UMB_RQR_HPB) at 13 / 257 / 769 / 1024 rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from umbrella_amd import _lib
dev = "cuda:0"; dt = torch.float16
Hq, Hkv, D, Lmax = 64, 8, 128, 4096
N = (Hq + 2 * Hkv) * D
for T, S in ((13, 7), (257, 6), (769, 2), (1024, 3)):
    part = torch.randn(S, T, N, device=dev)
    pos = torch.arange(128, 128 + T, dtype=torch.int32, device=dev); slot = pos.clone()
    cos = torch.randn(Lmax, D, device=dev).to(dt); sin = torch.randn(Lmax, D, device=dev).to(dt)
    q = torch.empty(T, Hq, D, dtype=dt, device=dev)
    kc = torch.zeros(Hkv, Lmax, D, dtype=dt, device=dev); vt = torch.zeros(Hkv, D, Lmax + 32, dtype=dt, device=dev)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        go = lambda: _lib.call("umb_reduce_qkv_rope", part, S, T, Hq, Hkv, D, Lmax, pos, slot, cos, sin, q, kc, vt, 1, None, 0)
        go(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(20): go()
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10): g.replay()
        e1.record(st); torch.cuda.synchronize()
    print(f"T={T} S={S}: {e0.elapsed_time(e1)*1000/200:.2f} us", flush=True)
