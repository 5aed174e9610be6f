"""Tree-attention microbench: time umb_tree_attn alone for one layer shape.
  python scripts/attn_bench.py --T 257 --prefix 128 --Hq 64 --Hkv 8 --D 128 --Lmax 4096
Reports us per launch (HIP events over a hipGraph of `--reps` launches), the K/V bytes of the visible keys
(algorithmic HBM traffic) and the QK^T + PV flops.
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from umbrella_amd import _lib
from umbrella_amd.models.llama import pack_mask_bits

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=257)
ap.add_argument("--prefix", type=int, default=128)
ap.add_argument("--Hq", type=int, default=64)
ap.add_argument("--Hkv", type=int, default=8)
ap.add_argument("--D", type=int, default=128)
ap.add_argument("--Lmax", type=int, default=4096)
ap.add_argument("--chunk", type=int, default=0)
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--causal", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
dt = torch.float16
T, Hq, Hkv, D, Lmax = a.T, a.Hq, a.Hkv, a.D, a.Lmax
chunk = a.chunk or max(128, (Lmax // 16 + 31) // 32 * 32)
splits = (Lmax + chunk - 1) // chunk
g = torch.Generator().manual_seed(0)
q = torch.randn(T, Hq, D, generator=g).to(dt).to(dev)
kc = torch.randn(Hkv, Lmax, D, generator=g).to(dt).to(dev)
vt = torch.randn(Hkv, D, Lmax + 32, generator=g).to(dt).to(dev)
# random tree mask: node t sees itself and a random ancestor chain
par = [0] + [int(torch.randint(0, t, (1,), generator=g)) for t in range(1, T)]
m = torch.zeros(T, T, dtype=torch.bool)
for t in range(T):
    m[t] = m[par[t]] if t else m[t]
    m[t, t] = True
bits = None if a.causal else pack_mask_bits(m).to(dev)
out = torch.empty(T, Hq, D, dtype=dt, device=dev)
po = torch.empty(splits * T * Hq * D, dtype=torch.float32, device=dev)
pml = torch.empty(splits * T * Hq * 2, dtype=torch.float32, device=dev)
pre = torch.tensor([a.prefix], dtype=torch.int32, device=dev)
# counters -> the single-launch kernel the models use (UMB_ATTN_SPLIT=1 forces the split + combine pair)
cnt = torch.zeros(Hkv * ((T * (Hq // Hkv) + 15) // 16) + 64, dtype=torch.int32, device=dev)


def launch():
    _lib.call("umb_tree_attn", out, q, kc, vt, po, pml, pre, bits, 0 if bits is None else bits.shape[1], T, T, Hq, Hkv, D, Lmax,
              chunk, splits, D ** -0.5, cnt, _lib.dtype_code(dt))


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    launch(); launch()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        for _ in range(a.reps):
            launch()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s); gr.replay(); e1.record(s); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / a.reps
keys = a.prefix + T
kv_bytes = 2 * Hkv * keys * D * 2
flops = 4 * T * Hq * keys * D
print(json.dumps({"T": T, "prefix": a.prefix, "Hq": Hq, "Hkv": Hkv, "D": D, "Lmax": Lmax, "chunk": chunk,
                  "us_per_launch(attn+combine)": round(us, 2), "kv_MB": round(kv_bytes / 1e6, 2),
                  "TFLOPs": round(flops / us / 1e6, 1)}))
