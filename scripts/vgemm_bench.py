"""Ablation of the register-resident verify GEMM (T = 256): each variant library (built with -DUMB_VG_*) gives WRONG
results and only tells what its piece costs.  Usage: python scripts/vgemm_bench.py <lib.so> [label]"""
import os, sys
os.environ.setdefault("UMBRELLA_SYNTHETIC", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] != "-":
    os.environ["UMB_LIB_PATH"] = os.path.abspath(sys.argv[1])
label = sys.argv[2] if len(sys.argv) > 2 else "default"
import torch
from umbrella_amd import _lib
from umbrella_amd.models.llama import PackedLinear
from umbrella_amd.models.synthetic import synth_awq_tensors
dev = "cuda:0"
gen = torch.Generator(device=dev).manual_seed(0)
lib = _lib.load()
T = int(os.environ.get("T", "256"))
dt = _lib.dtype_code(torch.float16)
tot = 0.0
res = []
for name, N, K, il in (("qkv", 10240, 8192, 0), ("o", 8192, 8192, 0), ("gu", 57344, 8192, 1), ("down", 8192, 28672, 0)):
    per = N * K // 2 + (N // 16) * (K // 128) * 64
    ncopy = max(2, int(500e6 // per) + 1)
    lins = [PackedLinear.from_awq(*synth_awq_tensors(N, K, 128, dev, gen), interleave=bool(il)) for _ in range(ncopy)]
    S = lib.umb_gemm_wide_split(T, N, lins[0].S)
    x = torch.randn(T, K, device=dev).to(torch.float16)
    if os.environ.get("XZERO"):          # DVFS probe: zero activations (the chip clocks to its power budget: MI355X_MICROARCH.md)
        x.zero_()
    out = torch.empty(max(S * T * N, 1), dtype=torch.float32, device=dev)
    epi = 2 if il else 0
    def launch(i):
        l = lins[i % ncopy]
        _lib.call("umb_gemm", out, x, K, l.w, l.meta, T, N, K, 1, S, l.Rtb, epi, dt)
    for i in range(3):
        launch(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    if os.environ.get("STAMP"):
        import time; print(f"{time.time():.3f} start {name}", flush=True)
    loops = int(os.environ.get("LOOPS", "20"))
    for i in range(loops):
        launch(i + 3)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / loops
    if os.environ.get("STAMP"):
        print(f"{time.time():.3f} end {name} {us:.1f} us", flush=True)
    tot += us
    res.append(f"{name} {us:6.1f} us ({2.0 * T * N * K / us / 1e6:6.0f} TF, S={S})")
    del lins
    torch.cuda.empty_cache()
print(f"{label:12s} T={T}: layer {tot:6.1f} us | " + " | ".join(res), flush=True)
