"""Diagnostic: low-latency int4 GEMM at full size vs the dequantised fp32 reference, per token tiling / ring depth."""
import os, sys
os.environ.setdefault("UMBRELLA_SYNTHETIC", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from umbrella_amd.models.llama import PackedLinear, ll_plan
from umbrella_amd.models.synthetic import synth_awq_tensors
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_hip_engine import _awq_dequant_torch

dev = "cuda:0"
N, K = int(os.environ.get("N", 8192)), int(os.environ.get("K", 8192))
gen = torch.Generator(device=dev).manual_seed(N + K)
qw, qz, sc = synth_awq_tensors(N, K, 128, dev, gen)
lin = PackedLinear.from_awq(qw, qz, sc)
wd = _awq_dequant_torch(qw, qz, sc).float()          # [K, N]
x = (torch.randn(64, K, device=dev, generator=gen) * 0.5).half()
ref = x.float() @ wd
print("plan", ll_plan(N, K, True), "PF cap", os.environ.get("UMB_LL_PF"))
for T in (1, 13, 16, 31, 40, 64):
    y = lin.apply_ll(x[:T].contiguous())
    err = (y - ref[:T]).abs()
    bad = (err > 0.02 * ref.abs().max()).nonzero()
    print(f"T={T}: max err {float(err.max()):.4f} (ref max {float(ref.abs().max()):.2f}), bad elems {bad.shape[0]}",
          "first bad (t, n):", bad[:4].tolist(), "cols range:", (int(bad[:, 1].min()), int(bad[:, 1].max())) if bad.numel() else None)
    y2 = lin.apply_ll(x[:T].contiguous())
    print("    repeatable:", bool(torch.equal(y, y2)))
ys = lin.apply(x[:13].contiguous())
print("split-K family err:", float((ys - ref[:13]).abs().max()))
