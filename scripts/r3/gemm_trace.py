"""Where the fixed ~5 us of a skinny int4 GEMM launch goes: per-block phase stamps (constant 100 MHz clock) of the
70B layer GEMMs at T = 13.  Builds a second library with -DUMB_GEMM_TRACE, runs each shape a few times with rotating
weights, prints when (relative to the first block's start) blocks start, have x staged, finish their first k-block,
leave the main loop and end.  GPU box only."""
import os, subprocess, sys
os.environ.setdefault("UMBRELLA_SYNTHETIC", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "umbrella_amd", "csrc")
lib = os.path.join(ROOT, "gpurun_out", "libumbrella_trace.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
srcs = ["gemm.hip", "lowlat.hip", "gemv.hip", "epilogue.hip", "attn.hip", "sample.hip", "tp.hip", "model.hip"]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DUMB_GEMM_TRACE",
                       "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-mllvm", "-amdgpu-kernarg-preload-count=16", *[os.path.join(CSRC, s) for s in srcs], "-o", lib], cwd=CSRC)
os.environ["UMB_LIB_PATH"] = lib
import numpy as np
import torch
from umbrella_amd import _lib
from umbrella_amd.models.llama import PackedLinear
from umbrella_amd.models.synthetic import synth_awq_tensors

dev = "cuda:0"
gen = torch.Generator(device=dev).manual_seed(0)
T, dt = 13, _lib.dtype_code(torch.float16)
for name, N, K, il, S_eff in (("gu", 57344, 8192, 1, 1), ("down", 8192, 28672, 0, 8), ("qkv", 10240, 8192, 0, 7), ("o", 8192, 8192, 0, 4)):
    per = N * K // 2 + (N // 16) * (K // 128) * 64
    ncopy = max(3, int(700e6 // per) + 1)
    lins = [PackedLinear.from_awq(*synth_awq_tensors(N, K, 128, dev, gen), interleave=bool(il)) for _ in range(ncopy)]
    ln = lins[0]
    x = torch.randn(T, K, device=dev).to(torch.float16)
    out = torch.empty(max(S_eff * T * N, T * N), dtype=torch.float32, device=dev)
    tbv = (ln.tb & 0x7f) or 4 * ln.R
    nblk = (N // 16 + tbv - 1) // tbv
    blocks = nblk * S_eff
    trace = torch.zeros(blocks * 16, dtype=torch.int64, device=dev)
    fx = _lib.UmbGemmFused()
    fx.counters = trace.data_ptr()
    epi = 2 if il else 0
    res = []
    for i in range(6):
        l = lins[i % ncopy]
        trace.zero_()
        torch.cuda.synchronize()
        _lib.call("umb_gemm_fused", out, x, K, l.w, l.meta, T, N, K, 1, S_eff, l.Rtb, epi, fx, dt)
        torch.cuda.synchronize()
        t = trace.view(blocks, 16).cpu().numpy().astype(np.int64)
        if i >= 2:
            res.append(t)
    print(f"== {name} N={N} K={K} S={S_eff} R={ln.R} blocks={blocks} ({per/1e6:.1f} MB)")
    for t in res[-2:]:
        t0 = t[:, 0].min()
        rel = (t[:, :6] - t0) / 100.0                     # us (100 MHz clock)
        names = ["start", "prologue issued", "x0 staged", "first k-block done", "loop done", "end"]
        for j, nm in enumerate(names):
            c = rel[:, j]
            print(f"   {nm:20s} min {c.min():6.2f}  p50 {np.median(c):6.2f}  p90 {np.percentile(c, 90):6.2f}  max {c.max():6.2f} us")
        ch = (t[:, 6:16] - t0) / 100.0
        ok = (t[:, 6:16] > 0).all(axis=0)
        med = [f"{np.median(ch[:, i]):.2f}" if ok[i] else "-" for i in range(10)]
        print("   chunk entry p50 (us): " + " ".join(med))
        dur = rel[:, 5] - rel[:, 0]
        print(f"   block lifetime       min {dur.min():6.2f}  p50 {np.median(dur):6.2f}  max {dur.max():6.2f} us; kernel span {rel[:,5].max():.2f} us; "
              f"late starters (start > 2 us): {(rel[:,0] > 2).sum()}")
    del lins
    torch.cuda.empty_cache()
