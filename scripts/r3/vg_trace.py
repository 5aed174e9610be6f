"""Where a step of the two-phase verify GEMM spends its time: a second library built with -DUMB_VG_TRACE accumulates, per
wave, the shader-clock cycles of six segments of every step (activations -> LDS | dequant | load issue | barrier 1 |
fragment reads + MFMAs | barrier 2).  GPU box only."""
import os, subprocess, sys
os.environ.setdefault("UMBRELLA_SYNTHETIC", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "umbrella_amd", "csrc")
lib = os.path.join(ROOT, "gpurun_out", "libumbrella_vgtrace.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
srcs = ["gemm.hip", "lowlat.hip", "gemv.hip", "epilogue.hip", "attn.hip", "sample.hip", "tp.hip", "model.hip"]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DUMB_VG_TRACE", *os.environ.get("VG_DEFS", "").split(),
                       "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-mllvm", "-amdgpu-kernarg-preload-count=16",
                       *[os.path.join(CSRC, s) for s in srcs], "-o", lib], cwd=CSRC)
os.environ["UMB_LIB_PATH"] = lib
import torch
from umbrella_amd import _lib
from umbrella_amd.models.llama import PackedLinear
from umbrella_amd.models.synthetic import synth_awq_tensors
dev = "cuda:0"
gen = torch.Generator(device=dev).manual_seed(0)
T = int(os.environ.get("T", "256"))
dt = _lib.dtype_code(torch.float16)
names = ["x->LDS", "dequant", "load issue", "barrier 1", "reads+MFMA", "barrier 2"]
for name, N, K, il in (("gu", 57344, 8192, 1), ("down", 8192, 28672, 0)):
    lin = PackedLinear.from_awq(*synth_awq_tensors(N, K, 128, dev, gen), interleave=bool(il))
    S = _lib.load().umb_gemm_wide_split(T, N, lin.S)
    x = torch.randn(T, K, device=dev).to(torch.float16)
    out = torch.empty(max(S * T * N, 1), dtype=torch.float32, device=dev)
    nblocks = 4096
    trace = torch.zeros(nblocks * 8 * 8, dtype=torch.int64, device=dev)
    fx = _lib.UmbGemmFused()
    fx.counters = trace.data_ptr()
    epi = 2 if il else 0
    for _ in range(3):
        trace.zero_()
        _lib.call("umb_gemm_fused", out, x, K, lin.w, lin.meta, T, N, K, 1, S, lin.Rtb, epi, fx, dt)
    torch.cuda.synchronize()
    tr = trace.view(nblocks, 8, 8).cpu().double()
    used = tr[:, :, 6] > 0
    steps = tr[:, :, 6][used]
    print(f"== {name} N={N} K={K} T={T} S={S}: {int(used.sum())} waves traced, steps per wave {steps.mean():.0f}")
    tot = 0.0
    for i, n in enumerate(names):
        per = (tr[:, :, i][used] / steps)
        tot += per.mean()
        print(f"   {n:12s} {per.mean():8.0f} cycles/step  (p10 {per.quantile(0.1):6.0f}  p90 {per.quantile(0.9):6.0f})")
    print(f"   sum {tot:8.0f} cycles/step; MFMA-only time of a step = 64 x 16 = 1024 cycles")
