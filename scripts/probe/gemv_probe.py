"""Times the GEMV probes (scripts/probe/gemv_probe.hip; v2 = gemv_probe2.hip with GV=2) against the low-latency MFMA kernels on the four linears of a
Llama-3.2-1B layer at T = 3: one hipGraph of 16 x (qkv, o, gate/up, down) launches each, weights rotated over 16 copies.
GPU box only."""
import ctypes as C, os, subprocess, sys, time
os.environ.setdefault("UMBRELLA_SYNTHETIC", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
GV = os.environ.get("GV", "1")
src = "scripts/probe/gemv_probe.hip" if GV == "1" else "scripts/probe/gemv_probe2.hip"
so = os.path.join(ROOT, "gpurun_out", f"libgemv_probe{GV}.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                       "-mllvm", "-amdgpu-kernarg-preload-count=16", os.path.join(ROOT, src), "-o", so])
probe = C.CDLL(so)
from umbrella_amd import _lib
from umbrella_amd.models.llama import PackedLinear
dev = "cuda:0"
T = int(os.environ.get("T", 3))
dtype = torch.float16
shapes = [("qkv", 3072, 2048, 12, 2), ("o", 2048, 2048, 8, 1), ("gu", 16384, 2048, 64, 2), ("down", 2048, 8192, 8, 1 if GV == "1" else 4)]
fn = probe.gv_probe if GV == "1" else probe.gv2_probe
L = 16
W = {n: [(torch.randn(N, K, device=dev) * 0.05).to(dtype) for _ in range(L)] for n, N, K, _, _ in shapes}
X = {n: torch.randn(T, K, device=dev).to(dtype) for n, N, K, _, _ in shapes}
O = {n: torch.empty(T, N, dtype=torch.float32, device=dev) for n, N, K, _, _ in shapes}
packed = {n: [PackedLinear.from_dense(w) for w in W[n]] for n in W}
st = torch.cuda.current_stream().cuda_stream

def gv(only=None):
    for l in range(L):
        for n, N, K, RB, RW in shapes:
            if only and n != only: continue
            rc = fn(C.c_void_p(O[n].data_ptr()), C.c_void_p(X[n].data_ptr()), C.c_void_p(W[n][l].data_ptr()), T, N, K, RB, RW, 0,
                                C.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, (n, rc)

LL = {}
def ll(only=None):
    for l in range(L):
        for n, N, K, RB, RW in shapes:
            if only and n != only: continue
            LL[n] = packed[n][l].apply_ll(X[n])

def timed(fn, *a):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(*a)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn(*a)
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    return (time.time() - t0) / 50 * 1e6 / L

# correctness of the probe
gv(); ll(); torch.cuda.synchronize()
for n, N, K, _, _ in shapes:
    ref = X[n].float() @ W[n][L - 1].float().t()
    err = float((O[n] - ref).abs().max() / ref.abs().max())
    assert err < 2e-3, (n, err)
print(f"T={T}: per-layer us   gemv {timed(gv):6.2f} | mfma-ll {timed(ll):6.2f}")
for n, N, K, _, _ in shapes:
    mb = N * K * 2 / 1e6
    a, b = timed(gv, n), timed(ll, n)
    print(f"  {n:5s} {mb:6.1f} MB  gemv {a:6.2f} us ({mb / a * 1e3 / 1e3:5.2f} TB/s) | mfma-ll {b:6.2f} us ({mb / b:5.2f} TB/s)")
