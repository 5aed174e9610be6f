"""Runs scripts/probe/gemm_probe.hip on the 70B gate/up footprint (3584 tiles x 64 k-blocks)."""
import ctypes as C
import os
import subprocess

import torch

here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/gemm_probe.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", so,
                       os.path.join(here, "gemm_probe.hip")], stderr=subprocess.DEVNULL)
lib = C.CDLL(so)
lib.gprobe_launch.argtypes = [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p, C.c_void_p]
dev = "cuda:0"
tiles, kb = 3584, int(os.environ.get("KB", 64))
per = tiles * kb * 1024
ncopy = 6
w = torch.empty(ncopy * per, dtype=torch.uint8, device=dev).random_(0, 255)
meta = torch.empty(ncopy * per // 16, dtype=torch.uint8, device=dev).random_(0, 255)
x = torch.randn(64 * 8192, device=dev).half()
out = torch.zeros(4, device=dev)
st = torch.cuda.current_stream().cuda_stream
NAMES = {1: "mfma", 2: "dequant", 4: "B from LDS", 8: "x staging+barrier", 16: "meta per-kb loads", 32: "meta staged", 64: "FOLDED dequant", 128: "x sums per chunk (8 MFMA on one wave per k-block + barrier)"}


def run(feat, r, cb):
    def go(i):
        k = i % ncopy
        rc = lib.gprobe_launch(w.data_ptr() + k * per, meta.data_ptr() + k * (per // 16), x.data_ptr(), tiles, kb, feat, r, cb,
                               out.data_ptr(), st)
        assert rc == 0, (feat, r, cb)
    for i in range(4):
        go(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(40):
        go(i + 4)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 40
    what = " + ".join(v for k, v in NAMES.items() if feat & k) or "loads only"
    print(f"KB={kb:3d} feat={feat:2d} R={r} CB={cb}: {us:6.1f} us  {per/us/1e3:6.0f} GB/s   [{what}]", flush=True)


CASES = ((0, 2, 4), (13, 2, 4), (45, 2, 4), (47, 2, 4), (77, 2, 4), (109, 2, 4), (205, 2, 4), (237, 2, 4), (237, 1, 4))
if os.environ.get("CASES"):
    CASES = tuple(tuple(int(v) for v in c.split(":")) for c in os.environ["CASES"].split(","))
for feat, r, cb in CASES:
    run(feat, r, cb)
