"""Runs scripts/probe/stream_probe.hip on the 70B gate/up footprint (3584 tiles x 64 k-blocks x 1 KiB = 235 MB per
launch, rotated over 8 copies).  Usage: python scripts/probe/run_stream_probe.py"""
import ctypes as C
import os
import subprocess
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/stream_probe.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so,
                       os.path.join(here, "stream_probe.hip")], stderr=subprocess.DEVNULL)
lib = C.CDLL(so)
lib.probe_launch.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
dev = "cuda:0"
tiles, npieces = int(os.environ.get("TILES", 3584)), int(os.environ.get("NP", 64))
per = tiles * npieces * 1024
ncopy = 8
buf = torch.empty(ncopy * per, dtype=torch.uint8, device=dev)
buf.random_(0, 255)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream


def run(u, lpp, bar, valu, lds):
    def go(i):
        rc = lib.probe_launch(buf.data_ptr() + (i % ncopy) * per, tiles, npieces, u, lpp, bar, valu, lds, sink.data_ptr(), st)
        assert rc == 0, (u, lpp, bar, valu)
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
    print(f"U={u:2d} KiB/piece={lpp} barrier={bar} valu={valu:2d} lds={lds//1024:3d}K: {us:6.1f} us  {per/us/1e3:6.0f} GB/s", flush=True)


for lds in (0, 40 << 10, 80 << 10):
    for u, lpp in ((4, 1), (8, 1), (16, 1), (4, 2), (8, 2), (16, 2)):
        run(u, lpp, 0, 0, lds)
if os.environ.get("ONLY_PLAIN"):      # per-CU rate experiments: TILES / NP choose how many CUs get a block
    sys.exit(0)
# valu: each unit is 3 full-rate VALU ops per dword; the exact int4 dequant is 13 per dword -> valu = 4; 7 = 21 per dword
for cfg in ((8, 2, 1, 0), (8, 1, 1, 0), (8, 2, 0, 4), (8, 2, 1, 4), (8, 2, 0, 7), (8, 2, 1, 7), (8, 1, 0, 4), (8, 1, 0, 7),
            (16, 1, 0, 4), (16, 1, 1, 7), (16, 1, 0, 7), (4, 2, 0, 7)):
    for lds in (0, 40 << 10, 80 << 10):
        run(*cfg, lds)
