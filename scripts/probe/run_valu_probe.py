import ctypes as C
import os
import subprocess

import torch

here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/valu_probe.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so,
                       os.path.join(here, "valu_probe.hip")], stderr=subprocess.DEVNULL)
lib = C.CDLL(so)
lib.vprobe_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
out = torch.zeros(4, dtype=torch.int32, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
NAMES = ["v_and_or_b32", "v_pk_add_f16", "v_pk_mul_f16", "v_pk_fma_f16", "v_add_u32", "v_fma_f32", "v_lshrrev_b32", "v_mul_f16", "v_fma_f16",
         "v_bfi_b32", "v_and_b32", "v_or_b32", "v_perm_b32", "v_fma_mix_f32", "v_cvt_f32_f16", "v_lshl_or_b32", None, "v_and_or(sgpr)",
         None, "v_xor_b32", "v_pk_add_u16", "v_bfe_u32", "v_cvt_f32_ubyte0", "v_mad_u32_u24", "v_and(lit)", "v_or(lit)"]
iters = 2000
for op, name in enumerate(NAMES):
    if name is None:
        continue
    row = []
    for bpc in (1, 2):                        # blocks per CU -> waves per SIMD
        blocks = 256 * bpc
        for _ in range(2):
            lib.vprobe_launch(op, blocks, iters, out.data_ptr(), st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            lib.vprobe_launch(op, blocks, iters, out.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 5
        n = iters * 64 * bpc                      # wave instructions per SIMD
        row.append(f"{bpc} w/SIMD: {us*1e3/n:5.2f} ns/instr")
    print(f"{name:14s} " + " | ".join(row), flush=True)
