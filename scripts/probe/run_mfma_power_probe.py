"""fp16 MFMA shapes under the power cap: TF/s of a pure matrix loop on random / small-magnitude / zero operands, 1 and 2 waves
per SIMD, with rocm-smi clock and power sampled while it runs.  python scripts/probe/run_mfma_power_probe.py"""
import ctypes as C, os, subprocess, threading, time, json
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/mfma_power_probe.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
                       "-o", so, os.path.join(here, "mfma_power_probe.hip")], stderr=subprocess.DEVNULL)
lib = C.CDLL(so)
lib.mfma_probe_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
dev = "cuda:0"
NL = int(os.environ.get("NL", "60"))
sink = torch.zeros(4, device=dev)
st = torch.cuda.current_stream().cuda_stream
def smi():
    try:
        d = json.loads(subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True).stdout)
        c = d[sorted(d)[0]]
        return c.get("sclk clock speed:", "?"), c.get("Current Socket Graphics Package Power (W)", "?")
    except Exception as e:
        return "?", "?"
for kind in ("randn", "randn*0.02 x randn", "zeros"):
    n16 = 2048 * 4 * 12 * 64 * 8
    if kind == "zeros":
        data = torch.zeros(n16, dtype=torch.float16, device=dev)
    elif kind == "randn":
        data = torch.randn(n16, device=dev).half()
    else:
        data = (torch.randn(n16, device=dev) * 0.02).half()
    for shape, name, macs, per_it in ((0, "16x16x32", 16 * 16 * 32, 32), (1, "32x32x16", 32 * 32 * 16, 8)):
        for bpc in (1, 2):
            blocks, iters = 256 * bpc, 40000 // bpc
            lib.mfma_probe_launch(shape, blocks, 200, data.data_ptr(), sink.data_ptr(), st)
            torch.cuda.synchronize()
            samples = []
            stop = threading.Event()
            th = threading.Thread(target=lambda: [samples.append(smi()) or time.sleep(0.05) for _ in iter(lambda: stop.is_set(), True)])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            th.start()
            e0.record()
            for _ in range(NL):
                lib.mfma_probe_launch(shape, blocks, iters, data.data_ptr(), sink.data_ptr(), st)
            e1.record()
            torch.cuda.synchronize()
            stop.set(); th.join()
            ms = e0.elapsed_time(e1) / NL
            flops = 2.0 * macs * per_it * iters * blocks * 4
            mid = samples[len(samples) // 2:] or [("?", "?")]
            print(f"{kind:20s} {name} {bpc} wave(s)/SIMD: {flops / ms / 1e9:7.0f} TF/s  ({ms:7.1f} ms per launch)  smi late samples: {mid[-3:]}", flush=True)
