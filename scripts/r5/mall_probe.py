"""Does a verify-layer GEMM run faster when its weights were touched just before (Infinity Cache / L2 resident)?
cold: rotating weight copies (HBM); hot: the same weights every launch; pre: rotating, but a reader kernel touches the
first PRE_MB of the weights right before the launch (what a prefetch rider on the small kernels would do)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.build()
from umbrella_amd import _lib
from umbrella_amd.models.llama import PackedLinear
from umbrella_amd.models.synthetic import synth_awq_tensors

dev = "cuda:0"
T = int(os.environ.get("T", "13"))
dtype = torch.float16
gen = torch.Generator(device=dev).manual_seed(0)
SH = [("qkv", 10240, 8192, 0), ("o", 8192, 8192, 0), ("gu", 57344, 8192, 1), ("down", 8192, 28672, 0)]
for name, N, K, il in SH:
    per = N * K // 2 + (N // 16) * (K // 128) * 64
    ncopy = max(2, int(600e6 // per) + 1)
    lins = []
    for _ in range(ncopy):
        qw, qz, sc = synth_awq_tensors(N, K, 128, dev, gen)
        lins.append(PackedLinear.from_awq(qw, qz, sc, interleave=bool(il)))
    ln = lins[0]
    x = torch.randn(T, K, device=dev).to(dtype)
    out = torch.empty(max(ln.S * T * N, 1), dtype=torch.float32, device=dev)
    epi = 2 if il else 0

    def launch(l):
        _lib.call("umb_gemm", out, x, K, l.w, l.meta, T, N, K, l.awq, l.S, l.Rtb, epi, _lib.dtype_code(dtype))

    def timed(mode, pre_mb=0):
        reps = 30
        tot = 0.0
        for i in range(reps + 3):
            l = lins[0] if mode == "hot" else lins[i % ncopy]
            if mode == "pre":
                wv = l.w.view(torch.int32).reshape(-1)
                n = min(wv.numel(), pre_mb * (1 << 20) // 4)
                _ = wv[:n].sum()
                _ = l.meta.view(torch.int32).reshape(-1).sum()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); launch(l); e1.record()
            torch.cuda.synchronize()
            if i >= 3:
                tot += e0.elapsed_time(e1) * 1e3
        return tot / reps

    c, h = timed("cold"), timed("hot")
    pres = {mb: timed("pre", mb) for mb in (16, 32, 64, 128, 256)}
    print(f"{name:5s} {per/1e6:6.1f} MB  cold {c:6.2f} us  hot {h:6.2f} us  pre " + "  ".join(f"{mb}MB {v:6.2f}" for mb, v in pres.items()), flush=True)
    del lins
    torch.cuda.empty_cache()
