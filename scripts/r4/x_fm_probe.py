"""Split-K int4 GEMMs of the 70B verify layer (T = 13) with the activations row-major (what the split schedule feeds them) vs
in FM / fragment order (what the low-latency schedule's buffers are): the kernel supports both (UmbGemmFused.pad1 bit 0).
Graph of 24 launches over rotating weight copies; prices 'FM activations on the split schedule' before building it."""
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
dt = _lib.dtype_code(dtype)
gen = torch.Generator(device=dev).manual_seed(0)
SH = [("qkv", 10240, 8192, 0), ("o", 8192, 8192, 0), ("gu", 57344, 8192, 1), ("down", 8192, 28672, 0)]
if os.environ.get("M") == "8b":
    SH = [("qkv", 6144, 4096, 0), ("o", 4096, 4096, 0), ("gu", 28672, 4096, 1), ("down", 4096, 14336, 0)]
for name, N, K, il in SH:
    per = N * K // 2
    ncopy = max(3, int(600e6 // per) + 1)
    lins = [PackedLinear.from_awq(*synth_awq_tensors(N, K, 128, dev, gen), interleave=bool(il)) for _ in range(ncopy)]
    ln = lins[0]
    S = 1 if il else max(1, min(ln.S, 4 if name == "o" else ln.S))
    x = torch.randn(T, K, device=dev).to(dtype)
    TT = _lib.load().umb_ll_token_tiles(T)
    xfm = torch.zeros(K // 32 * TT * 64 * 8, dtype=dtype, device=dev)
    _lib.call("umb_to_fm", xfm, x, T, K, dt)
    out = torch.empty(max(S * T * N, 1), dtype=torch.float32, device=dev)
    epi = 2 if il else 0
    res = {}
    for mode in ("row", "fm", "row", "fm"):
        fx = _lib.UmbGemmFused()
        fx.pad1 = 1 if mode == "fm" else 0
        xx = xfm if mode == "fm" else x

        def launch(i):
            l = lins[i % ncopy]
            _lib.call("umb_gemm_fused", out, xx, K, l.w, l.meta, T, N, K, 1, S, l.Rtb, epi, fx, dt)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            launch(0); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for i in range(24):
                    launch(i)
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(10):
                g.replay()
            e1.record(st); torch.cuda.synchronize()
        res.setdefault(mode, []).append(e0.elapsed_time(e1) * 1e3 / 240)
    print(f"{name:5s} N={N} K={K} S={S} T={T}: row-major x {min(res['row']):6.2f} us | FM x {min(res['fm']):6.2f} us", flush=True)
    del lins
    torch.cuda.empty_cache()
