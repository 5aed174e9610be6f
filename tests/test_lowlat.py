"""Low-latency GEMM family (csrc/lowlat.hip, T <= 64): op-level parity through the C ABI.

Each epilogue against an fp32 torch restatement of the reference lines it fuses (llama.py:75-134,
model_utils.py:17-64, cache.py:53-65, awq_utils.py:63-86), bitwise batch invariance (a token's result must not depend
on how many rows share the launch, nor on the token tiling), and the FM activation layout.
Tolerances: dense -- fp32 accumulation order only (1e-4 of the row scale).  AWQ int4 -- the folded form computes
s * sum_k (q - z) x_k in fp32 without rounding (q - z) * s to fp16 first, so it differs from the dequantise-then-matmul
oracle by at most the fp16 rounding of the weights: 2^-11 relative per weight, < 2e-3 of the row scale after the sum.
"""
import ctypes as C

import pytest
import torch

from umbrella_amd.attn.cache import k_from_frag, vt_from_frag      # semantic views of the fragment-ordered KV caches

from oracle import ops as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import __graft_entry__ as ge
    ge.build()
    return torch.device("cuda:0")


def _rel(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-6))


def _fx(**kw):
    from umbrella_amd import _lib
    fx = _lib.UmbGemmLL()
    keep = []
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            keep.append(v)
            v = v.data_ptr()
        setattr(fx, k, v)
    fx._keep = keep
    return fx


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("T,K", [(1, 256), (13, 2048), (16, 128), (17, 256), (31, 512), (33, 256), (64, 2048)])
def test_fm_layout_roundtrip(dev, dtype, T, K):
    from umbrella_amd.models.llama import from_fm, to_fm
    x = torch.randn(T, K, device=dev).to(dtype)
    fm = to_fm(x)
    assert torch.equal(from_fm(fm, T, K), x)
    # element (t, k) sits at the documented offset (include/umbrella_hip.h)
    tt = (fm.numel() // K) // 16
    g = torch.Generator().manual_seed(T * K)
    for _ in range(16):
        t, k = int(torch.randint(0, T, (1,), generator=g)), int(torch.randint(0, K, (1,), generator=g))
        off = ((((k // 32) * tt + t // 16) * 64 + (k % 32 // 8) * 16 + t % 16) * 8) + k % 8
        assert fm[off] == x[t, k]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,K", [(512, 256), (256, 128), (3072, 2048), (2048, 8192), (4096 + 16, 1024), (128256, 256)])
@pytest.mark.parametrize("T", [1, 3, 13, 16, 17, 31, 40, 64])
def test_ll_gemm_dense(dev, dtype, N, K, T):
    from umbrella_amd.models.llama import PackedLinear
    gen = torch.Generator(device=dev).manual_seed(N + K + T)
    w = (torch.randn(N, K, device=dev, generator=gen) * 0.05).to(dtype)
    x = torch.randn(T, K, device=dev, generator=gen).to(dtype)
    lin = PackedLinear.from_dense(w)
    y = lin.apply_ll(x)
    ref = x.float() @ w.float().t()
    assert _rel(y, ref) < 1e-4, _rel(y, ref)
    yr = lin.apply_ll(x, round_out=True)
    assert torch.equal(yr, y.to(dtype).float())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("N,K", [(512, 256), (1024, 2048), (8192, 1024), (256, 128)])
@pytest.mark.parametrize("T", [1, 13, 16, 31, 40, 64])
def test_ll_gemm_awq(dev, dtype, N, K, T):
    from umbrella_amd.models.llama import PackedLinear
    from umbrella_amd.models.synthetic import synth_awq_tensors
    gen = torch.Generator(device=dev).manual_seed(N * 7 + K + T)
    qw, qz, sc = synth_awq_tensors(N, K, 128, dev, gen)
    lin = PackedLinear.from_awq(qw, qz, sc)
    x = (torch.randn(T, K, device=dev, generator=gen) * 0.5).to(dtype)
    y = lin.apply_ll(x)
    # exact real-valued model (q - z) * s in fp64, and the reference's fp16-dequantised weights
    wd = O.awq_dequant(qw.cpu(), qz.cpu(), sc.cpu(), 128)               # [K, N] fp16
    q = O.awq_unpack(qw.cpu().numpy())
    z = O.awq_unpack(qz.cpu().numpy())
    import numpy as np
    exact = (torch.from_numpy((q.astype(np.float64) - np.repeat(z.astype(np.float64), 128, axis=0)))
             * sc.cpu().double().repeat_interleave(128, dim=0))
    ref64 = (x.cpu().double() @ exact).float()
    ref16 = x.cpu().float() @ wd.float()
    # exact products; fp32 accumulation of (c_k + q_k) x_k with the c_k x_k part cancelled afterwards (c = 1024 / 64 in
    # fp16, 128 in bf16, where x also carries only 8 mantissa bits)
    assert _rel(y.cpu(), ref64) < (5e-5 if dtype == torch.float16 else 4e-4), _rel(y.cpu(), ref64)
    assert _rel(y.cpu(), ref16) < 2e-3, _rel(y.cpu(), ref16)          # vs awq_ext-style dequantised weights


def test_ll_gemm_batch_invariance(dev):
    """Row t of a T-row launch is bit-identical to the same row alone, for every token tiling (TT = 1, 2, 4) and both
    weight formats: greedy speculative decoding == greedy autoregressive decoding depends on it."""
    from umbrella_amd.models.llama import PackedLinear
    from umbrella_amd.models.synthetic import synth_awq_tensors
    gen = torch.Generator(device=dev).manual_seed(5)
    for awq, (N, K) in ((False, (3072, 2048)), (False, (2048, 8192)), (True, (8192, 2048)), (True, (1024, 1024))):
        if awq:
            lin = PackedLinear.from_awq(*synth_awq_tensors(N, K, 128, dev, gen))
            dtype = torch.float16
        else:
            dtype = torch.bfloat16
            lin = PackedLinear.from_dense((torch.randn(N, K, device=dev, generator=gen) * 0.05).to(dtype))
        x = torch.randn(64, K, device=dev, generator=gen).to(dtype)
        y = lin.apply_ll(x)
        for T in (1, 2, 13, 16, 17, 32, 33, 48):
            assert torch.equal(lin.apply_ll(x[:T].contiguous()), y[:T]), (awq, N, K, T)
        assert torch.equal(lin.apply_ll(x[5:6].contiguous()), y[5:6])
        assert float(lin.apply_ll(torch.zeros_like(x[:13])).abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("awq", [False, True])
@pytest.mark.parametrize("T", [1, 13, 31, 40])
def test_ll_silu_epilogue(dev, dtype, awq, T):
    """gate/up with interleaved rows + per-token 1/rms from the producer's sums of squares: act = SiLU(g) * u in FM
    layout (llama.py:107-110 behind model_utils.py:54-64)."""
    from umbrella_amd.models.llama import PackedLinear, from_fm
    from umbrella_amd.models.synthetic import synth_awq_tensors
    if awq and dtype != torch.float16:
        pytest.skip("AWQ checkpoints are fp16")
    gen = torch.Generator(device=dev).manual_seed(T + awq)
    H, I = 512, 1024
    if awq:
        qw, qz, sc = synth_awq_tensors(2 * I, H, 128, dev, gen)
        lin = PackedLinear.from_awq(qw, qz, sc, interleave=True)
        wfull = O.awq_dequant(qw.cpu(), qz.cpu(), sc.cpu(), 128).t().float().to(dev)      # [2I, H]
    else:
        wfull = (torch.randn(2 * I, H, device=dev, generator=gen) * 0.05).to(dtype)
        lin = PackedLinear.from_dense(wfull, interleave=True)
        wfull = wfull.float()
    h = torch.randn(T, H, device=dev, generator=gen)
    nw = 1 + 0.1 * torch.randn(H, device=dev, generator=gen)
    hw = (h.to(dtype).float() * nw.to(dtype).float()).to(dtype)                # producer side: h * w
    G = 12
    ssq = torch.zeros(T, 16, device=dev)
    part = (h.to(dtype).float() ** 2).view(T, 4, H // 4).sum(-1)
    ssq[:, :4] = part
    ssq[:, 4:G] = 0.0
    act_fm = torch.zeros(((T + 15) // 16 * 16 if T <= 32 else 64) * I, dtype=dtype, device=dev)
    fx = _fx(ssq_in=ssq, ssq_groups=G, ssq_in_stride=16, ssq_dim=float(H), eps=1e-5)
    lin.apply_ll(hw, fx=fx, epi=2, out=act_fm)
    got = from_fm(act_fm, T, I).float()
    inv = torch.rsqrt(part.sum(-1) / H + 1e-5)[:, None]
    y = (hw.float() @ wfull.t()) * inv
    g_, u_ = y[:, :I].to(dtype).float(), y[:, I:].to(dtype).float()
    ref = (torch.nn.functional.silu(g_).to(dtype).float() * u_).to(dtype).float()
    tol = 3e-3 if awq else 2e-2 if dtype == torch.bfloat16 else 3e-3
    assert _rel(got, ref) < tol, _rel(got, ref)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T", [1, 13, 31, 64])
def test_ll_residual_epilogue(dev, dtype, T):
    """h <- round(round(gemm) + h); hw = h * w_next in FM layout; one sum of squares per row group (llama.py:104,112)."""
    from umbrella_amd.models.llama import PackedLinear, from_fm, ll_plan, to_fm
    gen = torch.Generator(device=dev).manual_seed(T)
    N, K = 512, 1024
    w = (torch.randn(N, K, device=dev, generator=gen) * 0.03).to(dtype)
    lin = PackedLinear.from_dense(w)
    x = torch.randn(T, K, device=dev, generator=gen).to(dtype)
    h0 = torch.randn(T, N, device=dev, generator=gen).to(dtype)
    nw = (1 + 0.1 * torch.randn(N, device=dev, generator=gen)).to(dtype)
    R, WN, WK, NW = ll_plan(N, K, False)
    groups = N // 16 // R
    h = h0.clone()
    tt = 1 if T <= 16 else 2 if T <= 32 else 4
    hw_fm = torch.zeros(tt * 16 * N, dtype=dtype, device=dev)
    ssq = torch.full((T, groups + 3), -1.0, device=dev)
    fx = _fx(h=h, hw=hw_fm, norm_w=nw, ssq_out=ssq, ssq_out_stride=groups + 3)
    lin.apply_ll(x, fx=fx, epi=4)
    o = (x.float() @ w.float().t()).to(dtype).float()
    href = (o + h0.float()).to(dtype)
    assert float((h.float() - href.float()).abs().max()) <= 2 * float(href.float().abs().max()) * (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11)
    hwref = (h.float() * nw.float()).to(dtype)                            # from the kernel's own h: exact
    assert torch.equal(from_fm(hw_fm, T, N), hwref)
    sref = (h.float() ** 2).view(T, groups, 16 * R).sum(-1)
    assert _rel(ssq[:, :groups], sref) < 1e-5
    assert float(ssq[:, groups:].max()) == -1.0                          # nothing written past the row groups


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T", [1, 7, 13, 40])
@pytest.mark.parametrize("bias", [False, True])
def test_ll_qkv_epilogue_matches_split_path(dev, dtype, T, bias):
    """qkv GEMM + (bias) + RoPE + KV append in one launch == the split-K GEMM followed by umb_reduce_qkv_rope
    (epilogue.hip, already pinned against ops.npz): same q rows, same K / V^T cache contents, up to the fp32
    accumulation order of the GEMM (one 16-bit ulp after the roundings)."""
    from umbrella_amd import _lib
    from umbrella_amd.models.llama import PackedLinear
    gen = torch.Generator(device=dev).manual_seed(T + 100 * bias)
    Hq, Hkv, D, H, Lmax = 4, 2, 64, 256, 128
    N = (Hq + 2 * Hkv) * D
    w = (torch.randn(N, H, device=dev, generator=gen) * 0.05).to(dtype)
    lin = PackedLinear.from_dense(w, rope=(D, Hq + Hkv))
    x = torch.randn(T, H, device=dev, generator=gen).to(dtype)
    b = (torch.randn(N, device=dev, generator=gen) * 0.1).to(dtype) if bias else None
    pos = torch.randint(0, Lmax, (T,), device=dev, generator=gen, dtype=torch.int32)
    slot = torch.randperm(Lmax, device=dev, generator=gen)[:T].to(torch.int32)
    ang = torch.rand(Lmax, D, device=dev, generator=gen) * 6.28
    cos, sin = torch.cos(ang).to(dtype), torch.sin(ang).to(dtype)
    dt = _lib.dtype_code(dtype)

    def caches():
        return (torch.zeros(T, Hq, D, dtype=dtype, device=dev), torch.zeros(Hkv, Lmax, D, dtype=dtype, device=dev),
                torch.zeros(Hkv, D, Lmax + 32, dtype=dtype, device=dev))
    q1, k1, v1 = caches()
    part = torch.empty(lin.S, T, N, dtype=torch.float32, device=dev)
    _lib.call("umb_gemm", part, x, x.stride(0), lin.w, lin.meta, T, N, H, 0, lin.S, lin.R, 0, dt)
    _lib.call("umb_reduce_qkv_rope", part, lin.S, T, Hq, Hkv, D, Lmax, pos, slot, cos, sin, q1, k1, v1, 1, b, dt)
    q2, k2, v2 = caches()
    fx = _fx(pos=pos, slot=slot, cosT=cos, sinT=sin, q_out=q2, k_cache=k2, vt_cache=v2, Hq=Hq, Hkv=Hkv, D=D, Lmax=Lmax,
             **({"bias": b} if bias else {}))
    lin.apply_ll(x, fx=fx, epi=3)
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    for a_, b_ in ((q1, q2), (k1, k2), (v1, v2)):
        scale = float(a_.float().abs().max())
        assert float((a_.float() - b_.float()).abs().max()) <= 2 * ulp * scale, (float((a_.float() - b_.float()).abs().max()), scale)
    free = torch.ones(Lmax, dtype=torch.bool, device=dev)
    free[slot.long()] = False
    k2s, v2s = k_from_frag(k2), vt_from_frag(v2)              # semantic views of the fragment-ordered caches
    assert float(k2s[:, free].abs().max()) == 0.0 and float(v2s[:, :, :Lmax][:, :, free].abs().max()) == 0.0   # only slot[t] written
    assert float(k2s[:, ~free].abs().max()) > 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T", [1, 13, 33])
@pytest.mark.parametrize("bias", [False, True])
def test_split_qkv_on_fm_with_rms_in_reduce(dev, dtype, T, bias):
    """Mixed schedule for a qkv linear whose low-latency plan has a ragged last round (70B: 320 blocks): the split-K
    kernel reads the FM h*w activations and umb_reduce_qkv_rope2 applies 1/rms from the strided sums of squares.
    Same q / K / V^T as the one-launch low-latency epilogue (to accumulation order), and as the row-major split path
    on pre-normalised input."""
    from umbrella_amd import _lib
    from umbrella_amd.models.llama import PackedLinear, to_fm
    gen = torch.Generator(device=dev).manual_seed(T + 100 * bias)
    Hq, Hkv, D, H, Lmax = 4, 2, 64, 512, 128
    N = (Hq + 2 * Hkv) * D
    w = (torch.randn(N, H, device=dev, generator=gen) * 0.05).to(dtype)
    lin = PackedLinear.from_dense(w, rope=(D, Hq + Hkv))
    x = torch.randn(T, H, device=dev, generator=gen).to(dtype)
    b = (torch.randn(N, device=dev, generator=gen) * 0.1).to(dtype) if bias else None
    pos = torch.randint(0, Lmax, (T,), device=dev, generator=gen, dtype=torch.int32)
    slot = torch.randperm(Lmax, device=dev, generator=gen)[:T].to(torch.int32)
    ang = torch.rand(Lmax, D, device=dev, generator=gen) * 6.28
    cos, sin = torch.cos(ang).to(dtype), torch.sin(ang).to(dtype)
    G, stride = 70, 80                                         # more groups than one 64-lane pass
    ssq = torch.rand(T, stride, device=dev, generator=gen) * 4 + 1
    ssq[:, G:] = 1e9
    dt = _lib.dtype_code(dtype)

    def caches():
        return (torch.zeros(T, Hq, D, dtype=dtype, device=dev), torch.zeros(Hkv, Lmax, D, dtype=dtype, device=dev),
                torch.zeros(Hkv, D, Lmax + 32, dtype=dtype, device=dev))
    q1, k1, v1 = caches()
    S = max(lin.S, 2)
    part = torch.empty(S, T, N, dtype=torch.float32, device=dev)
    f1 = _lib.UmbGemmFused()
    f1.pad1 = 1
    _lib.call("umb_gemm_fused", part, to_fm(x), H, lin.w, lin.meta, T, N, H, 0, S, lin.R, 0, f1, dt)
    _lib.call("umb_reduce_qkv_rope2", part, S, T, Hq, Hkv, D, Lmax, pos, slot, cos, sin, q1, k1, v1, 1, b, ssq, G, stride,
              float(H), 1e-5, dt)
    q2, k2, v2 = caches()
    fx = _fx(pos=pos, slot=slot, cosT=cos, sinT=sin, q_out=q2, k_cache=k2, vt_cache=v2, Hq=Hq, Hkv=Hkv, D=D, Lmax=Lmax,
             ssq_in=ssq, ssq_groups=G, ssq_in_stride=stride, ssq_dim=float(H), eps=1e-5, **({"bias": b} if bias else {}))
    lin.apply_ll(x, fx=fx, epi=3)
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    for a_, b_ in ((q1, q2), (k1, k2), (v1, v2)):
        scale = float(a_.float().abs().max())
        assert scale > 0 and float((a_.float() - b_.float()).abs().max()) <= 2 * ulp * scale
    # fp32 statement of the same thing: (x @ W^T) * inv (+ b), rotated -- on the q rows
    inv = torch.rsqrt(ssq[:, :G].sum(1) / H + 1e-5)
    y = (x.float() @ w.float().t()) * inv[:, None] + (b.float() if bias else 0.0)
    qf = y[:, :Hq * D].view(T, Hq, D)
    c, s_ = cos[pos.long()].float()[:, None, :], sin[pos.long()].float()[:, None, :]
    rot = torch.cat([-qf[..., D // 2:], qf[..., :D // 2]], -1)
    ref = qf * c + rot * s_
    assert _rel(q1, ref) < (2e-2 if dtype == torch.bfloat16 else 3e-3)


@pytest.mark.parametrize("N,K,awq", [(3072, 2048, False), (2048, 8192, False), (16384, 2048, False), (128256, 2048, False),
                                     (10240, 8192, True), (8192, 8192, True), (57344, 8192, True), (8192, 28672, True),
                                     (6144, 4096, True), (4096, 14336, True), (28672, 4096, False)])
def test_ll_gemm_full_size_properties(dev, N, K, awq):
    """BASELINE shapes at full size (1B / 8B / 70B-AWQ linears): bitwise batch invariance, zero in -> zero out,
    linearity, and the fp32 reference on row slices."""
    from umbrella_amd.models.llama import PackedLinear
    from umbrella_amd.models.synthetic import synth_awq_tensors
    gen = torch.Generator(device=dev).manual_seed(N + K)
    if awq:
        qw, qz, sc = synth_awq_tensors(N, K, 128, dev, gen)
        lin = PackedLinear.from_awq(qw, qz, sc)
        dtype = torch.float16
    else:
        dtype = torch.bfloat16
        w = (torch.randn(N, K, device=dev, generator=gen) * 0.02).to(dtype)
        lin = PackedLinear.from_dense(w)
    x = (torch.randn(31, K, device=dev, generator=gen) * 0.5).to(dtype)
    y = lin.apply_ll(x)
    assert torch.equal(lin.apply_ll(x[:1].contiguous()), y[:1]) and torch.equal(lin.apply_ll(x[:13].contiguous()), y[:13])
    assert float(lin.apply_ll(torch.zeros_like(x)).abs().max()) == 0.0
    x2 = (torch.randn(31, K, device=dev, generator=gen) * 0.5).to(dtype)
    ysum = lin.apply_ll((x.float() + x2.float()).to(dtype))
    assert float((ysum - (y + lin.apply_ll(x2))).abs().max() / y.abs().max()) < (5e-3 if awq else 4e-2)
    for c0 in (0, N // 2 + 64, N - 64):
        if awq:
            ref = O.awq_linear(x.cpu().float(), qw[:, c0 // 8:(c0 + 64) // 8].cpu(), qz[:, c0 // 8:(c0 + 64) // 8].cpu(),
                               sc[:, c0:c0 + 64].cpu(), 128)
            assert _rel(y[:, c0:c0 + 64].cpu(), ref) < 2e-3
        else:
            ref = x.float() @ w[c0:c0 + 64].float().t()
            assert _rel(y[:, c0:c0 + 64], ref) < 1e-4
    # and it agrees with the split-K family to fp32 summation order
    ys = lin.apply(x)
    assert _rel(y, ys) < (2e-3 if awq else 1e-4)


@pytest.mark.parametrize("awq", [False, True])
@pytest.mark.parametrize("T", [1, 13, 31, 40])
def test_shared_kernel_with_fm_buffers(dev, awq, T):
    """The LDS-shared kernel of gemm.hip (large-N gate/up inside the low-latency schedule) reading x and writing the
    SiLU output in FM layout, 1/rms from strided sums of squares: same act as the row-major launch of the same kernel
    (bitwise) and as the low-latency kernel (to accumulation order / dequant form)."""
    from umbrella_amd import _lib
    from umbrella_amd.models.llama import PackedLinear, from_fm, to_fm
    from umbrella_amd.models.synthetic import synth_awq_tensors
    dtype = torch.float16
    gen = torch.Generator(device=dev).manual_seed(T + 7 * awq)
    H, I = 1024, 2048
    if awq:
        lin = PackedLinear.from_awq(*synth_awq_tensors(2 * I, H, 128, dev, gen), interleave=True)
    else:
        lin = PackedLinear.from_dense((torch.randn(2 * I, H, device=dev, generator=gen) * 0.03).to(dtype), interleave=True)
    x = torch.randn(T, H, device=dev, generator=gen).to(dtype)
    G, stride = 8, 12
    ssq = torch.rand(T, stride, device=dev, generator=gen) * 50 + 10
    ssq[:, G:] = 1e9                                         # beyond the groups: must not be read
    dt = _lib.dtype_code(dtype)
    tt = 1 if T <= 16 else 2 if T <= 32 else 4
    # row-major reference launch of the same kernel (contiguous sums of squares)
    act_rm = torch.zeros(T, I, dtype=dtype, device=dev)
    ssq_c = ssq[:, :G].contiguous()
    f0 = _lib.UmbGemmFused()
    f0.ssq_in, f0.ssq_groups, f0.ssq_dim, f0.eps = ssq_c.data_ptr(), G, float(H), 1e-5
    _lib.call("umb_gemm_fused", act_rm, x, H, lin.w, lin.meta, T, 2 * I, H, lin.awq, 1, lin.R, 2, f0, dt)
    # FM in / FM out, strided sums of squares
    act_fm = torch.zeros(tt * 16 * I, dtype=dtype, device=dev)
    f1 = _lib.UmbGemmFused()
    f1.ssq_in, f1.ssq_groups, f1.ssq_dim, f1.eps, f1.pad0, f1.pad1 = ssq.data_ptr(), G, float(H), 1e-5, stride, 3
    _lib.call("umb_gemm_fused", act_fm, to_fm(x), H, lin.w, lin.meta, T, 2 * I, H, lin.awq, 1, lin.R, 2, f1, dt)
    assert torch.equal(from_fm(act_fm, T, I), act_rm)
    # the low-latency kernel on the same buffers
    act_ll = torch.zeros(tt * 16 * I, dtype=dtype, device=dev)
    fx = _fx(ssq_in=ssq, ssq_groups=G, ssq_in_stride=stride, ssq_dim=float(H), eps=1e-5)
    lin.apply_ll(x, fx=fx, epi=2, out=act_ll)
    assert _rel(from_fm(act_ll, T, I), act_rm) < 4e-3

