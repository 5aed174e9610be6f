"""Kernel-level parity (GPU): every HIP op through the C ABI vs the CPU oracle / fp32 torch."""
import ctypes as C
import math

import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ops as O                                     # noqa: E402
from umbrella_amd.attn.cache import VT_PAD, k_from_frag, k_to_frag, vt_from_frag, vt_to_frag     # noqa: E402
# The caches are stored in MFMA fragment order (umbrella_amd/attn/cache.py); tests build and read them through these views.


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from umbrella_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-9))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,K", [(512, 256), (3072, 2048), (4096 + 16, 1024), (128256, 256)])
@pytest.mark.parametrize("T", [1, 5, 13, 16, 17, 33, 64, 70, 257, 300])
def test_gemm_dense(dev, dtype, N, K, T):
    from umbrella_amd.models.llama import PackedLinear
    g = torch.Generator(device="cpu").manual_seed(N + K + T)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dtype)
    x = torch.randn(T, K, generator=g).to(dtype)
    lin = PackedLinear.from_dense(w.to(dev))
    out = lin.apply(x.to(dev)).cpu()
    ref = x.float() @ w.float().t()
    assert _rel(out, ref) < 2e-3, (lin.R, lin.S, _rel(out, ref))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,K", [(512, 256), (1024, 2048), (28672 // 4, 1024)])
@pytest.mark.parametrize("T", [1, 13, 16, 31, 40, 64, 65, 200, 257])
def test_gemm_awq(dev, dtype, N, K, T):
    from umbrella_amd.models.awq_format import pack_rows
    from umbrella_amd.models.llama import PackedLinear
    rs = np.random.RandomState(N + K + T)
    q = rs.randint(0, 16, size=(K, N)).astype(np.uint8)
    z = rs.randint(0, 16, size=(K // 128, N)).astype(np.uint8)
    s = (rs.rand(K // 128, N) * 0.02 + 0.002).astype(np.float16)
    qw, qz, sc = torch.from_numpy(pack_rows(q)), torch.from_numpy(pack_rows(z)), torch.from_numpy(s)
    x = torch.from_numpy(rs.randn(T, K).astype(np.float32)).to(dtype)
    lin = PackedLinear.from_awq(qw.to(dev), qz.to(dev), sc.to(dev))
    out = lin.apply(x.to(dev)).cpu()
    W = (q.astype(np.float32) - np.repeat(z, 128, 0)) * np.repeat(s.astype(np.float32), 128, 0)   # exact (q-z)*s
    ref = x.float() @ torch.from_numpy(W)
    assert _rel(out, ref) < 2e-3, (lin.R, lin.S, _rel(out, ref))
    # and the oracle's AwqLinear restatement (fp16-rounded W) within fp16 weight rounding
    ref2 = O.awq_linear(x.float(), qw, qz, sc, 128)
    assert _rel(out, ref2) < 5e-3


def test_gemm_batch_invariance(dev):
    """A token's result must not depend on how many other tokens share the launch."""
    from umbrella_amd.models.llama import PackedLinear
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(3072, 2048, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    x = torch.randn(40, 2048, generator=g).to(torch.bfloat16).to(dev)
    lin = PackedLinear.from_dense(w)
    full = lin.apply(x)
    for T in (1, 13, 17, 33):
        assert torch.equal(lin.apply(x[:T].contiguous()), full[:T])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_rmsnorm(dev, dtype):
    from umbrella_amd import _lib
    g = torch.Generator().manual_seed(0)
    x = torch.randn(13, 2048, generator=g).to(dtype)
    w = (1 + 0.1 * torch.randn(2048, generator=g)).to(dtype)
    out = torch.empty_like(x, device=dev)
    _lib.call("umb_rmsnorm", out, x.to(dev), w.to(dev), 1e-5, 13, 2048, _lib.dtype_code(dtype))
    ref = O.rmsnorm(x, w, 1e-5)
    assert (out.cpu().float() - ref.float()).abs().max() <= 2 * torch.finfo(dtype).eps * ref.float().abs().max()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_reduce_residual_norm_and_silu(dev, dtype):
    from umbrella_amd import _lib
    g = torch.Generator().manual_seed(1)
    S, T, N = 3, 13, 512
    part = torch.randn(S, T, N, generator=g)
    res = torch.randn(T, N, generator=g).to(dtype)
    w = (1 + 0.1 * torch.randn(N, generator=g)).to(dtype)
    h = torch.empty(T, N, dtype=dtype, device=dev)
    xn = torch.empty(T, N, dtype=dtype, device=dev)
    _lib.call("umb_reduce_residual_norm", part.to(dev), S, T, N, res.to(dev), h, xn, w.to(dev), 1e-5, _lib.dtype_code(dtype))
    href = (part.sum(0).to(dtype) + res)
    assert (h.cpu().float() - href.float()).abs().max() <= 2 * torch.finfo(dtype).eps * href.float().abs().max()
    xref = O.rmsnorm(h.cpu(), w, 1e-5)
    assert (xn.cpu().float() - xref.float()).abs().max() <= 2 * torch.finfo(dtype).eps * xref.float().abs().max()
    I = 256
    act = torch.empty(T, I, dtype=dtype, device=dev)
    _lib.call("umb_reduce_silu_mul", part.to(dev), S, T, I, act, _lib.dtype_code(dtype))
    gate, up = part.sum(0)[:, :I].to(dtype), part.sum(0)[:, I:].to(dtype)
    aref = torch.nn.functional.silu(gate) * up
    assert (act.cpu().float() - aref.float()).abs().max() <= 4 * torch.finfo(dtype).eps * aref.float().abs().max()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T", [1, 13, 31, 64])
def test_deferred_norm_schedule_matches_the_in_place_one(dev, dtype, T):
    """model.hip layer_split_defer (the default for <= 64-row forwards of split-schedule models with H % 512 == 0) against the
    one-block-per-row reduce that normalises in place (UMB_DEFER_NORM=0), 3 layers of the 8B shape (H = 4096): logits agree to
    16-bit rounding and pick the same arg-max on every row -- row-major operands (T < 16) and FM-ordered ones (16 .. 64)."""
    import copy
    from umbrella_amd.models.config import KNOWN
    from umbrella_amd.models.llama import Llama
    cfg = copy.copy(KNOWN["meta-llama/Llama-3.1-8B-Instruct"])
    cfg.num_hidden_layers = 3
    outs = []
    for defer in ("1", "0"):
        os.environ["UMB_DEFER_NORM"] = defer
        try:
            m = Llama("meta-llama/Llama-3.1-8B-Instruct", max_length=256, device=str(dev), dtype=dtype, config=cfg, sched="split", seed=2)
            m.alloc()
        finally:
            os.environ.pop("UMB_DEFER_NORM", None)
        assert m._ws.defer_norm == int(defer)
        g = torch.Generator().manual_seed(T)
        ids = torch.randint(3, 128000, (40 + T,), generator=g, dtype=torch.int32).to(dev)
        m.prefill_tokens(ids[:40], 0)
        pos = torch.arange(40, 40 + T, dtype=torch.int32, device=dev)
        pre = torch.tensor([40], dtype=torch.int32, device=dev)
        m.forward_explicit(ids[40:].contiguous(), pos, pos, pre, head_from=0)
        torch.cuda.synchronize()
        outs.append((m.logits_buffer[:T].clone(), m.hidden_buffer[:T].clone()))
        del m
    (la, ha), (lb, hb) = outs
    assert torch.isfinite(la).all()
    tol = (0.08 if dtype == torch.bfloat16 else 0.012) * float(lb.abs().max())
    assert float((la - lb).abs().max()) <= tol, (float((la - lb).abs().max()), tol)
    top2 = lb.topk(2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 2 * tol
    assert torch.equal(la.argmax(-1)[clear], lb.argmax(-1)[clear])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,S,N", [(13, 4, 8192), (31, 8, 4096), (1, 2, 2048), (64, 17, 4096), (5, 1, 2048)])
def test_reduce_residual_hw_many_blocks_per_row(dev, dtype, T, S, N):
    """umb_reduce_residual_hw (round 6: the split schedule's residual reduce with the RMSNorm deferred, N / 512 blocks per row):
    h is BIT-identical to umb_reduce_residual_norm's (same split order, same two roundings); hw = round(h * w) exactly, row-major
    and in FM order; the N / 512 sums of squares add up to sum h^2 (fp32 order only); and what the consumer makes of them --
    rsqrt(sum / N + eps) * (hw @ W) -- equals the GEMM over the normalised row (the reference's layer_norm then linear,
    model_utils.py:54-64, llama.py:87-91) up to 16-bit rounding."""
    from umbrella_amd import _lib
    g = torch.Generator().manual_seed(T * 31 + S)
    dt = _lib.dtype_code(dtype)
    part = torch.randn(S, T, N, generator=g).to(dev)
    res = torch.randn(T, N, generator=g).to(dtype).to(dev)
    w = (1 + 0.1 * torch.randn(N, generator=g)).to(dtype).to(dev)
    h0, x0 = torch.empty(T, N, dtype=dtype, device=dev), torch.empty(T, N, dtype=dtype, device=dev)
    _lib.call("umb_reduce_residual_norm", part, S, T, N, res, h0, x0, w, 1e-5, dt)
    G, stride = N // 512, 64
    for fm in (0, _lib.load().umb_ll_token_tiles(T)):
        h1 = torch.empty(T, N, dtype=dtype, device=dev)
        hw = torch.zeros(max(fm * 16, T) * N, dtype=dtype, device=dev)
        ssq = torch.full((T, stride), float("nan"), device=dev)
        _lib.call("umb_reduce_residual_hw", part, S, T, N, res, h1, hw, w, ssq, stride, fm, dt)
        torch.cuda.synchronize()
        assert torch.equal(h1.view(torch.int16), h0.view(torch.int16))
        if fm:
            back = torch.empty(T, N, dtype=dtype, device=dev)
            _lib.call("umb_from_fm", back, hw, T, N, dt)
            hw_rm = back
        else:
            hw_rm = hw[:T * N].view(T, N)
        want = (h0.float() * w.float()).to(dtype)
        assert torch.equal(hw_rm.view(torch.int16), want.view(torch.int16))
        assert torch.isnan(ssq[:, G:]).all() and torch.isfinite(ssq[:, :G]).all()          # exactly N / 512 groups written
        tot = ssq[:, :G].double().sum(1)
        ref = (h0.double() ** 2).sum(1)
        assert ((tot - ref).abs() <= 1e-5 * ref).all()
        per = (h0.float() ** 2).view(T, G, 512).sum(2)
        assert ((ssq[:, :G] - per).abs() <= 1e-5 * per).all()
        # the consumer's view: (1 / rms) (hw @ W^T) against (rmsnorm(h) * w) @ W^T
        Wm = (torch.randn(64, N, generator=torch.Generator().manual_seed(3)) * 0.02).to(dev)
        inv = torch.rsqrt(tot / N + 1e-5).float()
        a = (hw_rm.float() @ Wm.T) * inv[:, None]
        b = x0.float() @ Wm.T
        assert (a - b).abs().max() <= 8 * torch.finfo(dtype).eps * b.abs().max()
    # no hw / no ssq (a pipeline stage's last layer hands on h only)
    h2 = torch.empty(T, N, dtype=dtype, device=dev)
    _lib.call("umb_reduce_residual_hw", part, S, T, N, res, h2, None, None, None, 0, 0, dt)
    assert torch.equal(h2.view(torch.int16), h0.view(torch.int16))
    lib = _lib.load()
    assert lib.umb_reduce_residual_hw(part.data_ptr(), S, T, 1000, res.data_ptr(), h2.data_ptr(), None, None, None, 0, 0, dt, None) != 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,S,N", [(13, 4, 8192), (31, 8, 4096), (1, 2, 2048), (64, 13, 4096), (40, 9, 8192)])
def test_fm_ordered_norm_outputs_equal_the_row_major_ones(dev, dtype, T, S, N):
    """umb_rmsnorm_fm / umb_reduce_residual_norm_fm (the split schedule's <= 64-row forwards keep their GEMM operands in FM order):
    the same bits as the row-major forms, at the FM positions (umb_from_fm converts back); h_out stays row-major."""
    from umbrella_amd import _lib
    g = torch.Generator().manual_seed(T + S)
    dt = _lib.dtype_code(dtype)
    tt = _lib.load().umb_ll_token_tiles(T)
    part = torch.randn(S, T, N, generator=g).to(dev)
    res = torch.randn(T, N, generator=g).to(dtype).to(dev)
    w = (1 + 0.1 * torch.randn(N, generator=g)).to(dtype).to(dev)
    h0, x0 = torch.empty(T, N, dtype=dtype, device=dev), torch.empty(T, N, dtype=dtype, device=dev)
    _lib.call("umb_reduce_residual_norm", part, S, T, N, res, h0, x0, w, 1e-5, dt)
    h1 = torch.empty(T, N, dtype=dtype, device=dev)
    xfm = torch.zeros(tt * 16 * N, dtype=dtype, device=dev)
    _lib.call("umb_reduce_residual_norm_fm", part, S, T, N, res, h1, xfm, w, 1e-5, tt, dt)
    back = torch.empty(T, N, dtype=dtype, device=dev)
    _lib.call("umb_from_fm", back, xfm, T, N, dt)
    assert torch.equal(h1, h0) and torch.equal(back, x0)
    r0 = torch.empty(T, N, dtype=dtype, device=dev)
    _lib.call("umb_rmsnorm", r0, h0, w, 1e-5, T, N, dt)
    xfm.zero_()
    _lib.call("umb_rmsnorm_fm", xfm, h0, w, 1e-5, T, N, tt, dt)
    _lib.call("umb_from_fm", back, xfm, T, N, dt)
    assert torch.equal(back, r0)
    assert _lib.load().umb_rmsnorm_fm(None, None, None, C.c_float(1e-5), 17, N, 1, dt, None) == -22       # 17 rows do not fit one tile


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("S,N", [(1, 2048), (4, 8192), (8, 8192), (8, 5120), (16, 2048), (13, 4096), (9, 8192), (16, 8192),
                                 (3, 12288)])
def test_reduce_residual_norm_paths(dev, dtype, S, N):
    """Row reduce over every code path: whole row in registers (<= 8 splits x 2 column groups, <= 16 splits x 1 group)
    and the generic loop; with / without residual and norm; splits summed in order 0..S-1 (bit-exact vs fp32 torch
    summed the same way)."""
    from umbrella_amd import _lib
    g = torch.Generator().manual_seed(S * 100003 + N)
    T = 5
    part = torch.randn(S, T, N, generator=g)
    res = torch.randn(T, N, generator=g).to(dtype)
    w = (1 + 0.1 * torch.randn(N, generator=g)).to(dtype)
    acc = part[0].clone()
    for s in range(1, S):
        acc += part[s]                                           # the kernel's summation order
    for with_res, with_norm in ((True, True), (False, True), (True, False)):
        h = torch.zeros(T, N, dtype=dtype, device=dev)
        xn = torch.zeros(T, N, dtype=dtype, device=dev)
        _lib.call("umb_reduce_residual_norm", part.to(dev), S, T, N, res.to(dev) if with_res else None, h,
                  xn if with_norm else None, w.to(dev) if with_norm else None, 1e-5, _lib.dtype_code(dtype))
        href = acc.to(dtype)
        if with_res:
            href = href + res
        assert torch.equal(h.cpu(), href), (S, N, with_res)
        if with_norm:
            xref = O.rmsnorm(h.cpu(), w, 1e-5)
            assert (xn.cpu().float() - xref.float()).abs().max() <= 2 * torch.finfo(dtype).eps * xref.float().abs().max()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_qkv_rope_kv_append(dev, dtype):
    from umbrella_amd import _lib
    g = torch.Generator().manual_seed(2)
    T, Hq, Hkv, D, Lmax = 7, 4, 2, 64, 64
    N = (Hq + 2 * Hkv) * D
    part = torch.randn(2, T, N, generator=g)
    pos = torch.tensor([9, 10, 10, 11, 11, 11, 12], dtype=torch.int32)
    slot = torch.arange(20, 27, dtype=torch.int32)
    inv = 1.0 / (500000.0 ** (torch.arange(0, D, 2).float() / D))
    cos, sin = O.rope_cache(inv, 1.0, Lmax, dtype)
    q = torch.zeros(T, Hq, D, dtype=dtype, device=dev)
    kc = torch.zeros(Hkv, Lmax, D, dtype=dtype, device=dev)
    vt = torch.zeros(Hkv, D, Lmax + VT_PAD, dtype=dtype, device=dev)
    _lib.call("umb_reduce_qkv_rope", part.to(dev), 2, T, Hq, Hkv, D, Lmax, pos.to(dev), slot.to(dev), cos.to(dev),
              sin.to(dev), q, kc, vt, 0, None, _lib.dtype_code(dtype))
    full = part.sum(0).to(dtype)
    qr, kr, vr = full[:, :Hq * D].view(T, Hq, D), full[:, Hq * D:(Hq + Hkv) * D].view(T, Hkv, D), full[:, (Hq + Hkv) * D:].view(T, Hkv, D)
    qe, ke = O.apply_rope(qr, kr, cos, sin, pos.long())
    eps = torch.finfo(dtype).eps
    assert (q.cpu().float() - qe.float()).abs().max() <= 2 * eps * qe.float().abs().max()
    kc, vt = k_from_frag(kc.cpu()), vt_from_frag(vt.cpu())
    kgot = kc[:, 20:27].permute(1, 0, 2)
    assert (kgot.float() - ke.float()).abs().max() <= 2 * eps * ke.float().abs().max()
    vgot = vt[:, :, 20:27].permute(2, 0, 1)
    assert torch.equal(vgot, vr)
    assert kc[:, :20].abs().max() == 0 and vt[:, :, 27:].abs().max() == 0


def _tree_mask(T, rs):
    par = [0] + [int(rs.randint(0, i)) for i in range(1, T)]
    m = torch.zeros(T, T, dtype=torch.bool)
    for i in range(T):
        j = i
        while True:
            m[i, j] = True
            if j == 0:
                break
            j = par[j]
    return m


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Hq,Hkv,D", [(4, 2, 64), (8, 1, 128), (32, 8, 64), (4, 4, 32)])
@pytest.mark.parametrize("T,prefix", [(1, 0), (1, 300), (13, 5), (13, 777), (31, 64), (70, 129), (13, 2500), (5, 4200),
                                      (13, 1011), (13, 1012), (31, 1500)])     # narrow launches: one span up to 1024 keys, then 512-key spans
@pytest.mark.parametrize("path", ["single", "split", "spans"])
def test_tree_attention(dev, dtype, Hq, Hkv, D, T, prefix, path):
    """Tree-masked and causal attention vs the oracle through the three kernel paths: one launch, one 2048-key span
    (merge inside the block); key splits + combine kernel (long context, no counters); one launch with several
    spans merged by the last-arriving block (long context, counters)."""
    if path == "single" and prefix + T > 1024:
        pytest.skip("context longer than this path's Lmax")
    from umbrella_amd import _lib
    from umbrella_amd.models.llama import pack_mask_bits
    rs = np.random.RandomState(T * 1000 + prefix + D)
    g = torch.Generator().manual_seed(T + prefix)
    Lmax, chunk = (1024, 256) if path == "single" else (8192, 512)
    splits = Lmax // chunk
    S = prefix + T
    q = torch.randn(T, Hq, D, generator=g).to(dtype)
    k = torch.randn(S, Hkv, D, generator=g).to(dtype)
    v = torch.randn(S, Hkv, D, generator=g).to(dtype)
    tm = _tree_mask(T, rs)
    mask = torch.cat([torch.ones(T, prefix, dtype=torch.bool), tm], dim=1)
    ref = O.masked_attention(q.float(), k.float(), v.float(), mask)
    kc = torch.zeros(Hkv, Lmax, D, dtype=dtype)
    vt = torch.zeros(Hkv, D, Lmax + VT_PAD, dtype=dtype)
    kc[:, :S] = k.permute(1, 0, 2)
    vt[:, :, :S] = v.permute(1, 2, 0)
    # stale garbage after the valid region must be ignored
    kc[:, S:S + 40] = 7.0
    vt[:, :, S:min(S + 40, Lmax)] = -9.0
    kc, vt = k_to_frag(kc), vt_to_frag(vt)
    bits = pack_mask_bits(tm).to(dev)
    out = torch.empty(T, Hq, D, dtype=dtype, device=dev)
    po = torch.empty(splits * T * Hq * D, dtype=torch.float32, device=dev)
    pml = torch.empty(splits * T * Hq * 2, dtype=torch.float32, device=dev)
    pre = torch.tensor([prefix], dtype=torch.int32, device=dev)
    nqt = (T * (Hq // Hkv) + 15) // 16
    counters = torch.zeros(Hkv * nqt, dtype=torch.int32, device=dev) if path == "spans" else None
    tol = 0.03 if dtype == torch.bfloat16 else 0.004
    _lib.call("umb_tree_attn", out, q.to(dev), kc.to(dev), vt.to(dev), po, pml, pre, bits, bits.shape[1], T, T, Hq, Hkv,
              D, Lmax, chunk, splits, 1.0 / math.sqrt(D), counters, _lib.dtype_code(dtype))
    if counters is not None:
        assert int(counters.abs().sum()) == 0                                # arrival counters reset themselves
    err = (out.cpu().float() - ref).abs().max()
    assert err < tol, float(err)
    # causal mode (mask_bits = NULL): row t sees prefix + new keys 0..t
    out.zero_()
    _lib.call("umb_tree_attn", out, q.to(dev), kc.to(dev), vt.to(dev), po, pml, pre, None, 0, T, T, Hq, Hkv, D, Lmax,
              chunk, splits, 1.0 / math.sqrt(D), counters, _lib.dtype_code(dtype))
    cm = torch.cat([torch.ones(T, prefix, dtype=torch.bool), torch.tril(torch.ones(T, T, dtype=torch.bool))], dim=1)
    ref = O.masked_attention(q.float(), k.float(), v.float(), cm)
    err = (out.cpu().float() - ref).abs().max()
    assert err < tol, float(err)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,prefix,Lmax", [(257, 128, 1024), (300, 77, 4096), (769, 128, 4096), (200, 2300, 4096)])
@pytest.mark.parametrize("kind", ["tree", "arbitrary"])
@pytest.mark.parametrize("Hq,Hkv", [(16, 2), (64, 8)])
def test_tree_attention_wide(dev, dtype, T, prefix, Lmax, kind, Hq, Hkv):
    """Wide trees (several mask words, several query tiles per kv head) through the single-launch kernel, which ends
    each query tile at the last key its rows can see.  "arbitrary": a random mask in which rows also see LATER keys
    (not a tree) -- the end of a tile is taken from the mask bits, not assumed from the row index."""
    from umbrella_amd import _lib
    from umbrella_amd.models.llama import pack_mask_bits
    D = 128                                  # 2 or 8 kv heads x 17 .. 385 query tiles: 8, 2 and 1 waves per tile
    rs = np.random.RandomState(T + prefix)
    g = torch.Generator().manual_seed(T * 7 + prefix)
    S = prefix + T
    q = torch.randn(T, Hq, D, generator=g).to(dtype)
    k = torch.randn(S, Hkv, D, generator=g).to(dtype)
    v = torch.randn(S, Hkv, D, generator=g).to(dtype)
    if kind == "tree":
        tm = _tree_mask(T, rs)
    else:
        tm = torch.from_numpy(rs.rand(T, T) < 0.05)
        tm[torch.arange(T), torch.arange(T)] = True
        tm[5] = False                                                         # a row that sees prefix keys only
        tm[5, 0] = prefix == 0
        tm[T // 2, T - 1] = True                                              # ... and one that sees the very last key
    mask = torch.cat([torch.ones(T, prefix, dtype=torch.bool), tm], dim=1)
    ref = O.masked_attention(q.float(), k.float(), v.float(), mask)
    kc = torch.zeros(Hkv, Lmax, D, dtype=dtype)
    vt = torch.zeros(Hkv, D, Lmax + VT_PAD, dtype=dtype)
    kc[:, :S] = k.permute(1, 0, 2)
    vt[:, :, :S] = v.permute(1, 2, 0)
    kc, vt = k_to_frag(kc), vt_to_frag(vt)
    bits = pack_mask_bits(tm).to(dev)
    spans = (Lmax + 2047) // 2048
    out = torch.empty(T, Hq, D, dtype=dtype, device=dev)
    po = torch.empty(spans * T * Hq * D, dtype=torch.float32, device=dev)
    pml = torch.empty(spans * T * Hq * 2, dtype=torch.float32, device=dev)
    pre = torch.tensor([prefix], dtype=torch.int32, device=dev)
    counters = torch.zeros(Hkv * ((T * (Hq // Hkv) + 15) // 16), dtype=torch.int32, device=dev)
    _lib.call("umb_tree_attn", out, q.to(dev), kc.to(dev), vt.to(dev), po, pml, pre, bits, bits.shape[1], T, T, Hq, Hkv,
              D, Lmax, 2048, spans, 1.0 / math.sqrt(D), counters, _lib.dtype_code(dtype))
    assert int(counters.abs().sum()) == 0
    tol = 0.03 if dtype == torch.bfloat16 else 0.004
    err = (out.cpu().float() - ref).abs().max()
    assert err < tol, float(err)


def test_argmax_and_topk(dev):
    from umbrella_amd import _lib
    g = torch.Generator().manual_seed(4)
    for V in (512, 128256):
        logits = torch.randn(7, V, generator=g)
        logits[2, 100] = logits[2, 50] = 9.0                    # tie -> lower index
        logits = logits.bfloat16().float()                       # bf16-rounded logits produce many exact ties
        d = logits.to(dev)
        am = torch.empty(7, dtype=torch.int32, device=dev)
        _lib.call("umb_argmax_rows", am, d, 7, V)
        assert am.cpu().tolist() == logits.argmax(-1).tolist()
        assert am.cpu()[2] == 50
        for k in (1, 3, 8, 24, 32):
            idx = torch.empty(7, k, dtype=torch.int32, device=dev)
            val = torch.empty(7, k, dtype=torch.float32, device=dev)
            _lib.call("umb_topk_rows", idx, val, d, 7, V, k, None, None, None, None)
            tv, _ = logits.topk(k, dim=-1)
            assert torch.equal(val.cpu(), tv)
            assert torch.equal(torch.gather(logits, 1, idx.cpu().long()), tv)
            # ties in index order, no duplicates
            for r in range(7):
                row = idx.cpu()[r].tolist()
                assert len(set(row)) == k
                for a in range(k - 1):
                    if tv[r, a] == tv[r, a + 1]:
                        assert row[a] < row[a + 1]
    # pathological: all equal -> the k lowest indices
    d = torch.zeros(2, 5000, device=dev)
    idx = torch.empty(2, 8, dtype=torch.int32, device=dev)
    _lib.call("umb_topk_rows", idx, None, d, 2, 5000, 8, None, None, None, None)
    assert idx.cpu().tolist() == [list(range(8))] * 2


@pytest.mark.parametrize("V", [128256, 151936, 32000, 16388])
def test_topk_split_matches_single_block(dev, V):
    """umb_topk_rows_ws (vocabulary of a row split over 16 blocks, last arriver merges) == umb_topk_rows bit for bit:
    values, indices, tie order, Sequoia child placement; counters reset themselves; flat rows (all ties) included."""
    from umbrella_amd import _lib
    g = torch.Generator().manual_seed(V)
    rows = 9
    logits = torch.randn(rows, V, generator=g).bfloat16().float()
    logits[3] = 0.0                                              # all equal: the k lowest indices
    logits[4, V - 1] = logits[4, 0] = 11.0                        # tie across the first and the last slice
    logits[5, :] = -float("inf")
    logits[5, 7] = 1.0                                           # one finite entry, the rest -inf (ties among -inf)
    d = logits.to(dev)
    for k in (1, 4, 32, 64):
        ws = torch.zeros(4096 + rows * 16 * k * 8, dtype=torch.uint8, device=dev)
        ia, ib = (torch.empty(rows, k, dtype=torch.int32, device=dev) for _ in range(2))
        va, vb = (torch.empty(rows, k, dtype=torch.float32, device=dev) for _ in range(2))
        _lib.call("umb_topk_rows", ia, va, d, rows, V, k, None, None, None, None)
        for _ in range(2):                                       # second launch: the counters have reset themselves
            ib.fill_(-1)
            _lib.call("umb_topk_rows_ws", ib, vb, d, rows, V, k, None, None, None, None, ws, ws.numel())
            assert torch.equal(ia.cpu(), ib.cpu()), k
            assert torch.equal(va.cpu(), vb.cpu()), k
        assert int(ws[:4096].sum()) == 0
    # child placement through the split kernel
    k = 4
    ws = torch.zeros(4096 + rows * 16 * k * 8, dtype=torch.uint8, device=dev)
    cs = torch.arange(rows, dtype=torch.int32, device=dev) * 4 + 1
    cc = torch.tensor([4, 0, 2, 1, 3, 4, 0, 1, 2], dtype=torch.int32, device=dev)
    n = torch.tensor([10], dtype=torch.int32, device=dev)
    ta, tb = (torch.full((64,), -7, dtype=torch.int32, device=dev) for _ in range(2))
    _lib.call("umb_topk_rows", None, None, d, rows, V, k, ta, n, cs, cc)
    _lib.call("umb_topk_rows_ws", None, None, d, rows, V, k, tb, n, cs, cc, ws, ws.numel())
    assert torch.equal(ta.cpu(), tb.cpu())
    # too small a workspace falls back to the single-block kernel
    ib = torch.empty(rows, k, dtype=torch.int32, device=dev)
    ia = torch.empty(rows, k, dtype=torch.int32, device=dev)
    _lib.call("umb_topk_rows", ia, None, d, rows, V, k, None, None, None, None)
    _lib.call("umb_topk_rows_ws", ib, None, d, rows, V, k, None, None, None, None, ws, 100)
    assert torch.equal(ia.cpu(), ib.cpu())


def test_topk_places_sequoia_children(dev):
    from umbrella_amd import _lib
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(3, 512, generator=g)
    tokens = torch.zeros(64, dtype=torch.int32, device=dev)
    n = torch.tensor([10], dtype=torch.int32, device=dev)
    cs = torch.tensor([4, 6, 7], dtype=torch.int32, device=dev)
    cc = torch.tensor([2, 1, 0], dtype=torch.int32, device=dev)
    _lib.call("umb_topk_rows", None, None, logits.to(dev), 3, 512, 2, tokens, n, cs, cc)
    exp = O.topk_flatten_gather(logits, 2, torch.tensor([0, 1, 2]))
    assert tokens.cpu()[14:17].tolist() == exp.tolist()
    assert tokens.cpu()[17:].abs().sum() == 0 and tokens.cpu()[:14].abs().sum() == 0


def _oracle_filtered(logits, hist, penalty, temperature, topk, topp):
    """The reference's filtered target distribution (static:298-310 / dynamic:266-281) through the oracle ops."""
    lg = logits.clone()
    if penalty > 1.01:
        lg = O.repetition_penalty(hist[None].expand(lg.shape[0], -1), lg, penalty)
    pen = lg.clone()
    lg = O.keep_topk(lg, topk)
    return pen, O.top_p_renorm(torch.softmax(lg / temperature, dim=-1), topp)


@pytest.mark.parametrize("V,topk,penalty", [(512, 8, 1.0), (128256, 32, 1.05), (5001, 32, 1.3), (128256, 64, 1.0)])
def test_sample_rows_distribution(dev, V, topk, penalty):
    """umb_sample_rows: the filtered, renormalised distribution equals the oracle's (penalty with duplicate
    history tokens, top-k, temperature, top-p); draws land in its support and follow it (chi-square)."""
    from umbrella_amd import _lib
    g = torch.Generator().manual_seed(V + topk)
    rows, n = 6, 200
    logits = torch.randn(rows, V, generator=g) * 3.0
    hist = torch.randint(0, V, (n + 1,), generator=g)
    hist[5:40] = hist[0]                                   # duplicates: penalised once
    hist[50:60] = logits[0].topk(10)[1]                    # make the penalty matter for row 0's head
    temperature, topp = 0.6, 0.9
    pen, p_ref = _oracle_filtered(logits, hist, penalty, temperature, topk, topp)
    tokens = torch.zeros(n + 64, dtype=torch.int32); tokens[:n + 1] = hist.int()
    tokens, nd = tokens.to(dev), torch.tensor([n], dtype=torch.int32, device=dev)
    seed = torch.tensor([1234], dtype=torch.int64, device=dev)
    sampled = torch.zeros(rows, dtype=torch.int32, device=dev)
    dk = 128
    di = torch.zeros(rows, dk, dtype=torch.int32, device=dev); dp = torch.zeros(rows, dk, device=dev)
    d = logits.to(dev).clone()
    _lib.call("umb_sample_rows", sampled, d, rows, V, tokens, nd, penalty, temperature, topk, topp, seed, dk, di, dp)
    torch.cuda.synchronize()
    assert torch.equal(d.cpu(), pen), "in-place repetition penalty differs from the oracle's gather/scatter"
    di, dp = di.cpu(), dp.cpu()
    for r in range(rows):
        got = torch.zeros(V)
        keep = di[r] >= 0
        got[di[r][keep].long()] = dp[r][keep]
        assert torch.allclose(got, p_ref[r], atol=2e-6, rtol=1e-4), (r, (got - p_ref[r]).abs().max())
        assert p_ref[r, sampled[r].item()] > 0
    # chi-square on 4000 draws of row 0 (fresh n => fresh draw each launch)
    draws = torch.zeros(4000, dtype=torch.int32, device=dev)
    for i in range(4000):
        nd.fill_(n)            # history unchanged ...
        seed.fill_(i)          # ... new stream
        _lib.call("umb_sample_rows", draws[i:i + 1], d[:1].clone() if penalty <= 1.01 else logits[:1].to(dev).clone(),
                  1, V, tokens, nd, penalty, temperature, topk, topp, seed, 0, None, None)
    cnt = torch.bincount(draws.cpu().long(), minlength=V).float()
    exp = p_ref[0] * 4000
    assert cnt[exp == 0].sum() == 0
    big = exp >= 5
    chi = (((cnt - exp) ** 2)[big] / exp[big]).sum().item()
    dof = int(big.sum()) - 1
    assert chi < dof + 5 * (2 * max(dof, 1)) ** 0.5 + 10, (chi, dof)


@pytest.mark.parametrize("V,topk,penalty", [(512, 8, 1.0), (128256, 32, 1.05), (5001, 32, 1.3), (128256, 20, 1.0)])
def test_sample_rows_uniform_equals_the_reference_sampler(dev, V, topk, penalty):
    """umb_sample_rows_uniform == the reference's static-engine draw (static:298-310): repetition penalty -> logits / T ->
    flashinfer.sampling.top_k_top_p_sampling_from_logits(logits, uniform_samples, top_k, top_p) as restated in
    oracle/ops.py (0.2.x rejection sampler, 3 rounds, vocabulary-order cumulative sums): the SAME uniforms give the SAME
    token, row for row -- including rows driven to the tail of the kept set in every round (rejections; when the rounds
    run out the last round's token is returned) and uniforms at the edges of [0, 1).  A row may differ only where u q sits within float rounding of a cumulative-sum boundary."""
    from umbrella_amd import _lib
    g = torch.Generator().manual_seed(7 * V + topk)
    rows, n = 96, 150
    logits = torch.randn(rows, V, generator=g) * 2.5
    logits[1] = logits[1] * 0.2                              # a flat row: rejections are common (nucleus is wide)
    hist = torch.randint(0, V, (n + 1,), generator=g)
    hist[40:50] = logits[0].topk(10)[1]
    temperature, topp = 0.6, 0.9
    u = torch.rand(3, rows, generator=g)
    u[:, 2] = torch.tensor([0.0, 0.0, 0.0]); u[:, 3] = torch.tensor([0.999999, 0.999999, 0.999999])
    lg = logits.clone()
    if penalty > 1.01:
        lg = O.repetition_penalty(hist[None].expand(rows, -1), lg, penalty)
    # rows 4..11: uniforms near 1 in every round walk to the tail of the kept set, where a draw is rejected while more
    # than top_p of the mass lies strictly above it -- the rounds are used up and the last round's token is returned
    u[:, 4:12] = 0.97 + 0.03 * torch.rand(3, 8, generator=g)
    want, ok = O.top_k_top_p_sampling_from_logits(lg / temperature, u, topk, topp)
    tokens = torch.zeros(n + 64, dtype=torch.int32); tokens[:n + 1] = hist.int()
    tokens, nd = tokens.to(dev), torch.tensor([n], dtype=torch.int32, device=dev)
    sampled = torch.zeros(rows, dtype=torch.int32, device=dev)
    _lib.call("umb_sample_rows_uniform", sampled, logits.to(dev).clone(), rows, V, tokens, nd, penalty, temperature, topk, topp,
              u.to(dev).contiguous(), 3, rows, 0, None, None)
    got = sampled.cpu().long()
    diff = (got != want).nonzero().flatten().tolist()
    probs = torch.softmax(O.top_k_mask_logits(lg / temperature, topk), dim=-1)
    for r in diff:                                            # only a rounding-boundary case may differ: both tokens in the top-k set
        assert probs[r][got[r]] > 0 and probs[r][want[r]] > 0, (r, int(got[r]), int(want[r]))
    assert len(diff) <= 1, (diff, got[diff], want[diff])


def test_sample_rows_greedy_with_penalty_and_seed(dev):
    """temperature < 0.05 with a penalty: arg-max of the penalised row; same (seed, n) -> same draw."""
    from umbrella_amd import _lib
    g = torch.Generator().manual_seed(3)
    V, rows, n = 4096, 5, 30
    logits = torch.randn(rows, V, generator=g)
    hist = torch.cat([logits.argmax(-1), torch.randint(0, V, (n + 1 - rows,), generator=g)])
    tokens = torch.zeros(n + 8, dtype=torch.int32); tokens[:n + 1] = hist.int()
    tokens, nd = tokens.to(dev), torch.tensor([n], dtype=torch.int32, device=dev)
    seed = torch.tensor([7], dtype=torch.int64, device=dev)
    out = torch.zeros(rows, dtype=torch.int32, device=dev)
    _lib.call("umb_sample_rows", out, logits.to(dev).clone(), rows, V, tokens, nd, 1.5, 0.0, 32, 0.9, seed, 0, None, None)
    exp = O.repetition_penalty(hist[None].expand(rows, -1), logits, 1.5).argmax(-1)
    assert out.cpu().tolist() == exp.tolist()
    a = torch.zeros(rows, dtype=torch.int32, device=dev); b = torch.zeros_like(a)
    for dst in (a, b):
        _lib.call("umb_sample_rows", dst, logits.to(dev).clone(), rows, V, tokens, nd, 1.0, 0.8, 32, 0.95, seed, 0, None, None)
    assert torch.equal(a, b)
    with pytest.raises(RuntimeError):
        _lib.call("umb_sample_rows", a, logits.to(dev).clone(), rows, V, tokens, nd, 1.0, 0.8, 2000, 0.95, seed, 0, None, None)


def test_accept_scan_matches_oracle(dev):
    from oracle import sequoia
    from umbrella_amd import _lib
    rs = np.random.RandomState(7)
    for name, gm in (("3x4", sequoia.generate(3, 4)), ("5x6", sequoia.generate(5, 6, [0.5, 0.2, 0.12, 0.08, 0.05, 0.03]))):
        T = gm["size"]
        mask = torch.tensor(gm["mask"]) == 1
        want = mask.sum(-1)
        parents = torch.zeros(T, dtype=torch.int32)
        for v, s in enumerate(gm["Successors"]):
            parents[s] = v
        for trial in range(60):
            n = int(rs.randint(0, 50))
            spec = torch.from_numpy(rs.randint(10, 14, size=T)).int()
            sampled = torch.from_numpy(rs.randint(10, 14, size=T)).int()
            eos = [13] if trial % 3 == 0 else [999]
            path, bonus = O.accept_scan(sampled, spec, parents, mask, want)
            a = len(path)
            toks = spec[path].tolist() + [bonus]
            e = O.first_eos(toks, eos)
            keep = e if e >= 0 else a
            tokens = torch.zeros(128, dtype=torch.int32)
            tokens[n:n + T] = spec
            td, nd = tokens.to(dev), torch.tensor([n], dtype=torch.int32, device=dev)
            res = torch.zeros(8, dtype=torch.int32, device=dev)
            pth = torch.zeros(16, dtype=torch.int32, device=dev)
            _lib.call("umb_accept_scan", sampled.to(dev), parents.to(dev), td, nd, T, torch.tensor(eos, dtype=torch.int32, device=dev),
                      1, res, pth)
            r = res.cpu().tolist()
            assert r[:5] == [keep, bonus, int(e >= 0), n + keep, a], (name, trial, r, keep, bonus, e, a)
            assert pth.cpu()[:keep].tolist() == path[:keep].tolist()
            assert td.cpu()[n:n + a + 1].tolist() == toks
            assert int(nd.cpu()) == n + keep


@pytest.mark.parametrize("D", [64, 128])
def test_kv_compaction(dev, D):
    from umbrella_amd.attn.cache import TreeKVCache
    L, Hkv, Lmax = 3, 2, 64
    c = TreeKVCache(L, Hkv, D, Lmax, dev, torch.bfloat16)
    g = torch.Generator().manual_seed(9)
    # semantic contents (key-major K, feature-major V^T); the cache stores them in fragment order
    k0 = torch.randn(c.k.shape, generator=g).to(torch.bfloat16)
    v0 = torch.randn(c.vt.shape, generator=g).to(torch.bfloat16)
    v0[..., Lmax:] = 0
    c.k.copy_(k_to_frag(k0)); c.vt.copy_(vt_to_frag(v0))
    n_old, path = 20, [0, 2, 5, 11]
    res = torch.tensor([len(path), 0, 0, n_old + len(path), len(path), 0, 0, 0], dtype=torch.int32, device=dev)
    c.compact(res, torch.tensor(path + [0] * 4, dtype=torch.int32, device=dev), 8)
    ke, ve = k0.clone(), v0.clone()
    idx = torch.tensor(path) + n_old
    ke[:, :, n_old:n_old + 4] = k0[:, :, idx]
    ve[:, :, :, n_old:n_old + 4] = v0[:, :, :, idx]
    assert torch.equal(k_from_frag(c.k.cpu()), ke) and torch.equal(vt_from_frag(c.vt.cpu()), ve)
    # reference-signature gather gives the same result
    c2 = TreeKVCache(L, Hkv, D, Lmax, dev, torch.bfloat16)
    c2.k.copy_(k_to_frag(k0)); c2.vt.copy_(vt_to_frag(v0))
    c2.gather_kv_incremental(idx.to(dev), n_old)
    assert torch.equal(c2.k, c.k) and torch.equal(c2.vt, c.vt) and c2.kv_offset == n_old + 4
    assert torch.equal(c2.k_rows(idx + 0)[:, :, 0].cpu(), k0[:, :, n_old]) and torch.equal(c2.v_rows([n_old + 1]).cpu()[:, :, 0], v0[:, :, :, n_old + 2])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("awq", [False, True])
@pytest.mark.parametrize("T", [1, 13, 40, 130])
def test_gemm_fused_silu_epilogue(dev, dtype, awq, T):
    """[gate; up] stored as interleaved rows: SiLU(gate)*up is the GEMM epilogue (umbrella/models/llama.py:107-110)."""
    from umbrella_amd.models.awq_format import pack_rows
    from umbrella_amd.models.llama import PackedLinear
    rs = np.random.RandomState(T + int(awq))
    I, K = 768, 512
    x = torch.from_numpy(rs.randn(T, K).astype(np.float32)).to(dtype)
    if awq:
        q = rs.randint(0, 16, size=(K, 2 * I)).astype(np.uint8)
        z = rs.randint(0, 16, size=(K // 128, 2 * I)).astype(np.uint8)
        s = (rs.rand(K // 128, 2 * I) * 0.02 + 0.002).astype(np.float16)
        W = torch.from_numpy((q.astype(np.float32) - np.repeat(z, 128, 0)) * np.repeat(s.astype(np.float32), 128, 0)).t()
        lin = PackedLinear.from_awq(torch.from_numpy(pack_rows(q)).to(dev), torch.from_numpy(pack_rows(z)).to(dev),
                                    torch.from_numpy(s).to(dev), interleave=True)
    else:
        W = (torch.from_numpy(rs.randn(2 * I, K).astype(np.float32)) * 0.05).to(dtype).float()
        lin = PackedLinear.from_dense(W.to(dtype).to(dev), interleave=True)
    assert lin.S == 1
    act = lin.apply_silu(x.to(dev)).cpu()
    full = x.float() @ W.t()
    gate, up = full[:, :I].to(dtype), full[:, I:].to(dtype)
    ref = torch.nn.functional.silu(gate) * up
    assert act.shape == (T, I)
    assert (act.float() - ref.float()).abs().max() <= 8 * torch.finfo(dtype).eps * ref.float().abs().max()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T", [1, 13, 40, 70])
def test_gemm_fused_residual_epilogue(dev, dtype, T):
    """epi 4: split-K partials merged by the last-arriving block, h += gemm, hw = h * w, ssq partial sums."""
    from umbrella_amd import _lib
    from umbrella_amd.models.llama import PackedLinear
    g = torch.Generator().manual_seed(T)
    N, K = 512, 1024
    W = (torch.randn(N, K, generator=g) * 0.05).to(dtype)
    x = torch.randn(T, K, generator=g).to(dtype)
    h0 = torch.randn(T, N, generator=g).to(dtype)
    nw = (1 + 0.1 * torch.randn(N, generator=g)).to(dtype)
    lin = PackedLinear.from_dense(W.to(dev))
    assert lin.S > 1
    stride = 12
    counters = torch.zeros(64, dtype=torch.int32, device=dev)
    part = torch.empty(lin.S * T * N, dtype=torch.float32, device=dev)
    o = (x.float() @ W.float().t()).to(dtype)
    hn = (o.float() + h0.float()).to(dtype)
    hw_ref = (hn.float() * nw.float()).to(dtype)
    ssq_ref = hn.float().pow(2).view(T, N // 64, 64).sum(-1)
    for rep in range(2):                                           # second run: counters must have reset
        h, hw = h0.clone().to(dev), torch.zeros(T, N, dtype=dtype, device=dev)
        ssq = torch.zeros(T, stride, dtype=torch.float32, device=dev)
        fx = _lib.UmbGemmFused()
        fx.counters, fx.h, fx.hw, fx.norm_w = counters.data_ptr(), h.data_ptr(), hw.data_ptr(), nw.to(dev).data_ptr()
        nwd = nw.to(dev); fx.norm_w = nwd.data_ptr()
        fx.ssq_out, fx.ssq_out_stride = ssq.data_ptr(), stride
        _lib.call("umb_gemm_fused", part, x.to(dev), K, lin.w, lin.meta, T, N, K, 0, lin.S, lin.R, 4, fx,
                  _lib.dtype_code(dtype))
        eps = torch.finfo(dtype).eps
        assert (h.cpu().float() - hn.float()).abs().max() <= 2 * eps * hn.float().abs().max()
        assert (hw.cpu().float() - hw_ref.float()).abs().max() <= 4 * eps * hw_ref.float().abs().max()
        got = ssq.cpu()[:, :N // 64]
        assert (got - ssq_ref).abs().max() <= 0.05 * ssq_ref.abs().max()
        assert ssq.cpu()[:, N // 64:].abs().max() == 0 and int(counters.abs().sum()) == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T", [1, 7, 40])
@pytest.mark.parametrize("awq", [False, True])
def test_gemm_fused_qkv_epilogue(dev, dtype, T, awq):
    """epi 3: 1/rms from ssq, rotate-half RoPE on lane-local partner rows, q out, K / V^T cache append."""
    from umbrella_amd import _lib
    from umbrella_amd.models.awq_format import pack_rows
    from umbrella_amd.models.llama import PackedLinear
    rs = np.random.RandomState(T)
    g = torch.Generator().manual_seed(T)
    Hq, Hkv, D, Lmax, K = 4, 2, 64, 128, 1024
    N = (Hq + 2 * Hkv) * D
    x = torch.randn(T, K, generator=g).to(dtype)
    if awq:
        q = rs.randint(0, 16, size=(K, N)).astype(np.uint8); z = rs.randint(0, 16, size=(K // 128, N)).astype(np.uint8)
        sc = (rs.rand(K // 128, N) * 0.02 + 0.002).astype(np.float16)
        Wf = torch.from_numpy((q.astype(np.float32) - np.repeat(z, 128, 0)) * np.repeat(sc.astype(np.float32), 128, 0)).t().contiguous()
        lin = PackedLinear.from_awq(torch.from_numpy(pack_rows(q)).to(dev), torch.from_numpy(pack_rows(z)).to(dev),
                                    torch.from_numpy(sc).to(dev), rope=(D, Hq + Hkv))
    else:
        Wf = (torch.randn(N, K, generator=g) * 0.05).to(dtype).float()
        lin = PackedLinear.from_dense(Wf.to(dtype).to(dev), rope=(D, Hq + Hkv))
    stride = 16
    ssq = torch.zeros(T, stride); ssq[:, :K // 64] = torch.rand(T, K // 64, generator=g) * 100 + 10
    inv = torch.rsqrt(ssq.sum(-1) / K + 1e-5)
    pos = torch.from_numpy(rs.randint(0, Lmax, size=T)).int()
    slot = torch.from_numpy(rs.permutation(Lmax)[:T].copy()).int()
    ifr = 1.0 / (500000.0 ** (torch.arange(0, D, 2).float() / D))
    cos, sin = O.rope_cache(ifr, 1.0, Lmax, dtype)
    y = ((x.float() @ Wf.t()) * inv[:, None]).to(dtype)
    qr, kr, vr = y[:, :Hq * D].view(T, Hq, D), y[:, Hq * D:(Hq + Hkv) * D].view(T, Hkv, D), y[:, (Hq + Hkv) * D:].view(T, Hkv, D)
    qe, ke = O.apply_rope(qr, kr, cos, sin, pos.long())
    counters = torch.zeros(64, dtype=torch.int32, device=dev)
    part = torch.empty(lin.S * T * N + 64, dtype=torch.float32, device=dev)
    qo = torch.zeros(T, Hq, D, dtype=dtype, device=dev)
    kc = torch.zeros(Hkv, Lmax, D, dtype=dtype, device=dev); vt = torch.zeros(Hkv, D, Lmax + VT_PAD, dtype=dtype, device=dev)
    keep = [x.to(dev), ssq.to(dev), pos.to(dev), slot.to(dev), cos.to(dev), sin.to(dev)]
    fx = _lib.UmbGemmFused()
    fx.ssq_in, fx.ssq_groups, fx.ssq_dim, fx.eps = keep[1].data_ptr(), stride, float(K), 1e-5
    fx.counters, fx.pos, fx.slot, fx.cosT, fx.sinT = counters.data_ptr(), keep[2].data_ptr(), keep[3].data_ptr(), keep[4].data_ptr(), keep[5].data_ptr()
    fx.q_out, fx.k_cache, fx.vt_cache, fx.Hq, fx.Hkv, fx.D, fx.Lmax = qo.data_ptr(), kc.data_ptr(), vt.data_ptr(), Hq, Hkv, D, Lmax
    _lib.call("umb_gemm_fused", part, keep[0], K, lin.w, lin.meta, T, N, K, int(awq), lin.S, lin.R, 3, fx, _lib.dtype_code(dtype))
    eps = torch.finfo(dtype).eps
    tol = 6 * eps
    assert (qo.cpu().float() - qe.float()).abs().max() <= tol * qe.float().abs().max()
    kg = k_from_frag(kc.cpu())[:, slot.long()].permute(1, 0, 2)
    assert (kg.float() - ke.float()).abs().max() <= tol * ke.float().abs().max()
    vg = vt_from_frag(vt.cpu())[:, :, slot.long()].permute(2, 0, 1)
    assert (vg.float() - vr.float()).abs().max() <= tol * vr.float().abs().max()
    assert int(counters.abs().sum()) == 0


@pytest.mark.parametrize("N,K", [(57344, 8192), (8192, 28672), (10240, 8192)])
def test_gemm_awq_full_size_properties(dev, N, K):
    """Llama-70B AWQ linear shapes at full size (BASELINE configs 3-5): size-independent properties --
    batch invariance (bitwise), linearity, zero in -> zero out -- plus the CPU oracle on column slices."""
    from umbrella_amd.models.llama import PackedLinear
    from umbrella_amd.models.synthetic import synth_awq_tensors
    gen = torch.Generator(device=dev).manual_seed(N + K)
    qw, qz, sc = synth_awq_tensors(N, K, 128, dev, gen)
    lin = PackedLinear.from_awq(qw, qz, sc)
    x = (torch.randn(13, K, device=dev, generator=gen) * 0.5).half()
    y = lin.apply(x)
    assert torch.equal(lin.apply(x[:1].contiguous()), y[:1]) and torch.equal(lin.apply(x[:5].contiguous()), y[:5])
    assert float(lin.apply(torch.zeros_like(x)).abs().max()) == 0.0
    x2 = (torch.randn(13, K, device=dev, generator=gen) * 0.5).half()
    ysum = lin.apply((x.float() + x2.float()).half())
    lin_err = (ysum - (y + lin.apply(x2))).abs().max() / y.abs().max()
    assert float(lin_err) < 5e-3, float(lin_err)                      # fp16 rounding of x1 + x2 only
    for c0 in (0, N // 2 + 64, N - 64):
        ref = O.awq_linear(x.cpu().float(), qw[:, c0 // 8:(c0 + 64) // 8].cpu(), qz[:, c0 // 8:(c0 + 64) // 8].cpu(),
                           sc[:, c0:c0 + 64].cpu(), 128)
        got = y[:, c0:c0 + 64].cpu()
        assert _rel(got, ref) < 2e-3, (c0, _rel(got, ref))


@pytest.mark.parametrize("awq", [True, False])
@pytest.mark.parametrize("T", [128, 256, 257, 300, 385, 769])
def test_verify_gemm_wide_full_width(dev, awq, T):
    """Wide-token GEMMs at 70B widths (N = 8192 / 28672-row groups, so the 256-row register-resident kernels are
    selected: 128-token blocks, 144-token blocks for T = w d + 1, skinny tail otherwise): every row must equal what
    the T <= 16 kernel gives for that token alone up to fp32 summation order, and the fp32 reference on slices."""
    from umbrella_amd.models.llama import PackedLinear
    from umbrella_amd.models.synthetic import synth_awq_tensors
    gen = torch.Generator(device=dev).manual_seed(T + awq)
    N, K = 8192, 2048
    if awq:
        qw, qz, sc = synth_awq_tensors(N, K, 128, dev, gen)
        lin = PackedLinear.from_awq(qw, qz, sc)
        dtype = torch.float16
    else:
        w = (torch.randn(N, K, device=dev, generator=gen) * 0.02).bfloat16()
        lin = PackedLinear.from_dense(w)
        dtype = torch.bfloat16
    x = (torch.randn(T, K, device=dev, generator=gen) * 0.5).to(dtype)
    y = lin.apply(x)                                                  # wide path
    rows = [0, 1, 15, 127, 128, T - 2, T - 1]
    rows = sorted(set(r for r in rows if 0 <= r < T))
    y1 = torch.cat([lin.apply(x[r:r + 1].contiguous()) for r in rows])
    scale = float(y1.abs().max())
    # dense: fp32 summation order only.  int4: the wide kernels dequantise exactly (W = fp16((q - z) s)) while launches of
    # <= 64 rows use the folded form s (sum q x - z sum x) -- the difference is the fp16 rounding of the weights, 2^-11
    # relative per weight, < 2e-3 of the row scale after the sum (same bound as against the oracle below)
    tol = 2e-3 if awq else 2e-5
    assert float((y[rows] - y1).abs().max()) <= tol * scale + 1e-6, float((y[rows] - y1).abs().max())
    if awq:
        from test_hip_engine import _awq_dequant_torch
        wd = _awq_dequant_torch(qw[:, :16], qz[:, :16], sc[:, :128]).float()       # first 128 output columns
    else:
        wd = w[:128].float().t()
    ref = x.float() @ wd
    assert float((y[:, :128] - ref).abs().max()) <= 2e-3 * float(ref.abs().max())


@pytest.mark.parametrize("T", [65, 80, 96, 111, 128, 143, 160, 176, 191, 208, 224, 239, 256, 257, 272, 288, 289, 300, 400, 577, 1000, 1024])
def test_vgemm_w_every_instantiation_on_small_matrices(dev, T):
    """csrc/vgemm.hip on small matrices (UMB_VGW_MIN_WGS=1 lets the 256-row kernel take them): every token-tile instantiation
    (TT = 5 ... 18 through T = 65 ... 288, multi-chunk splits beyond), K slabs of ONE 128-k block, ragged splits (K = 384 at S = 2:
    2 + 1 blocks), N = 256 ... 768, split-K partials and the fused SiLU epilogue with per-token 1/rms -- EVERY output against an fp32
    matmul over the exactly dequantised fp16 weights (fp32 summation order is the only freedom)."""
    from test_hip_engine import _awq_dequant_torch
    from umbrella_amd import _lib
    from umbrella_amd.models.llama import PackedLinear
    from umbrella_amd.models.synthetic import synth_awq_tensors
    lib = _lib.load()
    dt = _lib.dtype_code(torch.float16)
    os.environ["UMB_VGW_MIN_WGS"] = "1"
    try:
        for N, K, S, il in ((256, 128, 1, False), (512, 384, 2, False), (768, 1024, 3, False), (512, 512, 1, True)):
            gen = torch.Generator(device=dev).manual_seed(T * 7 + N + K)
            qw, qz, sc = synth_awq_tensors(N, K, 128, dev, gen)
            lin = PackedLinear.from_awq(qw, qz, sc, interleave=il)
            assert lib.umb_vgemm_w_ok(T, N, K, S, 2 if il else 0) == 1
            x = (torch.randn(T + 3, K + 64, device=dev, generator=gen) * 0.5).half()[:T, :K]      # a strided view: ldx > K
            wd = _awq_dequant_torch(qw, qz, sc).float()
            ref = x.float() @ wd
            if il:
                G, stride = 8, 12
                ssq = torch.rand(T, stride, device=dev, generator=gen) * 30 + 10
                ssq[:, G:] = 1e9                                             # beyond the valid groups: must not be read
                fx = _lib.UmbGemmFused()
                fx.ssq_in, fx.ssq_groups, fx.pad0, fx.ssq_dim, fx.eps = ssq.data_ptr(), G, stride, float(K), 1e-5
                act = torch.full((T, N // 2), float("nan"), dtype=torch.float16, device=dev)
                _lib.call("umb_gemm_fused", act, x, x.stride(0), lin.w, lin.meta, T, N, K, 1, 1, lin.Rtb, 2, fx, dt)
                inv = torch.rsqrt(ssq[:, :G].sum(1) / K + 1e-5)[:, None]
                I = N // 2
                gate, up = (ref[:, :I] * inv).half(), (ref[:, I:] * inv).half()
                want = torch.nn.functional.silu(gate.float()).half().float() * up.float()
                assert torch.isfinite(act).all()
                assert float((act.float() - want).abs().max()) <= 8 * torch.finfo(torch.float16).eps * float(want.abs().max()) + 1e-6
            else:
                part = torch.full((S, T, N), float("nan"), dtype=torch.float32, device=dev)
                _lib.call("umb_gemm", part, x, x.stride(0), lin.w, lin.meta, T, N, K, 1, S, lin.Rtb, 0, dt)
                assert torch.isfinite(part).all()
                y = part.sum(0)
                assert float((y - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-6, (N, K, S)
    finally:
        os.environ.pop("UMB_VGW_MIN_WGS", None)


@pytest.mark.parametrize("T", [256, 257, 385, 769])
@pytest.mark.parametrize("shape", ["qkv", "gu", "down"])
def test_verify_gemm_wide_every_output_70b_shapes(dev, shape, T):
    """The int4 verify GEMM at the real 70B layer shapes (two-phase kernels: 128-token items, 144-token items for
    T = w d + 1, every split count): EVERY output -- all T rows, all N columns, every K split summed -- against an fp32
    matmul over the exactly dequantised fp16 weights W = fp16((q - z) s) (awq_ext.dequantize_weights_cuda's values,
    awq_utils.py:67-77); the only difference allowed is fp32 summation order.  gate/up runs its fused SiLU(gate) * up
    epilogue (llama.py:107-110) and is compared after the same roundings."""
    from test_hip_engine import _awq_dequant_torch
    from umbrella_amd.models.llama import PackedLinear
    from umbrella_amd.models.synthetic import synth_awq_tensors
    N, K, il = {"qkv": (10240, 8192, False), "gu": (2 * 28672, 8192, True), "down": (8192, 28672, False)}[shape]
    if shape == "gu" and T == 385:
        pytest.skip("covered by 257 / 769 (same kernel instantiation)")
    gen = torch.Generator(device=dev).manual_seed(T + N)
    qw, qz, sc = synth_awq_tensors(N, K, 128, dev, gen)
    lin = PackedLinear.from_awq(qw, qz, sc, interleave=il)
    x = (torch.randn(T, K, device=dev, generator=gen) * 0.5).half()
    wd = _awq_dequant_torch(qw, qz, sc).float()                       # [K, N], exact fp16 values
    ref = x.float() @ wd
    del wd
    if il:
        act = lin.apply_silu(x)
        I = N // 2
        gate, up = ref[:, :I].half(), ref[:, I:].half()
        want = (torch.nn.functional.silu(gate.float()).half().float() * up.float())
        err = (act.float() - want).abs().max()
        assert float(err) <= 8 * torch.finfo(torch.float16).eps * float(want.abs().max()), float(err)
        # and no output may be off by more than a rounding step of its own magnitude (a misplaced row / token would be)
        bad = ((act.float() - want).abs() > 4e-3 * want.abs() + 1e-4 * float(want.abs().max())).sum()
        assert int(bad) == 0, int(bad)
    else:
        S = _lib_wide_split(T, N, lin.S)
        part = torch.empty(S, T, N, dtype=torch.float32, device=dev)
        from umbrella_amd import _lib
        _lib.call("umb_gemm", part, x, x.stride(0), lin.w, lin.meta, T, N, K, 1, S, lin.Rtb, 0, _lib.dtype_code(x.dtype))
        y = part.sum(0)
        scale = float(ref.abs().max())
        err = float((y - ref).abs().max())
        assert err <= 2e-4 * scale, (err, scale)


def _lib_wide_split(T, N, S):
    from umbrella_amd import _lib
    return _lib.load().umb_gemm_wide_split(T, N, S)


# ------------------------------------------------------------------ stand-alone 16-bit RoPE / KV append / slab copy (C-ABI completeness)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("layout", [0, 1])
def test_rope_inplace_matches_oracle(dev, dtype, layout):
    """umb_rope_inplace == apply_rotary_pos_emb (model_utils.py:17-52) on 16-bit q / k at tree positions: bit-exact
    (every product and the sum rounded to the model dtype, as the reference's eager torch does)."""
    from umbrella_amd import _lib
    from umbrella_amd.models.config import LlamaCfg, rope_tables
    T, Hq, Hkv, D, Lmax = 13, 8, 2, 128, 64
    g = torch.Generator().manual_seed(3 + layout)
    q = torch.randn(T, Hq, D, generator=g).to(dtype)
    k = torch.randn(T, Hkv, D, generator=g).to(dtype)
    pos = torch.tensor([9, 10, 10, 11, 11, 11, 12, 12, 12, 12, 13, 40, 63], dtype=torch.int32)
    cos, sin = rope_tables(LlamaCfg(head_dim=D), Lmax, dtype)
    qe, ke = O.apply_rope(q, k, cos, sin, pos.long())
    qd = (q if layout == 0 else q.permute(1, 0, 2)).contiguous().to(dev)
    kd = (k if layout == 0 else k.permute(1, 0, 2)).contiguous().to(dev)
    _lib.call("umb_rope_inplace", qd, kd, cos.to(dev).contiguous(), sin.to(dev).contiguous(), pos.to(dev), T, Hq, Hkv, D,
              layout, _lib.dtype_code(dtype))
    got_q = qd.cpu() if layout == 0 else qd.cpu().permute(1, 0, 2)
    got_k = kd.cpu() if layout == 0 else kd.cpu().permute(1, 0, 2)
    assert torch.equal(got_q, qe) and torch.equal(got_k, ke)
    # k only (q == NULL, Hq == 0)
    kd2 = (k if layout == 0 else k.permute(1, 0, 2)).contiguous().to(dev)
    _lib.call("umb_rope_inplace", None, kd2, cos.to(dev).contiguous(), sin.to(dev).contiguous(), pos.to(dev), T, 0, Hkv, D,
              layout, _lib.dtype_code(dtype))
    assert torch.equal(kd2, kd)
    with pytest.raises(_lib.UmbError):
        _lib.call("umb_rope_inplace", None, kd2, cos.to(dev), sin.to(dev), pos.to(dev), T, Hq, Hkv, D, layout, _lib.dtype_code(dtype))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_kv_append_then_attention(dev, dtype):
    """umb_kv_append places k / v rows at their slots of the K and V^T caches (cache.py:53-65,155-156); attention over
    the appended cache equals the oracle's masked attention on the same keys."""
    from umbrella_amd import _lib
    T, Hq, Hkv, D, Lmax = 7, 4, 2, 64, 64
    g = torch.Generator().manual_seed(5)
    kc = torch.zeros(Hkv, Lmax, D, dtype=dtype, device=dev)
    vt = torch.zeros(Hkv, D, Lmax + VT_PAD, dtype=dtype, device=dev)
    k = torch.randn(T, Hkv, D, generator=g).to(dtype)
    v = torch.randn(T, Hkv, D, generator=g).to(dtype)
    slot = torch.tensor([9, 10, 11, 12, 13, 14, 70], dtype=torch.int32)            # the last slot is outside the cache: dropped
    _lib.call("umb_kv_append", kc, vt, k.to(dev), v.to(dev), slot.to(dev), T, Hkv, D, Lmax, _lib.dtype_code(dtype))
    kref = torch.zeros(Hkv, Lmax, D, dtype=dtype)
    vref = torch.zeros(Hkv, Lmax, D, dtype=dtype)
    for t in range(T - 1):
        kref[:, int(slot[t])] = k[t]
        vref[:, int(slot[t])] = v[t]
    assert torch.equal(k_from_frag(kc.cpu()), kref)
    assert torch.equal(vt_from_frag(vt.cpu())[:, :, :Lmax], vref.permute(0, 2, 1))
    assert float(vt.view(Hkv, -1)[:, Lmax * D:].abs().max()) == 0.0          # nothing lands past the Lmax D fragment region


def test_h2d_layer_event_ordered(dev):
    """umb_h2d_layer: one pinned-host -> device slab copy on a side stream, ordered by (ev_free, ev_copied)."""
    import ctypes as C
    from umbrella_amd import _lib
    lib = _lib.load()
    n = 8 << 20
    host = torch.randint(0, 255, (n,), dtype=torch.uint8).pin_memory()
    dst = torch.zeros(n, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream(device=dev)
    ev_free, ev_copied = torch.cuda.Event(), torch.cuda.Event()
    dst.add_(1)                                                    # a kernel that still uses `dst` on the compute stream
    ev_free.record()
    ev_copied.record()
    rc = lib.umb_h2d_layer(C.c_void_p(dst.data_ptr()), C.c_void_p(host.data_ptr()), C.c_size_t(n),
                           C.c_void_p(side.cuda_stream), C.c_void_p(ev_free.cuda_event), C.c_void_p(ev_copied.cuda_event))
    assert rc == 0
    torch.cuda.current_stream().wait_event(ev_copied)
    out = dst + 0
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), host)
    assert lib.umb_h2d_layer(None, C.c_void_p(host.data_ptr()), C.c_size_t(n), C.c_void_p(side.cuda_stream), None, None) == -22
