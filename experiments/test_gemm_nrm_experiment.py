"""Norm-on-staging GEMM (GPU): umb_reduce_residual_wide + umb_gemm_nrm against the form they replace in the layer
schedule, umb_reduce_residual_norm + umb_gemm (reference: llama.py:104,112 residual adds, model_utils.py:54-64
rmsnorm, the next F.linear / awq gemm), and against fp32 arithmetic."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from umbrella_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _awq(rs, N, K, dev, interleave):
    from umbrella_amd.models.awq_format import pack_rows
    from umbrella_amd.models.llama import PackedLinear
    q = rs.randint(0, 16, size=(K, N)).astype(np.uint8)
    z = rs.randint(0, 16, size=(K // 128, N)).astype(np.uint8)
    s = (rs.rand(K // 128, N) * 0.02 + 0.002).astype(np.float16)
    return PackedLinear.from_awq(torch.from_numpy(pack_rows(q)).to(dev), torch.from_numpy(pack_rows(z)).to(dev),
                                 torch.from_numpy(s).to(dev), interleave=interleave)


def _dense(rs, N, K, dev, dtype, interleave):
    from umbrella_amd.models.llama import PackedLinear
    W = (torch.from_numpy(rs.randn(N, K).astype(np.float32)) * 0.05).to(dtype)
    return PackedLinear.from_dense(W.to(dev), interleave=interleave, force_s1=interleave)


def _outbuf(lin, T, epi, dtype, dev):
    S = 1 if epi == 2 else lin.S
    return (torch.empty(T, lin.N // 2, dtype=dtype, device=dev) if epi == 2
            else torch.empty(S, T, lin.N, dtype=torch.float32, device=dev)), S


def _old(lin, part, Sp, h0, nw, eps, T, K, epi, dtype, dev):
    from umbrella_amd import _lib
    h, xn = h0.clone(), torch.empty_like(h0)
    _lib.call("umb_reduce_residual_norm", part, Sp, T, K, h, h, xn, nw, eps, _lib.dtype_code(dtype))
    out, S = _outbuf(lin, T, epi, dtype, dev)
    _lib.call("umb_gemm", out, xn, K, lin.w, lin.meta, T, lin.N, K, lin.awq, S, lin.Rtb, epi, _lib.dtype_code(dtype))
    return h, out


def _new(lin, part, Sp, h0, nw, eps, T, K, epi, dtype, dev):
    from umbrella_amd import _lib
    h = h0.clone()
    groups = K // 1024
    ssq = torch.zeros(T, groups, dtype=torch.float32, device=dev)
    _lib.call("umb_reduce_residual_wide", part, Sp, T, K, h, ssq, _lib.dtype_code(dtype))
    out, S = _outbuf(lin, T, epi, dtype, dev)
    _lib.call("umb_gemm_nrm", out, h, nw, ssq, groups, eps, lin.w, lin.meta, T, lin.N, K, lin.awq, S, lin.Rtb, epi,
              _lib.dtype_code(dtype))
    return h, ssq, out


CASES = [(2048, 1024, 0, 0), (2048, 1024, 1, 0), (3072, 1024, 0, 2), (4096, 2048, 1, 2), (10240, 2048, 1, 0), (512, 4096, 0, 0)]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("N,K,awq,epi", CASES)
@pytest.mark.parametrize("T", [1, 13, 16, 31, 40, 64])
def test_gemm_nrm_matches_two_kernel_form(dev, dtype, N, K, awq, epi, T):
    rs = np.random.RandomState(N + K + T + epi)
    lin = _awq(rs, N, K, dev, epi == 2) if awq else _dense(rs, N, K, dev, dtype, epi == 2)
    Sp = 1 + (N + T) % 9                          # 1 .. 9 producer splits (the unrolled and the tail loop of the reduce)
    part = torch.from_numpy(rs.randn(Sp, T, K).astype(np.float32) * 0.3).to(dev)
    h0 = torch.from_numpy(rs.randn(T, K).astype(np.float32)).to(dtype).to(dev)
    nw = (1 + 0.1 * torch.from_numpy(rs.randn(K).astype(np.float32))).to(dtype).to(dev)
    eps = 1e-5
    h_ref, out_ref = _old(lin, part, Sp, h0, nw, eps, T, K, epi, dtype, dev)
    h, ssq, out = _new(lin, part, Sp, h0, nw, eps, T, K, epi, dtype, dev)
    assert torch.equal(h, h_ref)                   # the residual stream: bit-exact
    ss = h.float().pow(2).view(T, K // 1024, 1024).sum(-1)
    assert float((ssq - ss).abs().max()) <= 1e-5 * float(ss.abs().max())
    # x differs from the two-kernel form only through the summation order of the row's sum of squares (1 ulp of 1/rms)
    a, b = out.float(), out_ref.float()
    tol = (8 * torch.finfo(dtype).eps if epi == 2 else 2e-3) * float(b.abs().max())
    assert float((a - b).abs().max()) <= tol


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_gemm_nrm_batch_invariance(dev, dtype):
    """a token's h, sums of squares and outputs do not depend on how many rows share the launch"""
    rs = np.random.RandomState(5)
    N, K = 2048, 2048
    lin = _dense(rs, N, K, dev, dtype, False)
    Sp, T = 3, 40
    part = torch.from_numpy(rs.randn(Sp, T, K).astype(np.float32) * 0.3).to(dev)
    h0 = torch.from_numpy(rs.randn(T, K).astype(np.float32)).to(dtype).to(dev)
    nw = (1 + 0.1 * torch.from_numpy(rs.randn(K).astype(np.float32))).to(dtype).to(dev)
    h_full, ssq_full, out_full = _new(lin, part, Sp, h0, nw, 1e-5, T, K, 0, dtype, dev)
    for t in (1, 5, 13, 17, 33):
        h, ssq, out = _new(lin, part[:, :t].contiguous(), Sp, h0[:t].contiguous(), nw, 1e-5, t, K, 0, dtype, dev)
        assert torch.equal(h, h_full[:t]) and torch.equal(ssq, ssq_full[:t]) and torch.equal(out, out_full[:, :t])


def test_gemm_nrm_70b_gate_up_full_size(dev):
    """the headline launch: 70B-AWQ gate/up (N = 57344, K = 8192, one 8-wave block per CU) on the norm-on-staging path"""
    rs = np.random.RandomState(11)
    N, K, T, dtype = 57344, 8192, 13, torch.float16
    lin = _awq(rs, N, K, dev, True)
    assert lin.tb == (14 | 0x80)
    Sp = 4
    part = torch.from_numpy(rs.randn(Sp, T, K).astype(np.float32) * 0.3).to(dev)
    h0 = torch.from_numpy(rs.randn(T, K).astype(np.float32)).to(dtype).to(dev)
    nw = (1 + 0.1 * torch.from_numpy(rs.randn(K).astype(np.float32))).to(dtype).to(dev)
    h_ref, out_ref = _old(lin, part, Sp, h0, nw, 1e-5, T, K, 2, dtype, dev)
    h, _, out = _new(lin, part, Sp, h0, nw, 1e-5, T, K, 2, dtype, dev)
    assert torch.equal(h, h_ref)
    assert float((out.float() - out_ref.float()).abs().max()) <= 8 * torch.finfo(dtype).eps * float(out_ref.float().abs().max())
    assert float((out == out_ref).float().mean()) > 0.98


def test_gemm_nrm_rejects_what_it_does_not_cover(dev):
    from umbrella_amd import _lib
    lib = _lib.load()
    assert lib.umb_gemm_nrm_ok(13, 8192, 1, 8) == 1
    assert lib.umb_gemm_nrm_ok(65, 8192, 1, 8) == 0 and lib.umb_gemm_nrm_ok(13, 8192, 1, 0) == 0
    x = torch.zeros(4, 1000, dtype=torch.float16, device=dev)
    ssq = torch.zeros(4, 4, dtype=torch.float32, device=dev)
    with pytest.raises(_lib.UmbError):
        _lib.call("umb_reduce_residual_wide", ssq, 1, 4, 1000, x, ssq, _lib.dtype_code(torch.float16))   # N % 1024
