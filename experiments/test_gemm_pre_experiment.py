"""umb_gemm_pre (GPU): the consumer GEMM that finishes its producer -- split reduce + residual add + RMSNorm ahead of an
in-kernel grid barrier -- against the two-kernel form it replaces (umb_reduce_residual_norm + umb_gemm) and against the
oracle's fp32 arithmetic (reference: llama.py:104,112 residual adds, model_utils.py:54-64 rmsnorm, the next F.linear)."""
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


def _two_kernel(lin, part, Sp, h0, nw, eps, T, K, epi, dtype, dev):
    from umbrella_amd import _lib
    h = h0.clone()
    xn = torch.empty_like(h)
    _lib.call("umb_reduce_residual_norm", part, Sp, T, K, h, h, xn, nw, eps, _lib.dtype_code(dtype))
    S = 1 if epi == 2 else lin.S
    out = (torch.empty(T, lin.N // 2, dtype=dtype, device=dev) if epi == 2
           else torch.empty(S, T, lin.N, dtype=torch.float32, device=dev))
    _lib.call("umb_gemm", out, xn, K, lin.w, lin.meta, T, lin.N, K, lin.awq, S, lin.Rtb, epi, _lib.dtype_code(dtype))
    return h, xn, out


def _pre(lin, part, Sp, h0, nw, eps, T, K, epi, dtype, dev, cols, barrier, alias=False):
    from umbrella_amd import _lib
    S = 1 if epi == 2 else lin.S
    h = h0.clone()
    ssq = torch.zeros(T * (K // cols), dtype=torch.float32, device=dev)
    if epi == 2:
        out = torch.empty(T, lin.N // 2, dtype=dtype, device=dev)
    elif alias:                                   # the model runtime's qkv launch: output over the producer's partials
        buf = torch.empty(max(part.numel(), S * T * lin.N), dtype=torch.float32, device=dev)
        buf[:part.numel()].copy_(part.view(-1))
        part, out = buf, buf
    else:
        out = torch.empty(S, T, lin.N, dtype=torch.float32, device=dev)
    pre = _lib.UmbGemmPre()
    pre.partial, pre.h, pre.norm_w, pre.ssq, pre.barrier = part.data_ptr(), h.data_ptr(), nw.data_ptr(), ssq.data_ptr(), barrier.data_ptr()
    pre.S, pre.cols, pre.eps = Sp, cols, eps
    _lib.call("umb_gemm_pre", out, pre, lin.w, lin.meta, T, lin.N, K, lin.awq, S, lin.Rtb, epi, _lib.dtype_code(dtype))
    torch.cuda.synchronize()
    if alias:
        out = out[:S * T * lin.N].view(S, T, lin.N).clone()
    return h, out


CASES = [   # N, K, awq, epi, cols
    (2048, 1024, 0, 0, 32), (2048, 1024, 1, 0, 32), (3072, 1024, 0, 2, 32), (4096, 1024, 1, 2, 32),
    (10240, 2048, 1, 0, 32), (1024, 512, 0, 0, 64)]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("N,K,awq,epi,cols", CASES)
@pytest.mark.parametrize("T", [1, 13, 16, 31])
def test_gemm_pre_matches_two_kernel_form(dev, dtype, N, K, awq, epi, cols, T):
    from umbrella_amd import _lib
    rs = np.random.RandomState(N + K + T + epi)
    lin = _awq(rs, N, K, dev, epi == 2) if awq else _dense(rs, N, K, dev, dtype, epi == 2)
    S = 1 if epi == 2 else lin.S
    if not _lib.load().umb_gemm_pre_ok(T, N, K, lin.awq, S, lin.Rtb, epi, cols, _lib.dtype_code(dtype)):
        pytest.skip("no producer-finishing variant for this shape")
    Sp = 4
    part = torch.from_numpy(rs.randn(Sp, T, K).astype(np.float32) * 0.3).to(dev)
    h0 = torch.from_numpy(rs.randn(T, K).astype(np.float32)).to(dtype).to(dev)
    nw = (1 + 0.1 * torch.from_numpy(rs.randn(K).astype(np.float32))).to(dtype).to(dev)
    barrier = torch.zeros(2, dtype=torch.int32, device=dev)
    eps = 1e-5
    h_ref, xn_ref, out_ref = _two_kernel(lin, part, Sp, h0, nw, eps, T, K, epi, dtype, dev)
    for rep in range(3):                           # the barrier words reset themselves
        h, out = _pre(lin, part, Sp, h0, nw, eps, T, K, epi, dtype, dev, cols, barrier, alias=(epi == 0 and rep == 2))
        assert int(barrier.abs().sum()) == 0
        assert torch.equal(h, h_ref)               # the residual stream: bit-exact
        # x differs from the two-kernel form only through the summation order of the row's sum of squares (1 ulp of 1/rms)
        a, b = out.float(), out_ref.float()
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= (8 * torch.finfo(dtype).eps if epi == 2 else 2e-3) * scale
    # fp32 restatement of the whole step
    hs = (part.sum(0).to(dtype).float() + h0.float()).to(dtype)
    assert torch.equal(h, hs) or float((h.float() - hs.float()).abs().max()) <= 2 * torch.finfo(dtype).eps * float(hs.float().abs().max())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_gemm_pre_batch_invariance(dev, dtype):
    """a token's h and outputs do not depend on how many rows share the launch (within one token-tile count)"""
    from umbrella_amd import _lib
    rs = np.random.RandomState(5)
    N, K, cols = 2048, 1024, 32
    lin = _dense(rs, N, K, dev, dtype, False)
    Sp, T = 3, 16
    part = torch.from_numpy(rs.randn(Sp, T, K).astype(np.float32) * 0.3).to(dev)
    h0 = torch.from_numpy(rs.randn(T, K).astype(np.float32)).to(dtype).to(dev)
    nw = (1 + 0.1 * torch.from_numpy(rs.randn(K).astype(np.float32))).to(dtype).to(dev)
    barrier = torch.zeros(2, dtype=torch.int32, device=dev)
    if not _lib.load().umb_gemm_pre_ok(T, N, K, 0, lin.S, lin.Rtb, 0, cols, _lib.dtype_code(dtype)):
        pytest.skip("no variant")
    h_full, out_full = _pre(lin, part, Sp, h0, nw, 1e-5, T, K, 0, dtype, dev, cols, barrier)
    for t in (1, 5, 13):
        h, out = _pre(lin, part[:, :t].contiguous(), Sp, h0[:t].contiguous(), nw, 1e-5, t, K, 0, dtype, dev, cols, barrier)
        assert torch.equal(h, h_full[:t]) and torch.equal(out, out_full[:, :t])


def test_gemm_pre_70b_gate_up_full_size(dev):
    """the headline launch: 70B-AWQ gate/up (N = 57344, K = 8192, one 8-wave block per CU) finishing the o-projection"""
    from umbrella_amd import _lib
    rs = np.random.RandomState(11)
    N, K, T, dtype, cols = 57344, 8192, 13, torch.float16, 32
    lin = _awq(rs, N, K, dev, True)
    assert lin.tb == (14 | 0x80)
    assert _lib.load().umb_gemm_pre_ok(T, N, K, 1, 1, lin.Rtb, 2, cols, _lib.dtype_code(dtype))
    Sp = 4
    part = torch.from_numpy(rs.randn(Sp, T, K).astype(np.float32) * 0.3).to(dev)
    h0 = torch.from_numpy(rs.randn(T, K).astype(np.float32)).to(dtype).to(dev)
    nw = (1 + 0.1 * torch.from_numpy(rs.randn(K).astype(np.float32))).to(dtype).to(dev)
    barrier = torch.zeros(2, dtype=torch.int32, device=dev)
    h_ref, _, out_ref = _two_kernel(lin, part, Sp, h0, nw, 1e-5, T, K, 2, dtype, dev)
    for _ in range(2):
        h, out = _pre(lin, part, Sp, h0, nw, 1e-5, T, K, 2, dtype, dev, cols, barrier)
        assert torch.equal(h, h_ref)
        assert float((out.float() - out_ref.float()).abs().max()) <= 8 * torch.finfo(dtype).eps * float(out_ref.float().abs().max())
        assert float((out == out_ref).float().mean()) > 0.98
