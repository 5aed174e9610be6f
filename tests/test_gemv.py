"""Row-streaming GEMV family (csrc/gemv.hip, T <= 4 rows): op-level parity through the C ABI.

Each epilogue against an fp32 torch restatement of the reference lines it fuses (llama.py:75-114 residual adds / SiLU * up,
model_utils.py:17-64 RoPE + rmsnorm factor, cache.py:53-65 KV append) and against the low-latency MFMA kernel on the same
weights (q / K / V^T contents, up to the fp32 accumulation order); bitwise batch invariance; the row repack."""
import pytest
import torch

from umbrella_amd.attn.cache import k_from_frag, vt_from_frag      # semantic views of the fragment-ordered KV caches

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import __graft_entry__ as ge
    ge.build()
    return torch.device("cuda:0")


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


def _rows(w, mode=0, D=0, rope_heads=0):
    from umbrella_amd import _lib
    out = torch.empty_like(w)
    _lib.call("umb_repack_rows", out, w, w.shape[0], w.shape[1], mode, D, rope_heads)
    return out


def _ulp(dtype):
    return 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10


def test_repack_rows(dev):
    g = torch.Generator(device=dev).manual_seed(0)
    w = torch.randn(512, 256, device=dev, generator=g).to(torch.float16)
    assert torch.equal(_rows(w), w)
    il = _rows(w, 1)
    assert torch.equal(il[0::2], w[:256]) and torch.equal(il[1::2], w[256:])
    D, heads = 64, 6                                       # 6 rope heads of 64 rows, then 2 plain heads
    rp = _rows(w, 2, D, heads)
    for h in range(8):
        blk, src = rp[h * D:(h + 1) * D], w[h * D:(h + 1) * D]
        if h < heads:
            assert torch.equal(blk[0::2], src[:D // 2]) and torch.equal(blk[1::2], src[D // 2:])
        else:
            assert torch.equal(blk, src)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("N,K", [(2048, 2048), (2048, 8192), (512, 2048)])
@pytest.mark.parametrize("T", [1, 3, 4])
def test_gemv_residual_epilogue(dev, dtype, N, K, T):
    """epi 4: h <- round(round(x W^T) + h); hw <- round(h * w_next); per-workgroup sums of h^2"""
    from umbrella_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(N + K + T)
    w = (torch.randn(N, K, device=dev, generator=g) * 0.03).to(dtype)
    x = torch.randn(T, K, device=dev, generator=g).to(dtype)
    h0 = torch.randn(T, N, device=dev, generator=g).to(dtype)
    nw = (1 + 0.1 * torch.randn(N, device=dev, generator=g)).to(dtype)
    assert lib.umb_gemv_ok(T, N, K, 4) == 1
    groups = lib.umb_gemv_groups(T, N, K)
    stride = groups + 4
    for rep in range(2):
        h, hw = h0.clone(), torch.zeros(T, N, dtype=dtype, device=dev)
        ssq = torch.full((T, stride), -1.0, device=dev)
        fx = _fx(h=h, hw=hw, norm_w=nw, ssq_out=ssq, ssq_out_stride=stride)
        _lib.call("umb_gemv", None, x, _rows(w), T, N, K, 4, fx, _lib.dtype_code(dtype))
        v = (x.float() @ w.float().t()).to(dtype)
        hn = (v.float() + h0.float()).to(dtype)
        scale = float(hn.float().abs().max())
        assert float((h.float() - hn.float()).abs().max()) <= 2 * _ulp(dtype) * scale
        hw_ref = (h.float() * nw.float()).to(dtype)          # from the h the kernel produced: exact
        assert torch.equal(hw, hw_ref)
        got = ssq[:, :groups].sum(1)
        ref = h.float().pow(2).sum(1)
        assert float((got - ref).abs().max()) <= 1e-4 * float(ref.max())
        assert float(ssq[:, groups:].max()) == -1.0           # nothing written past the groups


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("T", [1, 2, 4])
def test_gemv_silu_epilogue(dev, dtype, T):
    """epi 2: [gate; up] rows as (gate_m, up_m) pairs; act = round(silu(round(g / rms)) * round(u / rms))"""
    from umbrella_amd import _lib
    g = torch.Generator(device=dev).manual_seed(T)
    I, K = 8192, 2048
    w = (torch.randn(2 * I, K, device=dev, generator=g) * 0.03).to(dtype)
    x = torch.randn(T, K, device=dev, generator=g).to(dtype)
    G, stride = 70, 80
    ssq = torch.rand(T, stride, device=dev, generator=g) * 40 + 10
    ssq[:, G:] = 1e9                                          # beyond the groups: must not be read
    act = torch.zeros(T, I, dtype=dtype, device=dev)
    fx = _fx(ssq_in=ssq, ssq_groups=G, ssq_in_stride=stride, ssq_dim=float(K), eps=1e-5)
    _lib.call("umb_gemv", act, x, _rows(w, 1), T, 2 * I, K, 2, fx, _lib.dtype_code(dtype))
    inv = torch.rsqrt(ssq[:, :G].sum(1) / K + 1e-5)[:, None]
    full = x.float() @ w.float().t()
    gate, up = (full[:, :I] * inv).to(dtype), (full[:, I:] * inv).to(dtype)
    ref = (torch.nn.functional.silu(gate.float()).to(dtype).float() * up.float()).to(dtype)
    assert float((act.float() - ref.float()).abs().max()) <= 8 * torch.finfo(dtype).eps * float(ref.float().abs().max())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("T", [1, 3, 4])
@pytest.mark.parametrize("bias", [False, True])
def test_gemv_qkv_epilogue_matches_lowlat(dev, dtype, T, bias):
    """epi 3 == the low-latency MFMA kernel's qkv epilogue on the same weights: q rows, K / V^T cache contents"""
    from umbrella_amd import _lib
    from umbrella_amd.models.llama import PackedLinear
    g = torch.Generator(device=dev).manual_seed(T + 100 * bias)
    Hq, Hkv, D, H, Lmax = 32, 8, 64, 2048, 128                # Llama-3.2-1B head geometry
    N = (Hq + 2 * Hkv) * D
    w = (torch.randn(N, H, device=dev, generator=g) * 0.03).to(dtype)
    lin = PackedLinear.from_dense(w, rope=(D, Hq + Hkv))
    x = torch.randn(T, H, device=dev, generator=g).to(dtype)
    b = (torch.randn(N, device=dev, generator=g) * 0.1).to(dtype) if bias else None
    pos = torch.randint(0, Lmax, (T,), device=dev, generator=g, dtype=torch.int32)
    slot = torch.randperm(Lmax, device=dev, generator=g)[:T].to(torch.int32)
    ang = torch.rand(Lmax, D, device=dev, generator=g) * 6.28
    cos, sin = torch.cos(ang).to(dtype), torch.sin(ang).to(dtype)
    ssq = torch.rand(T, 32, device=dev, generator=g) * 40 + 10

    def caches():
        return (torch.zeros(T, Hq, D, dtype=dtype, device=dev), torch.zeros(Hkv, Lmax, D, dtype=dtype, device=dev),
                torch.zeros(Hkv, D, Lmax + 32, dtype=dtype, device=dev))
    kw = dict(pos=pos, slot=slot, cosT=cos, sinT=sin, Hq=Hq, Hkv=Hkv, D=D, Lmax=Lmax, ssq_in=ssq, ssq_groups=32,
              ssq_in_stride=32, ssq_dim=float(H), eps=1e-5, **({"bias": b} if bias else {}))
    q1, k1, v1 = caches()
    lin.apply_ll(x, fx=_fx(q_out=q1, k_cache=k1, vt_cache=v1, **kw), epi=3)
    q2, k2, v2 = caches()
    _lib.call("umb_gemv", None, x, _rows(w, 2, D, Hq + Hkv), T, N, H, 3, _fx(q_out=q2, k_cache=k2, vt_cache=v2, **kw),
              _lib.dtype_code(dtype))
    for a_, b_ in ((q1, q2), (k1, k2), (v1, v2)):
        scale = float(a_.float().abs().max())
        assert float((a_.float() - b_.float()).abs().max()) <= 2 * _ulp(dtype) * scale
    free = torch.ones(Lmax, dtype=torch.bool, device=dev)
    free[slot.long()] = False
    k2s, v2s = k_from_frag(k2), vt_from_frag(v2)              # semantic views of the fragment-ordered caches
    assert float(k2s[:, free].abs().max()) == 0.0 and float(v2s[:, :, :Lmax][:, :, free].abs().max()) == 0.0
    assert float(k2s[:, ~free].abs().max()) > 0


def test_gemv_batch_invariance(dev):
    """a token's results do not depend on how many rows share the launch"""
    from umbrella_amd import _lib
    g = torch.Generator(device=dev).manual_seed(9)
    dtype, N, K = torch.bfloat16, 2048, 8192
    w = _rows((torch.randn(N, K, device=dev, generator=g) * 0.03).to(dtype))
    x = torch.randn(4, K, device=dev, generator=g).to(dtype)
    h0 = torch.randn(4, N, device=dev, generator=g).to(dtype)
    nw = (1 + 0.1 * torch.randn(N, device=dev, generator=g)).to(dtype)

    def run(T):
        groups = _lib.load().umb_gemv_groups(T, N, K)
        h, hw = h0[:T].clone(), torch.zeros(T, N, dtype=dtype, device=dev)
        ssq = torch.zeros(T, groups, device=dev)
        _lib.call("umb_gemv", None, x[:T].contiguous(), w, T, N, K, 4, _fx(h=h, hw=hw, norm_w=nw, ssq_out=ssq, ssq_out_stride=groups),
                  _lib.dtype_code(dtype))
        return h, hw, ssq
    full = run(4)
    for T in (1, 2, 3):
        part = run(T)
        assert all(torch.equal(a, b[:T]) for a, b in zip(part, full))



def test_gemv_rejects_what_it_does_not_cover(dev):
    from umbrella_amd import _lib
    lib = _lib.load()
    assert lib.umb_gemv_ok(5, 2048, 2048, 4) == 0 and lib.umb_gemv_ok(3, 2048, 4096, 4) == 0
    assert lib.umb_gemv_ok(3, 2048, 2048, 0) == 0 and lib.umb_gemv_ok(3, 3072, 2048, 3) == 1


def _draft_logits(dev, gemv, T, dtype, layers=4):
    """Llama-3.2-1B shapes (seeded synthetic weights, `layers` decoder layers): prefill 40 tokens, then a T-row step"""
    import copy
    import os
    from umbrella_amd.models.config import KNOWN
    from umbrella_amd.models.llama import Llama
    os.environ["UMBRELLA_SYNTHETIC"] = "1"
    os.environ["UMB_GEMV"] = "1" if gemv else "0"
    try:
        cfg = copy.copy(KNOWN["meta-llama/Llama-3.2-1B-Instruct"])
        cfg.num_hidden_layers = layers
        m = Llama("meta-llama/Llama-3.2-1B-Instruct", max_length=256, device=dev, dtype=dtype, config=cfg)
        m.alloc()
    finally:
        os.environ.pop("UMB_GEMV", None)
    assert (getattr(m.layers[0]["qkv"], "w_rows", None) is not None) == gemv
    assert m._layer_structs[0].qkv.w_rows is None, "GEMV is opt-in: a freshly allocated model publishes no row-major copies"
    if gemv:
        m.use_gemv(True)                    # the engines do this for their draft
        assert m._layer_structs[0].qkv.w_rows and m._layer_structs[3].down.w_rows
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(3, 128000, (40 + T,), generator=g, dtype=torch.int32).to(dev)
    m.prefill_tokens(ids[:40], 0)
    pos = torch.arange(40, 40 + T, dtype=torch.int32, device=dev)
    pre = torch.tensor([40], dtype=torch.int32, device=dev)
    m.forward_explicit(ids[40:].contiguous(), pos, pos, pre, head_from=0)
    torch.cuda.synchronize()
    return m.logits_buffer[:T].clone()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("T", [1, 3, 4])
def test_gemv_schedule_model_logits_match_lowlat_schedule(dev, dtype, T):
    """the <= 4-row forward on the GEMV schedule == the same forward on the low-latency MFMA schedule (same weights, same
    prefilled cache) up to accumulation order: logits within 16-bit rounding noise, same arg-max where the margin allows"""
    a = _draft_logits(dev, True, T, dtype)
    b = _draft_logits(dev, False, T, dtype)
    scale = float(b.abs().max())
    tol = (6e-2 if dtype == torch.bfloat16 else 1e-2) * scale
    assert float((a - b).abs().max()) <= tol, (float((a - b).abs().max()), scale)
    top2 = b.topk(2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 2 * tol
    assert torch.equal(a.argmax(-1)[clear], b.argmax(-1)[clear])


def test_gemv_is_a_draft_role_only(dev):
    """a target of Llama-3.2-1B shapes (K = 2048 / 8192: the shapes the GEMV family serves) must keep ONE kernel
    path for every T <= 64, or its T = 1 (autoregressive) and T = 13 (tree verify) logits of the same token differ and
    greedy spec != greedy AR at near-ties.  The engines flag roles: draft -> GEMV on, target -> off; the target's row of
    a 1-row forward is then bit-identical to the same token's row in a 3-row forward."""
    import copy
    import os
    from umbrella_amd.models.config import KNOWN
    from umbrella_amd.models.llama import Llama
    from umbrella_amd.sequoia_utils import generate_sequoia_tree
    from umbrella_amd.speculation.speculation_utils import IdTokenizer
    from umbrella_amd.speculation.static_speculation_engine import StaticSpeculationEngine
    os.environ["UMBRELLA_SYNTHETIC"] = "1"
    cfg = copy.copy(KNOWN["meta-llama/Llama-3.2-1B-Instruct"])
    cfg.num_hidden_layers = 2
    dtype = torch.float16
    mk = lambda seed, **kw: Llama("meta-llama/Llama-3.2-1B-Instruct", max_length=256, device=dev, dtype=dtype, config=cfg,
                                  seed=seed, **kw)
    target, draft = mk(1), mk(2, cuda_graph=True)
    target.alloc(); draft.alloc()
    eng = StaticSpeculationEngine("d", "t", dtype=dtype, device=str(dev), growmap=generate_sequoia_tree(3, 4), max_length=256,
                                  draft_model_obj=draft, target_model_obj=target, tokenizer=IdTokenizer())
    eng.initialize()
    assert target._layer_structs[0].qkv.w_rows is None and target._layer_structs[1].down.w_rows is None
    assert draft._layer_structs[0].qkv.w_rows and draft._layer_structs[1].down.w_rows
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(3, 128000, (43,), generator=g, dtype=torch.int32).to(dev)
    pre = torch.tensor([40], dtype=torch.int32, device=dev)
    rows = {}
    for T in (1, 3):
        target.clear()
        target.prefill_tokens(ids[:40], 0)
        pos = torch.arange(40, 40 + T, dtype=torch.int32, device=dev)
        target.forward_explicit(ids[40:40 + T].contiguous(), pos, pos, pre, head_from=0)
        rows[T] = target.logits_buffer[0].clone()
    assert torch.equal(rows[1], rows[3]), "a target's logits must not depend on the rows sharing its launch"
    out = eng.generate(input_ids=ids[:40].tolist(), max_new_tokens=16)
    assert len(out["generated_tokens"]) >= 16
