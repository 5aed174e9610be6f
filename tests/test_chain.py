"""Persistent chain (csrc/chain.hip): the draft layer as tree attention + ONE persistent launch.

The chain reproduces the five-launch GEMV schedule's arithmetic operation for operation, so everything it writes -- logits,
the residual stream, the K / V^T caches -- must equal the GEMV schedule's BIT FOR BIT on the same weights and the same
prefilled cache (reference lines: umbrella/models/llama.py:75-114 layer_compute, :461-533 LlamaCudagraph); the GEMV schedule
itself is held against fp32 arithmetic and the low-latency schedule in tests/test_gemv.py.  Also: hand-offs are replay-safe
(a captured graph replayed 200 times gives the same bits and leaves the status word 0), and the bounded spins let a launch
that can never complete (a wrong epoch) end with a give-up code instead of hanging."""
import copy
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import __graft_entry__ as ge
    ge.build()
    return torch.device("cuda:0")


def _draft(dev, dtype, layers, chain, seed=0):
    from umbrella_amd.models.config import KNOWN
    from umbrella_amd.models.llama import Llama
    os.environ["UMBRELLA_SYNTHETIC"] = "1"
    cfg = copy.copy(KNOWN["meta-llama/Llama-3.2-1B-Instruct"])
    cfg.num_hidden_layers = layers
    m = Llama("meta-llama/Llama-3.2-1B-Instruct", max_length=256, device=dev, dtype=dtype, config=cfg, seed=seed)
    m.alloc()
    os.environ["UMB_CHAIN"] = "1" if chain else "0"
    try:
        m.use_gemv(True)
    finally:
        os.environ.pop("UMB_CHAIN", None)
    assert m.chain == chain
    return m


def _step(m, dev, T, n_prefill=40):
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(3, 128000, (n_prefill + T,), generator=g, dtype=torch.int32).to(dev)
    m.clear()
    m.prefill_tokens(ids[:n_prefill], 0)
    pos = torch.arange(n_prefill, n_prefill + T, dtype=torch.int32, device=dev)
    pre = torch.tensor([n_prefill], dtype=torch.int32, device=dev)
    m.forward_explicit(ids[n_prefill:].contiguous(), pos, pos, pre, head_from=0)
    torch.cuda.synchronize()
    return m.logits_buffer[:T].clone(), m.hidden_buffer[:T].clone(), m.kv_cache.k.clone(), m.kv_cache.vt.clone()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("T", [1, 2, 3])
def test_chain_equals_the_gemv_schedule_bit_for_bit(dev, dtype, T):
    a = _draft(dev, dtype, 4, True)
    b = _draft(dev, dtype, 4, False)
    la, ha, ka, va = _step(a, dev, T)
    lb, hb, kb, vb = _step(b, dev, T)
    assert a.chain_status() == 0
    assert torch.isfinite(la).all()
    assert torch.equal(ha.view(torch.int16), hb.view(torch.int16)), float((ha.float() - hb.float()).abs().max())
    assert torch.equal(ka.view(torch.int16), kb.view(torch.int16))
    assert torch.equal(va.view(torch.int16), vb.view(torch.int16))
    assert torch.equal(la, lb), float((la - lb).abs().max())


def test_chain_full_depth_replays_from_a_graph(dev):
    """16 layers (the 1B draft's depth), T = 3: one captured forward replayed 200 times -- epochs advance on the device, so
    every replay hands off under fresh tags; logits stay bit-identical and no spin ever gives up"""
    dtype = torch.float16
    m = _draft(dev, dtype, 16, True)
    ref = _draft(dev, dtype, 16, False)
    T = 3
    lr, hr, _, _ = _step(ref, dev, T)
    l0, h0, _, _ = _step(m, dev, T)
    assert torch.equal(l0, lr) and torch.equal(h0.view(torch.int16), hr.view(torch.int16))
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(3, 128000, (40 + T,), generator=g, dtype=torch.int32).to(dev)
    pos = torch.arange(40, 40 + T, dtype=torch.int32, device=dev)
    pre = torch.tensor([40], dtype=torch.int32, device=dev)
    step = ids[40:].contiguous()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        m.forward_explicit(step, pos, pos, pre, head_from=0)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        m.forward_explicit(step, pos, pos, pre, head_from=0)
    for i in range(200):
        gr.replay()
        if i % 50 == 49:
            torch.cuda.synchronize()
            assert torch.equal(m.logits_buffer[:T], lr), i
    torch.cuda.synchronize()
    assert m.chain_status() == 0


def test_chain_gives_up_instead_of_hanging(dev):
    """a hand-off that can never complete (UMB_CHAIN_TEST_DROP_CU: one workgroup withholds its o-projection granules, the
    test hook of chain.hip): every spin is bounded by the wall clock, so the launch ENDS within the timeout, the status
    word carries 0xDEADxxxx, and after umb_chain_xchg_init the same model computes the right bits again"""
    import time
    from umbrella_amd import _lib
    dtype = torch.float16
    m = _draft(dev, dtype, 2, True)
    ref = _draft(dev, dtype, 2, False)
    good = _step(ref, dev, 2)[0]
    assert torch.equal(_step(m, dev, 2)[0], good) and m.chain_status() == 0
    os.environ["UMB_CHAIN_TIMEOUT_MS"] = "5"
    os.environ["UMB_CHAIN_TEST_DROP_CU"] = "200"
    try:
        t0 = time.time()
        _step(m, dev, 2)
        took = time.time() - t0
    finally:
        os.environ.pop("UMB_CHAIN_TIMEOUT_MS", None)
        os.environ.pop("UMB_CHAIN_TEST_DROP_CU", None)
    st = m.chain_status()
    assert (st >> 16) == 0xDEAD, hex(st)
    assert took < 5.0, took
    cfg = m.config
    _lib.check(_lib.load().umb_chain_xchg_init(m._chain_xchg.data_ptr(), 4, cfg.hidden_size, cfg.intermediate_size,
                                               _lib.stream_ptr()))
    assert m.chain_status() == 0
    assert torch.equal(_step(m, dev, 2)[0], good)


def test_chain_rejects_what_it_does_not_cover(dev):
    from umbrella_amd import _lib
    lib = _lib.load()
    assert lib.umb_chain_ok(3, 2048, 8192, 3072, 64, 0) == 1
    assert lib.umb_chain_ok(4, 2048, 8192, 3072, 64, 0) == 0          # 4 rows: GEMV launches
    assert lib.umb_chain_ok(3, 4096, 14336, 6144, 128, 0) == 0         # 8B shapes: split schedule
    assert lib.umb_chain_ok(3, 2048, 8192, 3072, 64, 1) == 0           # q/k/v bias (Qwen2): GEMV launches
    c = _lib.UmbChain()
    assert lib.umb_draft_chain(C_byref(c), 0, None) != 0


def C_byref(x):
    import ctypes as C
    return C.byref(x)
