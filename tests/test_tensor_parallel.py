"""Tensor-parallel verify (umbrella_amd/tensor_parallel.py).

CPU: the sharding helpers against the unsharded oracle -- algebra per linear (dense and AutoAWQ tensors) and a
world-2 gloo run of a whole tiny-model forward with all-reduces where the module puts them.
GPU (single process, LocalComm: the all-reduce is a sum over the in-process shards): HIP shards reproduce the
unsharded HIP model up to fp32 summation order; the engine emits the fp32 oracle's greedy tokens."""
import os
import socket

import pytest
import torch

from helpers import load_golden, oracle_model
from oracle import ops as O
from umbrella_amd.models.config import LlamaCfg, rope_inv_freq
from umbrella_amd.models.synthetic import linear_shapes, synth_awq_small, synth_state_small
from umbrella_amd.tensor_parallel import local_config, shard_range, shard_state_dict

G = load_golden()
TCFG = dict(G["target_cfg"])


def _cfg(**kw):
    return LlamaCfg(**dict(TCFG, eos_token_id=[3, 5], **kw))


@pytest.mark.parametrize("awq", [False, True])
@pytest.mark.parametrize("world", [2])
def test_sharded_linears_recompose(awq, world):
    """Column-split linears concatenate, row-split ones sum to the unsharded linear -- for dense weights and for AutoAWQ
    qweight / qzeros / scales slices (whose dequantisation must be the slice of the full dequantisation)."""
    cfg = _cfg(awq=awq)
    sd = synth_awq_small(cfg, 3) if awq else synth_state_small(cfg, 3)
    shards = [shard_state_dict(sd, cfg, r, world) for r in range(world)]
    g = torch.Generator().manual_seed(0)

    def dense(d, base):
        if awq:
            return O.awq_dequant(d[base + ".qweight"], d[base + ".qzeros"], d[base + ".scales"], cfg.awq_group).float().t()
        return d[base + ".weight"].float()
    for name, (n, k) in linear_shapes(cfg).items():
        base = "model.layers.1." + name
        full = dense(sd, base)                                        # [N, K]
        parts = [dense(s, base) for s in shards]
        x = torch.randn(5, k, generator=g)
        if name in ("self_attn.o_proj", "mlp.down_proj"):             # row split: K sliced, outputs summed (the all-reduce)
            ks = [shard_range(k, r, world) for r in range(world)]
            y = sum(x[:, lo:hi] @ p.t() for (lo, hi), p in zip(ks, parts))
            assert all(p.shape == (n, k // world) for p in parts)
        else:                                                          # column split: N sliced, outputs concatenated
            y = torch.cat([x @ p.t() for p in parts], dim=-1)
            assert all(p.shape == (n // world, k) for p in parts)
        assert torch.allclose(y, x @ full.t(), rtol=1e-5, atol=1e-5)
    lc = local_config(cfg, world)
    assert (lc.num_attention_heads, lc.num_key_value_heads, lc.intermediate_size, lc.vocab_size) == \
        (cfg.num_attention_heads // world, cfg.num_key_value_heads // world, cfg.intermediate_size // world, cfg.vocab_size // world)
    assert torch.equal(torch.cat([s["lm_head.weight"] for s in shards]), sd["lm_head.weight"])


def _tp_forward_torch(cfg, sd_local, rank, world, ids, pos, mask, all_reduce):
    """fp32 torch restatement of TensorParallelLlama._forward on one rank (dense weights): local heads, all-reduce after
    o_proj and down_proj, vocabulary-sharded logits."""
    import torch.nn.functional as F
    lc = local_config(cfg, world)
    inv, scl = rope_inv_freq(cfg)
    cos, sin = O.rope_cache(inv, scl, 256, torch.float32)
    n = ids.shape[0]
    h = F.embedding(ids, sd_local["model.embed_tokens.weight"]).float()
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        x = O.rmsnorm(h, sd_local[p + "input_layernorm.weight"].float(), cfg.rms_norm_eps)
        q = (x @ sd_local[p + "self_attn.q_proj.weight"].float().t()).view(n, lc.num_attention_heads, cfg.head_dim)
        k = (x @ sd_local[p + "self_attn.k_proj.weight"].float().t()).view(n, lc.num_key_value_heads, cfg.head_dim)
        v = (x @ sd_local[p + "self_attn.v_proj.weight"].float().t()).view(n, lc.num_key_value_heads, cfg.head_dim)
        q, k = O.apply_rope(q, k, cos, sin, pos)
        a = O.masked_attention(q, k, v, mask).reshape(n, -1)
        part = a @ sd_local[p + "self_attn.o_proj.weight"].float().t()
        all_reduce(part)
        h = h + part
        x = O.rmsnorm(h, sd_local[p + "post_attention_layernorm.weight"].float(), cfg.rms_norm_eps)
        act = F.silu(x @ sd_local[p + "mlp.gate_proj.weight"].float().t()) * (x @ sd_local[p + "mlp.up_proj.weight"].float().t())
        part = act @ sd_local[p + "mlp.down_proj.weight"].float().t()
        all_reduce(part)
        h = h + part
    return O.rmsnorm(h, sd_local["model.norm.weight"].float(), cfg.rms_norm_eps) @ sd_local["lm_head.weight"].float().t()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _tp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from umbrella_amd.tensor_parallel import DistComm
    cfg = _cfg()
    sd = synth_state_small(cfg, G["seeds"]["target"])
    comm = DistComm()
    local = shard_state_dict(sd, cfg, rank, world)
    n = 12
    ids = torch.tensor(G["cases"]["static_3x4"]["prompt"][:n])
    pos = torch.arange(n)
    mask = torch.tril(torch.ones(n, n, dtype=torch.bool))
    logits = _tp_forward_torch(cfg, local, rank, world, ids, pos, mask, lambda t: comm.all_reduce([t]))
    vals, idx = logits.max(dim=-1)
    best = comm.gather_max([vals], [idx.int()], cfg.vocab_size // world)[0]
    q.put((rank, logits, best))
    dist.barrier()
    dist.destroy_process_group()


def test_tp_forward_gloo_world2():
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = _cfg()
    n = 12
    m = oracle_model(TCFG, G["seeds"]["target"], n + 1, torch.float32)
    ids = torch.tensor([G["cases"]["static_3x4"]["prompt"][:n]])
    ref = m.inference(ids, torch.arange(n)[None], torch.tril(torch.ones(n, n + 1, dtype=torch.bool)), torch.arange(n))[0]
    full = torch.cat([g[1] for g in got], dim=-1)                    # vocabulary shards side by side
    assert torch.allclose(full, ref, rtol=2e-4, atol=2e-4), float((full - ref).abs().max())
    for g in got:                                                     # every rank ends with the same global arg-max ids
        assert torch.equal(g[2].long(), ref.argmax(-1))


# ------------------------------------------------------------------ GPU: HIP shards in one process
@pytest.fixture(scope="module")
def dev():
    import __graft_entry__ as ge
    ge.build()
    return torch.device("cuda:0")


def _build_tp(dev, dtype, awq, world, max_length=256):
    from umbrella_amd.tensor_parallel import LocalComm, TensorParallelLlama
    cfg = _cfg(awq=awq)
    sd = synth_awq_small(cfg, G["seeds"]["target"]) if awq else synth_state_small(cfg, G["seeds"]["target"])
    tp = TensorParallelLlama.build(cfg, sd, world, LocalComm(world), max_length, str(dev), dtype, ranks=list(range(world)))
    return tp, cfg, sd


@pytest.mark.gpu
@pytest.mark.parametrize("awq", [False, True])
def test_tp_partials_sum_to_unsplit_linear(dev, awq):
    """Row-split o_proj / down_proj on the HIP kernels: the P partial products sum to the unsplit linear's output up to
    fp32 summation order; column-split q / gate rows are the unsplit rows exactly (same K, same kernel)."""
    from umbrella_amd.models.llama import PackedLinear
    dtype = torch.float16
    cfg = _cfg(awq=awq)
    sd = synth_awq_small(cfg, 5) if awq else synth_state_small(cfg, 5)
    world = 2
    shards = [shard_state_dict(sd, cfg, r, world) for r in range(world)]
    gen = torch.Generator(device=dev).manual_seed(1)

    def packed(d, base):
        if awq:
            return PackedLinear.from_awq(d[base + ".qweight"].to(dev), d[base + ".qzeros"].to(dev), d[base + ".scales"].to(dev))
        return PackedLinear.from_dense(d[base + ".weight"].to(dev).to(dtype))
    for name in ("self_attn.o_proj", "mlp.down_proj"):
        base = "model.layers.0." + name
        n, k = linear_shapes(cfg)[name]
        x = (torch.randn(13, k, device=dev, generator=gen) * 0.5).to(dtype)
        full = packed(sd, base).apply_ll(x)
        parts = sum(packed(s, base).apply_ll(x[:, lo:hi].contiguous())
                    for s, (lo, hi) in zip(shards, (shard_range(k, r, world) for r in range(world))))
        assert float((parts - full).abs().max()) <= 2e-5 * float(full.abs().max()) + 1e-6
    base = "model.layers.0.mlp.gate_proj"
    n, k = linear_shapes(cfg)["mlp.gate_proj"]
    x = (torch.randn(13, k, device=dev, generator=gen) * 0.5).to(dtype)
    full = packed(sd, base).apply_ll(x)
    cat = torch.cat([packed(s, base).apply_ll(x) for s in shards], dim=-1)
    assert torch.equal(cat, full)


@pytest.mark.gpu
@pytest.mark.parametrize("awq", [False, True])
def test_tp_model_matches_unsharded_model(dev, awq):
    """Prefix + a 13-node tree through two HIP shards (LocalComm) vs the unsharded HIP model: same arg-max ids wherever
    the unsharded fp32-rounded logits have a clear margin, and the residual streams agree to 16-bit noise."""
    from hip_helpers import growmap, hip_model
    from umbrella_amd.models.llama import pack_mask_bits
    dtype = torch.float16
    tp, cfg, sd = _build_tp(dev, dtype, awq, 2)
    full, _ = hip_model(TCFG, G["seeds"]["target"], 256, dtype, dev, awq=awq)
    gm = growmap("3x4")
    P, T = 24, gm["size"]
    prompt = torch.tensor(G["cases"]["static_3x4"]["prompt"][:P], dtype=torch.int32, device=dev)
    tree = torch.randint(6, 500, (T,), dtype=torch.int32, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    first = tp.prefill_tokens(prompt, 0)
    row = full.prefill_tokens(prompt, 0)
    assert int(first) == int(row.argmax())
    tokens = torch.zeros(256 + T + 8, dtype=torch.int32, device=dev)
    tokens[:P] = prompt
    tokens[P:P + T] = tree
    n_dev = torch.tensor([P], dtype=torch.int32, device=dev)
    depth = torch.tensor(gm["depth"], dtype=torch.int32, device=dev)
    bits = pack_mask_bits((torch.tensor(gm["mask"]) == 1).to(dev)).contiguous()
    tp.forward_tree(tokens, n_dev, depth, 0, T, bits, bits.shape[1], head_from=0)
    full.forward_tree(tokens, n_dev, depth, 0, T, bits, bits.shape[1], head_from=0)
    ref = full.logits_buffer[:T]
    top2 = ref.topk(2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > (0.25 if awq else 0.12)
    assert int(clear.sum()) >= T // 2
    assert torch.equal(tp.sampled_ids[:T][clear].long(), ref.argmax(-1)[clear])
    h_tp, h_full = tp.shards[0]._bufs["h"][:T].float(), full._bufs["h"][:T].float()
    assert torch.equal(tp.shards[0]._bufs["h"][:T], tp.shards[1]._bufs["h"][:T])      # the residual stream is replicated
    assert float((h_tp - h_full).abs().max()) <= 0.03 * float(h_full.abs().max())


@pytest.mark.gpu
def test_tp_engine_greedy_tokens(dev):
    """TensorParallelStaticEngine over two in-process HIP shards: every emitted token is a greedy choice of the fp32
    oracle target, acceptance works (self-draft), and a second request on the same engine repeats the first."""
    from hip_helpers import check_greedy, growmap, hip_model
    from umbrella_amd.speculation.speculation_utils import IdTokenizer
    from umbrella_amd.tensor_parallel import TensorParallelStaticEngine
    dtype = torch.float16
    tp, cfg, sd = _build_tp(dev, dtype, False, 2)
    draft, _ = hip_model(TCFG, G["seeds"]["target"], 256, dtype, dev, cuda_graph=True)
    eng = TensorParallelStaticEngine("tiny-draft", "tiny-target", dtype=dtype, device=str(dev), growmap=growmap("3x4"),
                                     max_length=256, safe_buffer=16, stop_distance=8, draft_model_obj=draft, tp_target=tp,
                                     tokenizer=IdTokenizer())
    eng.initialize()
    prompt = G["cases"]["static_3x4"]["prompt"]
    out = eng.generate(input_ids=prompt, max_new_tokens=32)
    check_greedy(G, sd, prompt, out["generated_tokens"], dtype)
    assert out["avg_accept_tokens"] > 2.5
    again = eng.generate(input_ids=prompt, max_new_tokens=32)
    assert again["generated_tokens"] == out["generated_tokens"]
    with pytest.raises(ValueError):
        eng.update_generation_args(temperature=0.7)
