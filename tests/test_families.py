"""Qwen2 (q/k/v bias, odd GQA group) and Mistral (head_dim != hidden/heads) runtimes.

CPU: the oracle replays traces recorded from the reference's own Qwen / Mistral classes
(tests/golden/make_golden_families.py).  GPU: the HIP runtime against those vectors and the fp32 oracle."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLD, oracle_model, replay_case
from oracle.engine import OracleDynamicEngine, OracleStaticEngine

with open(os.path.join(GOLD, "families.json")) as f:
    F = json.load(f)
LOG = np.load(os.path.join(GOLD, "families_logits.npz"))
with open(os.path.join(GOLD, "growmaps.json")) as f:
    GM = json.load(f)
KINDS = ("qwen", "mistral")


def _oracle(kind, L, **kw):
    return oracle_model(F["cfg"][kind], F["seeds"][kind], L, **kw)


@pytest.mark.parametrize("kind", KINDS)
def test_oracle_logits_match_reference_family(kind):
    assert F[kind + "_hf_vs_ref_max_abs"] < 2e-4                 # the reference class itself == HF forward
    m = _oracle(kind, 128)
    ids = torch.tensor([F["prompt"]])
    n = ids.shape[1]
    mask = torch.tril(torch.ones(n, 128, dtype=torch.bool))
    logits = m.inference(ids, torch.arange(n)[None], mask, torch.arange(n))
    np.testing.assert_allclose(logits[0, -1].numpy(), LOG[kind + "_logits_last"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(logits[0, ::5, :64].numpy(), LOG[kind + "_logits_rows"], rtol=2e-4, atol=2e-4)
    from umbrella_amd.models.config import LlamaCfg, rope_inv_freq
    np.testing.assert_allclose(rope_inv_freq(LlamaCfg(**F["cfg"][kind]))[0].numpy(), LOG[kind + "_inv_freq"], rtol=1e-6)


@pytest.mark.parametrize("case", sorted(F["cases"].keys()))
def test_oracle_replays_reference_family_trace(case):
    rec = F["cases"][case]
    kind = case.split("_")[0]
    dkind = "qwen" if "qwendraft" in case else kind
    c = rec["config"]
    L = c["max_length"]
    target = _oracle(kind, L)
    if c["engine"] == "static":
        eng = OracleStaticEngine(_oracle(dkind, L, slot_cache=True), target, GM["3x4"], rec["eos"], max_length=L,
                                 safe_buffer=c["safe_buffer"])
    else:
        eng = OracleDynamicEngine(_oracle(dkind, L), target, rec["eos"], width=c["width"], depth=c["depth"],
                                  num_beams=c["num_beams"], max_length=L, safe_buffer=c["safe_buffer"])
    replay_case(rec, eng, case)
    toks = rec["turns"][0]["tokens"]
    assert toks == F[kind + "_hf_greedy"][:len(toks)]            # greedy speculation == HF greedy AR


def test_family_registry():
    """The reference's Qwen2.5 / Mistral hub ids resolve (auto_model.py:21-55,80-154) with the right traits."""
    from umbrella_amd.models.config import KNOWN
    q = KNOWN["Qwen/Qwen2.5-Coder-7B-Instruct-AWQ"]
    assert q.attention_bias and q.awq and q.vocab_size == 151936 and q.num_attention_heads // q.num_key_value_heads == 7
    m = KNOWN["mistralai/Mistral-Small-24B-Instruct-2501"]
    assert not m.attention_bias and m.head_dim * m.num_attention_heads != m.hidden_size
    for name in ("Qwen/Qwen2.5-0.5B-Instruct", "Qwen/Qwen2.5-72B-Instruct-AWQ", "Qwen/QwQ-32B-Preview",
                 "casperhansen/deepseek-r1-distill-qwen-32b-awq", "mistralai/Mistral-7B-Instruct-v0.3",
                 "solidrust/Mistral-7B-Instruct-v0.3-AWQ", "stelterlab/Mistral-Small-24B-Instruct-2501-AWQ"):
        assert name in KNOWN


# ----------------------------------------------------------------------------- GPU
def _hip(kind, device, dtype, max_length=256, **kw):
    from umbrella_amd.models.config import LlamaCfg
    from umbrella_amd.models.llama import Llama
    from umbrella_amd.models.synthetic import synth_state_small
    cfg = LlamaCfg(**dict(F["cfg"][kind], eos_token_id=[3, 5]))
    sd = synth_state_small(cfg, F["seeds"][kind])
    alloc_kw = {k: kw.pop(k) for k in ("exit_layer", "num_cache_layers") if k in kw}
    m = Llama("tiny-" + kind, max_length=max_length, device=str(device), dtype=dtype, state_dict=sd, config=cfg, **kw)
    m.alloc(**alloc_kw)
    return m, sd


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_hip_family_logits(kind, dtype):
    """HIP runtime vs the reference's recorded fp32 logits (16-bit tolerance), resident and offloaded."""
    dev = torch.device("cuda:0")
    ids = torch.tensor([F["prompt"]])
    n = ids.shape[1]
    mask = torch.tril(torch.ones(n, 256, dtype=torch.bool))
    # logits reach |x| ~ 70 here and are rounded to the model dtype (F.linear(...).float(), llama.py:133):
    # absolute slack as for the Llama tests plus 2 ulp of the output format
    tol, rel = (0.06, 2.0 ** -9) if dtype == torch.float16 else (0.35, 2.0 ** -6)
    outs = []
    for offload in (False, True):
        m, _ = _hip(kind, dev, dtype, offload=offload)
        logits = m.inference(ids, torch.arange(n)[None], mask, torch.arange(n))[0].cpu()
        for got, ref in ((logits[-1], torch.from_numpy(LOG[kind + "_logits_last"])),
                         (logits[::5, :64], torch.from_numpy(LOG[kind + "_logits_rows"]))):
            # the bias-heavy tiny Qwen has |logit| ~ 94 and a 16-bit CPU oracle is itself 0.85 (bf16) off the fp32
            # reference there: the absolute slack scales with the logit range (Llama fixtures: range ~ 16)
            scale = max(1.0, float(ref.abs().max()) / 16.0)
            assert ((got - ref).abs() <= tol * scale + ref.abs() * rel).all(), float((got - ref).abs().max())
        outs.append(logits)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(F["cases"].keys()))
def test_hip_family_engines(case):
    """Both engines on the Qwen2 / Mistral runtimes: every token is a greedy choice of the fp32 oracle target."""
    from umbrella_amd.speculation.dynamic_speculation_engine import DynamicSpeculationEngine
    from umbrella_amd.speculation.speculation_utils import IdTokenizer
    from umbrella_amd.speculation.static_speculation_engine import StaticSpeculationEngine
    dev = torch.device("cuda:0")
    dtype = torch.float16
    rec = F["cases"][case]
    kind = case.split("_")[0]
    dkind = "qwen" if "qwendraft" in case else kind
    c = rec["config"]
    target, tsd = _hip(kind, dev, dtype)
    draft, _ = _hip(dkind, dev, dtype, cuda_graph=(c["engine"] == "static"))
    common = dict(dtype=dtype, device=str(dev), max_length=256, safe_buffer=16, stop_distance=8, draft_model_obj=draft,
                  target_model_obj=target, tokenizer=IdTokenizer())
    if c["engine"] == "static":
        eng = StaticSpeculationEngine("d", "t", growmap=GM["3x4"], **common)
    else:
        eng = DynamicSpeculationEngine("d", "t", width=c["width"], num_beams=c["num_beams"], depth=c["depth"], **common)
    eng.initialize()
    out = eng.generate(input_ids=rec["prompt"], max_new_tokens=rec["max_new_tokens"])
    toks = out["generated_tokens"]
    seq = list(rec["prompt"]) + toks
    o = oracle_model(F["cfg"][kind], F["seeds"][kind], len(seq) + 1, torch.float32, state=tsd)
    n = len(seq)
    logits = o.inference(torch.tensor([seq]), torch.arange(n)[None], torch.tril(torch.ones(n, n + 1, dtype=torch.bool)),
                         torch.arange(n))[0]
    for i, tok in enumerate(toks):
        row = logits[len(rec["prompt"]) + i - 1].clone()
        if i == 0 and c["engine"] == "dynamic":
            row[[3, 5]] = -float("inf")
        assert float(row.max() - row[tok]) <= 0.06, (i, tok)
    if "selfdraft" in case:
        assert out["avg_accept_tokens"] > 2.5
