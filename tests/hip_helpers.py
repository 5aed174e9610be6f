"""GPU-side test helpers: tiny HIP models / engines and their oracle counterparts."""
import json
import os

import torch

from helpers import GOLD, load_golden, oracle_model
from umbrella_amd.models.config import LlamaCfg
from umbrella_amd.models.llama import Llama
from umbrella_amd.models.synthetic import synth_awq_small, synth_state_small
from umbrella_amd.speculation.dynamic_speculation_engine import DynamicSpeculationEngine
from umbrella_amd.speculation.speculation_utils import IdTokenizer
from umbrella_amd.speculation.static_speculation_engine import StaticSpeculationEngine

TOL = {torch.bfloat16: 0.35, torch.float16: 0.06}


def growmap(name="3x4"):
    with open(os.path.join(GOLD, "growmaps.json")) as f:
        return json.load(f)[name]


def hip_model(cfgd, seed, max_length, dtype, device, awq=False, eos=(3, 5), **kw):
    cfg = LlamaCfg(**dict(cfgd, eos_token_id=list(eos), awq=awq))
    sd = synth_awq_small(cfg, seed) if awq else synth_state_small(cfg, seed)
    alloc_kw = {k: kw.pop(k) for k in ("exit_layer", "num_cache_layers", "layer_range") if k in kw}
    m = Llama("tiny", max_length=max_length, device=str(device), dtype=dtype, state_dict=sd, config=cfg, **kw)
    m.alloc(**alloc_kw)
    return m, sd


def static_engine(g, device, dtype, self_draft=True, gm="3x4", max_length=256, safe_buffer=16, awq=False,
                  eos=(3, 5), hip_graph=True, draft_exit_layer=None, **kw):
    tcfg, tseed = g["target_cfg"], g["seeds"]["target"]
    dcfg, dseed = (tcfg, tseed) if self_draft else (g["draft_cfg"], g["seeds"]["draft"])
    target, tsd = hip_model(tcfg, tseed, max_length, dtype, device, awq=awq, eos=eos)
    dkw = {"exit_layer": draft_exit_layer} if draft_exit_layer else {}
    draft, _ = hip_model(dcfg, dseed, max_length, dtype, device, awq=awq and self_draft, eos=eos, cuda_graph=True, **dkw)
    eng = StaticSpeculationEngine("tiny-draft", "tiny-target", dtype=dtype, device=str(device), growmap=growmap(gm),
                                  max_length=max_length, safe_buffer=safe_buffer, stop_distance=8,
                                  draft_model_obj=draft, target_model_obj=target, tokenizer=IdTokenizer(),
                                  hip_graph=hip_graph, **kw)
    eng.initialize()
    return eng, tsd


def dynamic_engine(g, device, dtype, self_draft=True, width=8, num_beams=8, depth=4, max_length=256, safe_buffer=16,
                   eos=(3, 5), offload=False, hip_graph=True, num_cache_layers=0, awq=False, **kw):
    tcfg, tseed = g["target_cfg"], g["seeds"]["target"]
    dcfg, dseed = (tcfg, tseed) if self_draft else (g["draft_cfg"], g["seeds"]["draft"])
    target, tsd = hip_model(tcfg, tseed, max_length, dtype, device, eos=eos, offload=offload,
                            num_cache_layers=num_cache_layers, awq=awq)
    draft, _ = hip_model(dcfg, dseed, max_length, dtype, device, eos=eos, awq=awq and self_draft)
    eng = DynamicSpeculationEngine("tiny-draft", "tiny-target", dtype=dtype, device=str(device), width=width,
                                   num_beams=num_beams, depth=depth, max_length=max_length, safe_buffer=safe_buffer,
                                   stop_distance=8, draft_model_obj=draft, target_model_obj=target,
                                   tokenizer=IdTokenizer(), offload=offload, hip_graph=hip_graph, **kw)
    eng.initialize()
    return eng, tsd


def check_greedy(g, state, prompt, generated, dtype, mask_first_eos=None, tol=None):
    """Every generated token must be an arg-max of the fp32 oracle target given its prefix, within the
    stated logit tolerance (near-ties may break differently in 16-bit arithmetic)."""
    tol = TOL[dtype] if tol is None else tol
    seq = list(prompt) + list(generated)
    m = oracle_model(g["target_cfg"], g["seeds"]["target"], len(seq) + 1, torch.float32, state=_dense_state(state))
    ids = torch.tensor([seq])
    n = len(seq)
    mask = torch.tril(torch.ones(n, n + 1, dtype=torch.bool))
    logits = m.inference(ids, torch.arange(n)[None], mask, torch.arange(n))[0]
    worst = 0.0
    for i, tok in enumerate(generated):
        row = logits[len(prompt) + i - 1].clone()
        if i == 0 and mask_first_eos:
            row[list(mask_first_eos)] = -float("inf")
        gap = float(row.max() - row[tok])
        worst = max(worst, gap)
        assert gap <= tol, f"token {i} ({tok}) is not a greedy choice: logit gap {gap:.4f} > {tol}"
    return worst


def _dense_state(sd):
    """fp32 dense view of a (possibly AWQ) tiny state dict for the oracle."""
    return sd


def run_smoke(device):
    g = load_golden()
    dtype = torch.bfloat16
    eng, sd = static_engine(g, device, dtype, self_draft=True)
    prompt = g["cases"]["static_3x4_selfdraft"]["prompt"]
    out = eng.generate(input_ids=prompt, max_new_tokens=24)
    toks = out["generated_tokens"]
    assert len(toks) >= 24, toks
    worst = check_greedy(g, sd, prompt, toks, dtype)
    assert out["avg_accept_tokens"] > 2.0, out["avg_accept_tokens"]
    print(f"smoke: {len(toks)} tokens, accept {out['avg_accept_tokens']:.2f}, worst logit gap {worst:.4f}")
