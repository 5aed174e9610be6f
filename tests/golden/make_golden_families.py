#!/usr/bin/env python
"""Golden vectors for the Qwen2 and Mistral runtimes, produced by running the REFERENCE's own classes
(umbrella/models/qwen.py, mistral.py and both engines) on CPU -- same shims as make_golden.py.

    python tests/golden/make_golden_families.py        # build container only (needs /root/reference)

Tiny seeded checkpoints: a Qwen2 with q/k/v bias and an odd GQA group (6 q heads on 2 kv heads), and a Mistral
whose head_dim (128) is not hidden/heads (256/4).  Test-only shim: the reference pins Qwen's vocabulary to 151936
(qwen.py:12,27); the tiny model uses 512, so that constant is patched for the run.
Outputs: families.json (engine traces), families_logits.npz.
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg                                               # noqa: E402

from umbrella_amd.models.config import LlamaCfg                        # noqa: E402
from umbrella_amd.models.synthetic import synth_state_small            # noqa: E402

QWEN = dict(vocab_size=512, hidden_size=384, intermediate_size=640, num_hidden_layers=4, num_attention_heads=6,
            num_key_value_heads=2, head_dim=64, rms_norm_eps=1e-6, rope_theta=1000000.0, rope_scaling=None,
            tie_word_embeddings=True, attention_bias=True)
MISTRAL = dict(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
               num_key_value_heads=2, head_dim=128, rms_norm_eps=1e-5, rope_theta=1000000.0, rope_scaling=None,
               tie_word_embeddings=False)
SEEDS = {"qwen": 33, "mistral": 44}


def save(kind, cfgd, seed, path, eos):
    from transformers import GenerationConfig, MistralConfig, MistralForCausalLM, Qwen2Config, Qwen2ForCausalLM
    kw = {k: v for k, v in cfgd.items() if k not in ("rope_scaling", "rope_theta", "attention_bias")}
    common = dict(rope_parameters={"rope_type": "default", "rope_theta": cfgd["rope_theta"]},
                  max_position_embeddings=32768, eos_token_id=eos)
    if kind == "qwen":
        m = Qwen2ForCausalLM(Qwen2Config(**kw, **common, use_sliding_window=False))
    else:
        m = MistralForCausalLM(MistralConfig(**kw, **common, sliding_window=None))
    sd = synth_state_small(LlamaCfg(**cfgd), seed)
    if cfgd["tie_word_embeddings"]:
        sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    missing = m.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and not [k for k in missing.missing_keys if "rotary" not in k], missing
    m.save_pretrained(path)
    GenerationConfig(eos_token_id=eos).save_pretrained(path)
    return m


def main():
    mg.install_shims()
    from transformers import MistralConfig, Qwen2Config
    for cc in (Qwen2Config, MistralConfig):
        if not hasattr(cc, "rope_theta"):
            cc.rope_theta = property(lambda self: (self.rope_parameters or {}).get("rope_theta", 10000.0))
    mg.patch_reference()
    import umbrella.models.qwen as rq
    import umbrella.models.mistral as rm
    from umbrella.models.auto_model import AutoModelLM
    rq.QWEN_2_5_VOCAB_SIZE = 512
    rq.QwenCudagraph.initialize_cuda_graph = lambda self, lens, n_warmups=12: self.clear()
    rm.MistralCudagraph.initialize_cuda_graph = lambda self, lens, n_warmups=12: self.clear()
    tmp = tempfile.mkdtemp(prefix="umb_golden_fam_")
    eos = [3, 5]
    out = {"cfg": {"qwen": QWEN, "mistral": MISTRAL}, "seeds": SEEDS, "cases": {}}
    npz = {}
    rs = np.random.RandomState(7)
    prompt = rs.randint(6, 512, size=20).tolist()
    g34 = os.path.join(tmp, "g34.json")
    import umbrella.sequoia_utils as squ
    squ.generate_sequoia_tree(3, 4, json_file=g34)
    for kind, cfgd, classes in (("qwen", QWEN, (rq.Qwen, rq.QwenOffload, rq.QwenCudagraph)),
                                ("mistral", MISTRAL, (rm.Mistral, rm.MistralOffload, rm.MistralCudagraph))):
        path = os.path.join(tmp, "tiny-" + kind)
        hf = save(kind, cfgd, SEEDS[kind], path, eos)
        AutoModelLM._MODEL_MAPPING[path], AutoModelLM._OFFLOAD_MODEL_MAPPING[path], AutoModelLM._CUDAGRAPH_MODEL_MAPPING[path] = classes
        m = classes[0](path, max_length=128, device="cpu", dtype=torch.float32)
        m.alloc()
        ids = torch.tensor([prompt])
        T = ids.shape[1]
        mask = torch.tril(torch.ones(T, 128, dtype=torch.bool))
        logits = m.inference(ids, torch.arange(T)[None], mask, torch.arange(T))
        with torch.no_grad():
            hf_logits = hf(ids).logits
        out[kind + "_hf_vs_ref_max_abs"] = float((logits - hf_logits).abs().max())
        npz[kind + "_logits_last"] = logits[0, -1].numpy()
        npz[kind + "_logits_rows"] = logits[0, ::5, :64].numpy()
        npz[kind + "_inv_freq"] = hf.model.rotary_emb.inv_freq.numpy()
        base = dict(model=path, max_length=256, safe_buffer=16, stop_distance=8, temperature=0.0)
        # The reference's MistralCudagraph splits its packed qkv by hidden_size (mistral.py:~470) and cannot run a
        # head_dim != hidden/heads model, so the Mistral target is drafted by the tiny Qwen (same vocabulary).
        draft = path if kind == "qwen" else os.path.join(tmp, "tiny-qwen")
        tag = "selfdraft" if kind == "qwen" else "qwendraft"
        out["cases"][f"{kind}_static_3x4_{tag}"] = mg.run_engine(
            "static", dict(base, engine="static", draft_model=draft, growmap_path=g34), prompt, 40)
        out["cases"][f"{kind}_dynamic_w4b6d3_{tag}"] = mg.run_engine(
            "dynamic", dict(base, engine="dynamic", draft_model=draft, width=4, num_beams=6, depth=3), prompt, 32)
        with torch.no_grad():
            ar = hf.generate(ids, max_new_tokens=48, do_sample=False, eos_token_id=None, pad_token_id=0)
        out[kind + "_hf_greedy"] = ar[0, len(prompt):].tolist()
    out["prompt"] = prompt
    np.savez_compressed(os.path.join(mg.OUT, "families_logits.npz"), **npz)
    json.dump(out, open(os.path.join(mg.OUT, "families.json"), "w"))
    for k in ("qwen", "mistral"):
        print(k, "hf_vs_ref_max_abs", out[k + "_hf_vs_ref_max_abs"])
    for k, v in out["cases"].items():
        kind = k.split("_")[0]
        toks = v["turns"][0]["tokens"]
        print(k, "acc", round(v["turns"][0]["avg_accept"], 2), "iters", len(v["iters"]),
              "== HF greedy:", toks == out[kind + "_hf_greedy"][:len(toks)])


if __name__ == "__main__":
    main()
