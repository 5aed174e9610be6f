#!/usr/bin/env python
"""Generate golden vectors by running the REFERENCE itself on CPU.

Run in the build container only (needs /root/reference, which never travels):

    python tests/golden/make_golden.py

The reference has no CPU path and imports three wheels that are not installed
(flashinfer, awq, awq_ext).  This script injects closed-form torch-CPU stand-ins
for those *third-party* symbols, neutralises the CUDA-only plumbing
(streams/graphs/synchronize), saves tiny seeded HF Llama checkpoints, registers
them in the reference's model registry, and then drives the reference's OWN
classes: StaticSpeculationEngine, DynamicSpeculationEngine, Llama.inference,
KV_Cache / StaticKV_Cache, apply_rotary_pos_emb, speculation_utils helpers and
generate_sequoia_tree.  Outputs (data only) land in tests/golden/*.json|npz.
"""
import json
import math
import os
import sys
import tempfile
import types

import numpy as np
import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from umbrella_amd.models.config import LlamaCfg, LLAMA3_ROPE          # noqa: E402
from umbrella_amd.models.synthetic import synth_state_small            # noqa: E402

# ----------------------------------------------------------------------------- third-party stand-ins


def _rmsnorm(x, w, eps):
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * w.float()).to(x.dtype)


def _single_prefill(q, k, v, kv_layout="NHD", custom_mask=None, allow_fp16_qk_reduction=True, logits_soft_cap=0):
    T, Hq, D = q.shape
    Hkv = k.shape[1]
    g = Hq // Hkv
    kk = k.repeat_interleave(g, dim=1).permute(1, 0, 2)
    vv = v.repeat_interleave(g, dim=1).permute(1, 0, 2)
    w = torch.matmul(q.permute(1, 0, 2), kk.transpose(1, 2)) / math.sqrt(D)
    w = w.masked_fill(~custom_mask[None], torch.finfo(w.dtype).min)
    p = torch.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    return torch.matmul(p, vv).permute(1, 0, 2).contiguous()


def install_shims():
    fi = types.ModuleType("flashinfer")
    fi.rmsnorm = _rmsnorm
    fi.gemma_rmsnorm = lambda x, w, eps: _rmsnorm(x, 1.0 + w, eps)
    fi.single_prefill_with_kv_cache = _single_prefill
    fi.sampling = types.ModuleType("flashinfer.sampling")
    sys.modules["flashinfer"] = fi
    sys.modules["flashinfer.sampling"] = fi.sampling
    awq = types.ModuleType("awq"); awq.modules = types.ModuleType("awq.modules")
    awq.modules.linear = types.ModuleType("awq.modules.linear")
    awq.modules.linear.WQLinear_GEMM = type("WQLinear_GEMM", (), {})
    for n, m in (("awq", awq), ("awq.modules", awq.modules), ("awq.modules.linear", awq.modules.linear)):
        sys.modules[n] = m
    sys.modules["awq_ext"] = types.ModuleType("awq_ext")
    import importlib.machinery
    import importlib.util
    for name in ("matplotlib", "matplotlib.pyplot", "networkx"):       # imported but unused (sequoia_utils.py:4-5)
        if name in sys.modules or (("." not in name) and importlib.util.find_spec(name) is not None):
            continue
        mod = types.ModuleType(name)
        mod.__spec__ = importlib.machinery.ModuleSpec(name, None)
        sys.modules[name] = mod
    # CUDA-only plumbing -> no-ops
    torch.cuda.synchronize = lambda *a, **k: None
    # transformers>=5 moved rope_theta into rope_parameters (read at llama.py:32)
    from transformers import LlamaConfig
    if not hasattr(LlamaConfig, "rope_theta"):
        LlamaConfig.rope_theta = property(lambda self: (self.rope_parameters or {}).get("rope_theta", 10000.0))
    sys.path.insert(0, REF)


class DummyTok:
    def decode(self, ids, **kw):
        return " ".join(str(i) for i in ids)

    def encode(self, text, return_tensors=None):
        return torch.tensor([[int(t) for t in text.split()]])


# ----------------------------------------------------------------------------- tiny checkpoints

TARGET = dict(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=4,
              num_attention_heads=4, num_key_value_heads=2, head_dim=64, rms_norm_eps=1e-5,
              rope_theta=500000.0, rope_scaling=LLAMA3_ROPE, tie_word_embeddings=False)
DRAFT = dict(vocab_size=512, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
             num_attention_heads=2, num_key_value_heads=1, head_dim=64, rms_norm_eps=1e-5,
             rope_theta=500000.0, rope_scaling=LLAMA3_ROPE, tie_word_embeddings=True)
SEEDS = {"target": 11, "draft": 22}


def save_hf(cfgd, seed, path, eos):
    from transformers import GenerationConfig, LlamaConfig, LlamaForCausalLM
    rp = dict(cfgd["rope_scaling"], rope_theta=cfgd["rope_theta"])
    hc = LlamaConfig(**{k: v for k, v in cfgd.items() if k not in ("rope_scaling", "rope_theta")},
                     rope_parameters=rp, max_position_embeddings=131072, eos_token_id=eos)
    m = LlamaForCausalLM(hc)
    sd = synth_state_small(LlamaCfg(**cfgd), seed)
    if cfgd["tie_word_embeddings"]:
        sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    missing = m.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys, missing
    m.save_pretrained(path)
    GenerationConfig(eos_token_id=eos).save_pretrained(path)
    return m


def register(name):
    from umbrella.models.auto_model import AutoModelLM
    from umbrella.models.llama import Llama, LlamaCudagraph, LlamaOffload
    AutoModelLM._MODEL_MAPPING[name] = Llama
    AutoModelLM._OFFLOAD_MODEL_MAPPING[name] = LlamaOffload
    AutoModelLM._CUDAGRAPH_MODEL_MAPPING[name] = LlamaCudagraph


def patch_reference():
    import transformers
    import umbrella.speculation.static_speculation_engine as se
    import umbrella.speculation.dynamic_speculation_engine as de
    from umbrella.models import llama as rl
    from umbrella.speculation.speculation_utils import sampling_argmax_gather
    for mod in (se, de):
        mod.AutoTokenizer = type("AT", (), {"from_pretrained": staticmethod(lambda *a, **k: DummyTok())})
    # CUDA graphs -> eager equivalents the reference itself defines
    se.cuda_graph_for_sampling_argmax_gather = (
        lambda device, idx_len, num_samples, dtype, dim, index_len:
        (lambda logits, idx: sampling_argmax_gather(logits, num_samples, idx)))
    rl.LlamaCudagraph.initialize_cuda_graph = lambda self, lens, n_warmups=12: self.clear()
    # offload classes: streams do not exist on CPU; keep their own layer loop
    torch.cuda.Stream = lambda *a, **k: None

    class _Ctx:
        def __init__(self, *a, **k): pass
        def __enter__(self): return self
        def __exit__(self, *a): return False
    torch.cuda.stream = _Ctx


def run_engine(kind, cfg, prompt, max_new, append=None, eos=None):
    """Drive the reference engine step by step and snapshot its own state."""
    from umbrella.speculation.auto_engine import AutoEngine
    eng = AutoEngine.from_config("cpu", **cfg)
    eng.dtype = torch.float32
    eng.initialize()
    if eos is not None:
        eng.eos_tokens = list(eos)
    rec = {"config": {k: v for k, v in cfg.items() if k not in ("model", "draft_model")}, "prompt": prompt,
           "max_new_tokens": max_new, "eos": list(eng.eos_tokens), "iters": [], "turns": []}

    def loop():
        start, go, steps = eng.num_nodes, True, 0
        while go and (eng.num_nodes - start) < max_new and eng.validate_status():
            n = eng.num_nodes
            eng.build_tree()
            it = {"n": n, "tree_tokens": eng.tokens[0, n:n + eng.tree_size].tolist(),
                  "parents": eng.parents.tolist()}
            if kind == "dynamic":
                it["tree_score"] = eng.tree_score.tolist()
                cur = eng.num_draft_model_tokens_this_iter
                it["tree_mask_rowsum"] = eng.attn_mask_this_iter[n:cur, n:cur].sum(-1).tolist()
            go = eng.verify()
            it.update(num_nodes=eng.num_nodes, accept_length=eng.num_nodes - n, go_on=go,
                      bonus=int(eng.tokens[0, eng.num_nodes]),
                      target_kv=eng.target_model.kv_cache.kv_offset, draft_kv=eng.draft_model.kv_cache.kv_offset)
            rec["iters"].append(it)
            steps += 1
        rec["turns"].append({"start": start, "tokens": eng.tokens[0, start:eng.num_nodes + 1].tolist(),
                             "steps": steps, "avg_accept": (eng.num_nodes - start + 1) / max(steps, 1)})

    ok = eng._prefill(torch.tensor([prompt]))
    rec["prefill_ok"] = bool(ok)
    if ok:
        rec["first_token"] = int(eng.tokens[0, eng.num_nodes])
        loop()
        if append is not None:
            rec["append"] = append
            rec["append_ok"] = bool(eng._append(torch.tensor([append])))
            rec["append_first_token"] = int(eng.tokens[0, eng.num_nodes])
            loop()
    # generate() API on a fresh state
    eng.reset()
    out = eng.generate(input_ids=list(prompt), max_new_tokens=max_new)
    rec["generate"] = {"generated_tokens": out["generated_tokens"], "avg_accept_tokens": out["avg_accept_tokens"]}
    return rec


def main():
    install_shims()
    patch_reference()
    tmp = tempfile.mkdtemp(prefix="umb_golden_")
    tdir, ddir = os.path.join(tmp, "tiny-target"), os.path.join(tmp, "tiny-draft")
    eos = [3, 5]
    hf_t = save_hf(TARGET, SEEDS["target"], tdir, eos)
    save_hf(DRAFT, SEEDS["draft"], ddir, eos)
    register(tdir); register(ddir)
    golden = {"target_cfg": TARGET, "draft_cfg": DRAFT, "seeds": SEEDS, "cases": {}}

    rs = np.random.RandomState(0)
    prompt = rs.randint(6, 512, size=24).tolist()
    prompt2 = rs.randint(6, 512, size=40).tolist()
    append = rs.randint(6, 512, size=7).tolist()

    # ---- model runtime: reference Llama.inference vs HF forward
    from umbrella.models.llama import Llama
    m = Llama(tdir, max_length=128, device="cpu", dtype=torch.float32); m.alloc()
    ids = torch.tensor([prompt])
    T = ids.shape[1]
    mask = torch.tril(torch.ones(T, 128, dtype=torch.bool))
    logits = m.inference(ids, torch.arange(T)[None], mask, torch.arange(T))
    with torch.no_grad():
        hf_logits = hf_t(ids).logits
    golden["hf_vs_ref_max_abs"] = float((logits - hf_logits).abs().max())
    np.savez_compressed(os.path.join(OUT, "model_logits.npz"), prompt=np.array(prompt),
                        logits_last=logits[0, -1].numpy(), logits_rows=logits[0, ::6, :64].numpy(),
                        inv_freq=hf_t.model.rotary_emb.inv_freq.numpy())

    # ---- op-level vectors from the reference's own torch code
    from umbrella.attn.cache import KV_Cache, StaticKV_Cache
    from umbrella.models.model_utils import apply_rotary_pos_emb
    from umbrella.speculation import speculation_utils as su
    g = torch.Generator().manual_seed(5)
    ops = {}
    Tq, Hq, Hkv, D, Lm = 7, 4, 2, 64, 32
    q = torch.randn(1, Tq, Hq, D, generator=g); k = torch.randn(1, Tq, Hkv, D, generator=g)
    pos = torch.tensor([[9, 10, 10, 11, 11, 11, 12]])
    qe, ke = apply_rotary_pos_emb(q, k, m.cos_cache[:Lm], m.sin_cache[:Lm], pos)
    ops.update(rope_q=q[0].numpy(), rope_k=k[0].numpy(), rope_pos=pos[0].numpy(), rope_qe=qe[0].numpy(),
               rope_ke=ke[0].numpy(), rope_cos=m.cos_cache[:Lm].numpy(), rope_sin=m.sin_cache[:Lm].numpy())
    cfg_small = types.SimpleNamespace(num_hidden_layers=1, num_key_value_heads=Hkv, num_attention_heads=Hq,
                                      hidden_size=Hq * D, head_dim=D)
    sk = StaticKV_Cache(cfg_small, max_length=Lm, device="cpu", dtype=torch.float32)
    sk.k_cache.copy_(torch.randn(sk.k_cache.shape, generator=g)); sk.v_cache.copy_(torch.randn(sk.v_cache.shape, generator=g))
    ops["attn_kcache"] = sk.k_cache[0].numpy().copy(); ops["attn_vcache"] = sk.v_cache[0].numpy().copy()
    amask = torch.zeros(Tq, Lm, dtype=torch.bool); amask[:, :9] = True
    tree = torch.tensor([[1,0,0,0,0,0,0],[1,1,0,0,0,0,0],[1,0,1,0,0,0,0],[1,1,0,1,0,0,0],[1,1,0,0,1,0,0],[1,0,1,0,0,1,0],[1,1,0,1,0,0,1]]) == 1
    amask[:, 9:16] = tree
    qh = q.transpose(1, 2).contiguous(); kh = k.transpose(1, 2).contiguous()
    vh = torch.randn(1, Hkv, Tq, D, generator=g)
    sids = torch.arange(9, 16)
    out = sk.compute_attention(qh, kh, vh, 0, sids, amask)
    ops.update(attn_q=qh[0].numpy(), attn_knew=kh[0].numpy(), attn_vnew=vh[0].numpy(), attn_mask=amask.numpy(),
               attn_out=out[0].numpy())
    kc = KV_Cache(cfg_small, max_length=Lm, device="cpu", dtype=torch.float32)
    kc.k_cache.copy_(torch.randn(kc.k_cache.shape, generator=g)); kc.v_cache.copy_(torch.randn(kc.v_cache.shape, generator=g))
    ops["gather_k_before"] = kc.k_cache.numpy().copy(); ops["gather_v_before"] = kc.v_cache.numpy().copy()
    kc.gather_kv_incremental(torch.tensor([9, 11, 14]), 9)
    ops["gather_k_after"] = kc.k_cache.numpy().copy(); ops["gather_v_after"] = kc.v_cache.numpy().copy()
    ops["gather_offset"] = np.array(kc.kv_offset)
    lg = torch.randn(5, 512, generator=g)
    hist = torch.randint(0, 512, (5, 12), generator=g)
    ops.update(rp_logits=lg.numpy(), rp_ids=hist.numpy(),
               rp_out=su.apply_repetition_penalty(hist, lg, 1.05).numpy(),
               topk_out=su.apply_topk(lg, 8).numpy(),
               argmax_gather=su.sampling_argmax_gather(lg[:3], 2, torch.tensor([0, 1, 2, 4])).numpy())
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **ops)

    # ---- Sequoia generator KAT
    import umbrella.sequoia_utils as squ
    gm = squ.generate_sequoia_tree(3, 4, json_file=os.path.join(tmp, "g34.json"))
    shipped = json.load(open(os.path.join(REF, "umbrella/trees/sequoia_tree-3x4.json")))
    golden["sequoia_3x4_equals_shipped"] = (gm == shipped) or (json.loads(json.dumps(gm, default=lambda o: o.tolist())) == shipped)
    gm56 = squ.generate_sequoia_tree(5, 6, acc=[0.5, 0.2, 0.12, 0.08, 0.05, 0.03], json_file=os.path.join(tmp, "g56.json"))
    json.dump({"3x4": shipped, "5x6_acc": [0.5, 0.2, 0.12, 0.08, 0.05, 0.03],
               "5x6": json.loads(json.dumps(gm56, default=lambda o: o.tolist()))},
              open(os.path.join(OUT, "growmaps.json"), "w"))
    g34 = os.path.join(tmp, "g34.json")
    g56 = os.path.join(tmp, "g56.json")

    # ---- engines
    base = dict(model=tdir, max_length=256, safe_buffer=16, stop_distance=8, temperature=0.0)
    C = golden["cases"]
    C["static_3x4"] = run_engine("static", dict(base, engine="static", draft_model=ddir, growmap_path=g34), prompt, 40, append=append)
    C["static_3x4_selfdraft"] = run_engine("static", dict(base, engine="static", draft_model=tdir, growmap_path=g34), prompt2, 48)
    C["static_5x6_selfdraft"] = run_engine("static", dict(base, engine="static", draft_model=tdir, growmap_path=g56), prompt, 48)
    C["static_3x4_exit2"] = run_engine("static", dict(base, engine="static", draft_model=tdir, growmap_path=g34, exit_layer=2), prompt, 32)
    C["dynamic_w4b6d3"] = run_engine("dynamic", dict(base, engine="dynamic", draft_model=ddir, width=4, num_beams=6, depth=3), prompt, 40, append=append)
    C["dynamic_w8b8d4_selfdraft"] = run_engine("dynamic", dict(base, engine="dynamic", draft_model=tdir, width=8, num_beams=8, depth=4), prompt2, 48)
    # EOS inside an accepted path: pick a token the self-draft run emits mid-stream
    toks = C["static_3x4_selfdraft"]["turns"][0]["tokens"]
    eos_tok = toks[9]
    C["static_3x4_selfdraft_eos"] = run_engine("static", dict(base, engine="static", draft_model=tdir, growmap_path=g34), prompt2, 48, eos=[eos_tok])
    C["dynamic_w8b8d4_selfdraft_eos"] = run_engine("dynamic", dict(base, engine="dynamic", draft_model=tdir, width=8, num_beams=8, depth=4), prompt2, 48, eos=[eos_tok])
    # overflow -> False
    C["static_overflow"] = run_engine("static", dict(base, engine="static", draft_model=ddir, growmap_path=g34, max_length=48), prompt, 8)
    # greedy AR of the target (property oracle): HF generate
    with torch.no_grad():
        ar = hf_t.generate(torch.tensor([prompt2]), max_new_tokens=60, do_sample=False, eos_token_id=None, pad_token_id=0)
    golden["hf_greedy_prompt2"] = ar[0, len(prompt2):].tolist()
    with torch.no_grad():
        ar = hf_t.generate(torch.tensor([prompt]), max_new_tokens=60, do_sample=False, eos_token_id=None, pad_token_id=0)
    golden["hf_greedy_prompt"] = ar[0, len(prompt):].tolist()
    json.dump(golden, open(os.path.join(OUT, "engines.json"), "w"))
    print("hf_vs_ref_max_abs", golden["hf_vs_ref_max_abs"], "sequoia KAT", golden["sequoia_3x4_equals_shipped"])
    for k, v in C.items():
        t = v["turns"][0] if v["turns"] else None
        print(k, "prefill_ok", v["prefill_ok"], "acc", None if t is None else round(t["avg_accept"], 2),
              "n_iters", len(v["iters"]))


if __name__ == "__main__":
    main()
