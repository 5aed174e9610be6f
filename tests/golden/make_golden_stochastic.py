#!/usr/bin/env python
"""Golden trace of the REFERENCE static engine on its stochastic verification path
(static_speculation_engine.py:131, 298-310), recorded on CPU in the build container:

    python tests/golden/make_golden_stochastic.py        ->  tests/golden/engines_stochastic.json

What this pins: the engine logic AROUND the sampler -- repetition penalty over tokens[:num_nodes + 1] first, then
logits / temperature, then ONE call of flashinfer.sampling.top_k_top_p_sampling_from_logits with the SAME
`uniform_samples = rand(3, tree_size)` tensor at every verify, then the accept scan on the sampled ids.
What it cannot pin: the sampler's internals.  The flashinfer wheel is absent (and unpinned by the reference), so the
third-party symbol is stood in for by the oracle's restatement of its published algorithm (oracle/ops.py) -- the same
kind of stand-in make_golden.py uses for flashinfer.rmsnorm / single_prefill_with_kv_cache.  The recorded
`uniform_samples`, per-iteration sampled ids and accept results are data only.
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import make_golden as mg          # noqa: E402
from oracle import ops as oops    # noqa: E402


def main():
    mg.install_shims()
    log = []

    def sampler(logits, uniform_samples, top_k, top_p, *a, **k):
        ids, ok = oops.top_k_top_p_sampling_from_logits(logits, uniform_samples, top_k, top_p)
        log.append({"sampled": ids.tolist(), "success": ok.tolist()})
        return ids, ok
    sys.modules["flashinfer.sampling"].top_k_top_p_sampling_from_logits = sampler
    mg.patch_reference()
    tmp = tempfile.mkdtemp(prefix="umb_golden_st_")
    tdir, ddir = os.path.join(tmp, "tiny-target"), os.path.join(tmp, "tiny-draft")
    eos = [3, 5]
    mg.save_hf(mg.TARGET, mg.SEEDS["target"], tdir, eos)
    mg.save_hf(mg.DRAFT, mg.SEEDS["draft"], ddir, eos)
    mg.register(tdir); mg.register(ddir)
    import umbrella.sequoia_utils as squ
    g34 = os.path.join(tmp, "g34.json")
    squ.generate_sequoia_tree(3, 4, json_file=g34)
    rs = np.random.RandomState(0)
    prompt = rs.randint(6, 512, size=24).tolist()

    from umbrella.speculation.auto_engine import AutoEngine
    out = {"cases": {}}
    for name, draft, gen in (("static_3x4_stochastic", ddir, dict(temperature=0.6, topp=0.9, topk=32, repetition_penalty=1.05)),
                             ("static_3x4_selfdraft_stochastic", tdir, dict(temperature=0.8, topp=0.95, topk=16, repetition_penalty=1.0))):
        cfg = dict(model=tdir, draft_model=draft, engine="static", growmap_path=g34, max_length=256, safe_buffer=16,
                   stop_distance=8, **gen)
        torch.manual_seed(7)
        eng = AutoEngine.from_config("cpu", **cfg)
        eng.dtype = torch.float32
        eng.initialize()
        eng.eos_tokens = list(eos)
        rec = {"config": {k: v for k, v in cfg.items() if k not in ("model", "draft_model", "growmap_path")},
               "prompt": prompt, "max_new_tokens": 40, "eos": eos, "uniform_samples": eng.uniform_samples.tolist(),
               "iters": []}
        assert eng._prefill(torch.tensor([prompt]))
        rec["first_token"] = int(eng.tokens[0, eng.num_nodes])
        start, go = eng.num_nodes, True
        del log[:]
        while go and (eng.num_nodes - start) < rec["max_new_tokens"] and eng.validate_status():
            n = eng.num_nodes
            eng.build_tree()
            tree = eng.tokens[0, n:n + eng.tree_size].tolist()
            go = eng.verify()
            s = log[-1]
            rec["iters"].append({"n": n, "tree_tokens": tree, "sampled": s["sampled"], "success": s["success"],
                                 "num_nodes": eng.num_nodes, "go_on": go, "bonus": int(eng.tokens[0, eng.num_nodes])})
        rec["tokens"] = eng.tokens[0, start:eng.num_nodes + 1].tolist()
        out["cases"][name] = rec
        acc = (eng.num_nodes - start + 1) / max(len(rec["iters"]), 1)
        print(name, "iters", len(rec["iters"]), "avg accept", round(acc, 2),
              "unaccepted draws", sum(1 for it in rec["iters"] for ok in it["success"] if not ok))
    with open(os.path.join(HERE, "engines_stochastic.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
