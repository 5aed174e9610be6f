"""Shared test helpers: tiny seeded models + oracle engines (tests only)."""
import json
import os

import numpy as np
import torch

from oracle import ops as _oracle_ops
from oracle.engine import OracleDynamicEngine, OracleStaticEngine
from oracle.model import OracleLlama
from umbrella_amd.models.config import LlamaCfg
from umbrella_amd.models.synthetic import synth_state_small

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_golden():
    with open(os.path.join(GOLD, "engines.json")) as f:
        return json.load(f)


def tiny_cfg(d: dict) -> LlamaCfg:
    return LlamaCfg(**d)


def oracle_model(cfgd, seed, max_length, dtype=torch.float32, slot_cache=False, exit_layer=-1, state=None):
    cfg = tiny_cfg(cfgd)
    sd = state if state is not None else synth_state_small(cfg, seed)
    # the oracle's OWN frequency table (oracle/ops.py rope_inv_freq), not the product's: see its docstring
    inv, scale = _oracle_ops.rope_inv_freq(cfg.head_dim, cfg.rope_theta, cfg.rope_scaling)
    return OracleLlama(cfg, sd, inv, scale, max_length=max_length, dtype=dtype, slot_cache=slot_cache,
                       exit_layer=exit_layer)


def oracle_engine_for_case(g, case_name, dtype=torch.float32):
    """Rebuild the engine a golden case was recorded with."""
    case = g["cases"][case_name]
    c = case["config"]
    self_draft = "selfdraft" in case_name or "exit2" in case_name
    dcfg, dseed = (g["target_cfg"], g["seeds"]["target"]) if self_draft else (g["draft_cfg"], g["seeds"]["draft"])
    L = c["max_length"]
    target = oracle_model(g["target_cfg"], g["seeds"]["target"], L, dtype)
    if c["engine"] == "static":
        draft = oracle_model(dcfg, dseed, L, dtype, slot_cache=True, exit_layer=c.get("exit_layer", -1))
        with open(os.path.join(GOLD, "growmaps.json")) as f:
            gms = json.load(f)
        gm = gms["5x6"] if "5x6" in case_name else gms["3x4"]
        return OracleStaticEngine(draft, target, gm, case["eos"], max_length=L, safe_buffer=c["safe_buffer"])
    draft = oracle_model(dcfg, dseed, L, dtype)
    return OracleDynamicEngine(draft, target, case["eos"], width=c["width"], depth=c["depth"],
                               num_beams=c["num_beams"], max_length=L, safe_buffer=c["safe_buffer"])


def replay_case(case, eng, case_name=""):
    """Drive an oracle engine through a recorded reference trace, asserting every per-iteration snapshot."""
    ok = eng._prefill(torch.tensor([case["prompt"]]))
    assert ok == case["prefill_ok"]
    if not ok:
        return
    assert int(eng.tokens[0, eng.num_nodes]) == case["first_token"]
    it_iter = iter(case["iters"])

    def loop(turn):
        start, go, steps = eng.num_nodes, True, 0
        while go and (eng.num_nodes - start) < case["max_new_tokens"] and eng.validate_status():
            n = eng.num_nodes
            eng.build_tree()
            rec = next(it_iter)
            assert rec["n"] == n
            assert eng.tokens[0, n:n + eng.tree_size].tolist() == rec["tree_tokens"], (case_name, steps)
            assert eng.parents.tolist() == rec["parents"]
            if "tree_score" in rec:
                np.testing.assert_allclose(eng.tree_score.numpy(), np.array(rec["tree_score"]), rtol=1e-4, atol=1e-5)
                cur = eng.cur
                assert eng.mask_iter[n:cur, n:cur].sum(-1).tolist() == rec["tree_mask_rowsum"]
            go = eng.verify()
            assert eng.num_nodes == rec["num_nodes"] and go == rec["go_on"]
            assert int(eng.tokens[0, eng.num_nodes]) == rec["bonus"]
            assert eng.target_model.kv_cache.kv_offset == rec["target_kv"]
            assert eng.draft_model.kv_cache.kv_offset == rec["draft_kv"]
            steps += 1
        assert eng.tokens[0, start:eng.num_nodes + 1].tolist() == turn["tokens"]
        assert steps == turn["steps"]

    loop(case["turns"][0])
    if "append" in case:
        assert eng._append(torch.tensor([case["append"]])) == case["append_ok"]
        assert int(eng.tokens[0, eng.num_nodes]) == case["append_first_token"]
        loop(case["turns"][1])
    eng.reset()
    toks, acc = eng.generate_ids(case["prompt"], case["max_new_tokens"])
    assert toks == case["generate"]["generated_tokens"]
    assert abs(acc - case["generate"]["avg_accept_tokens"]) < 1e-9
