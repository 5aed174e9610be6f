"""Shared test helpers: tiny seeded models + oracle engines (tests only)."""
import json
import os

import numpy as np
import torch

from oracle.engine import OracleDynamicEngine, OracleStaticEngine
from oracle.model import OracleLlama
from umbrella_amd.models.config import LlamaCfg, rope_inv_freq
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
    inv, scale = rope_inv_freq(cfg)
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
