"""Two-turn speculative-decoding demo on token ids (counterpart of the reference's
examples/spec_generate.py:26-57: prefill -> decode -> append -> decode -> reset).

    python examples/spec_generate.py --configuration configs/static_70b_awq_on_device.yaml
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umbrella_amd.speculation.auto_engine import AutoEngine  # noqa: E402
from umbrella_amd.utils import load_config  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--configuration", default="configs/static_8b_code.yaml")
ap.add_argument("--prompt-len", type=int, default=128)
args = ap.parse_args()
config = load_config(args.configuration)
GEN_LEN = config.pop("generation_length", 256)
config.pop("max_turns", None), config.pop("template", None)
dtype = torch.float16 if "awq" in config["model"].lower() else torch.bfloat16
engine = AutoEngine.from_config(device="cuda:0", dtype=dtype, **config)
engine.initialize()
g = torch.Generator().manual_seed(0)
ids = torch.randint(3, 128000, (1, args.prompt_len), generator=g)
assert engine._prefill(ids)
print(engine.speculative_decoding(max_new_tokens=GEN_LEN))
assert engine._append(torch.randint(3, 128000, (1, 32), generator=g))
print(engine.speculative_decoding(max_new_tokens=GEN_LEN))
engine.reset()
