"""Multi-prompt speculative-decoding benchmark with the reference's aggregation
(examples/spec_bench.py:79-134: per prompt prefill -> decode -> [append -> decode] -> reset;
Avg Accept Tokens = sum(tokens) / sum(target steps), TPOT = sum(time) / sum(tokens)).
MT-Bench is not available offline: prompts are synthetic token ids with the MT-Bench length profile
(first turn 64..512 tokens incl. system prompt, second turn 32 tokens).

    python examples/spec_bench.py --configuration configs/static_70b_awq_on_device.yaml --num-prompts 8
"""
import argparse
import contextlib
import io
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umbrella_amd.speculation.auto_engine import AutoEngine  # noqa: E402
from umbrella_amd.utils import load_config  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--configuration", default="configs/static_8b_code.yaml")
ap.add_argument("--num-prompts", type=int, default=8)
ap.add_argument("--verbose", action="store_true", help="stream the decoded ids like the reference does")
args = ap.parse_args()
config = load_config(args.configuration)
GEN_LEN = config.pop("generation_length", 256)
MAX_TURNS = config.pop("max_turns", 2)
config.pop("template", None)
dtype = torch.float16 if "awq" in config["model"].lower() else torch.bfloat16
engine = AutoEngine.from_config(device="cuda:0", dtype=dtype, **config)
engine.initialize()

LENGTHS = [64, 128, 256, 512]
g = torch.Generator().manual_seed(0)
steps = time_s = tokens = 0
per_len = {}
for idx in range(args.num_prompts):
    P = LENGTHS[idx % len(LENGTHS)]
    turns = [torch.randint(3, 128000, (1, P), generator=g)] + [torch.randint(3, 128000, (1, 32), generator=g)] * (MAX_TURNS - 1)
    for t, ids in enumerate(turns):
        ok = engine._prefill(ids) if t == 0 else engine._append(ids)
        if not ok:
            break
        with contextlib.nullcontext() if args.verbose else contextlib.redirect_stdout(io.StringIO()):
            n, dt, st = engine.speculative_decoding(max_new_tokens=GEN_LEN)
        tokens += n; time_s += dt; steps += st
        a = per_len.setdefault(P, [0, 0.0, 0])
        a[0] += n; a[1] += dt; a[2] += st
    engine.reset()
for P, (n, dt, st) in sorted(per_len.items()):
    print("prompt {:4d} | Avg Accept Tokens {:.2f} | TPOT {:.2f} ms".format(P, n / st, 1000 * dt / n))
print("Summary | Avg Accept Tokens {:.2f} | TPOT {:.2f} ms | {:.1f} tokens/s".format(tokens / steps, 1000 * time_s / tokens, tokens / time_s))
