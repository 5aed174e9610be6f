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

def run_prompts(engine, prompts, gen_len, verbose=False):
    """The reference's loop and sums (examples/spec_bench.py:96-124): per prompt prefill -> decode -> [append -> decode]
    -> reset; tokens, seconds and target steps are summed per category and overall.  prompts: [(category, [turn ids])]."""
    total = [0, 0.0, 0]
    per = {}
    for cat, turns in prompts:
        a = per.setdefault(cat, [0, 0.0, 0])
        for t, ids in enumerate(turns):
            ok = engine._prefill(ids) if t == 0 else engine._append(ids)
            if not ok:
                break
            with contextlib.nullcontext() if verbose else contextlib.redirect_stdout(io.StringIO()):
                n, dt, st = engine.speculative_decoding(max_new_tokens=gen_len)
            for acc in (a, total):
                acc[0] += n; acc[1] += dt; acc[2] += st
        engine.reset()
    return per, total


def report(per, total):
    """Avg Accept Tokens = sum(tokens) / sum(target steps), TPOT = 1000 sum(seconds) / sum(tokens)  (spec_bench.py:126-134)"""
    rows = ["{} | Avg Accept Tokens {:.2f} | TPOT {:.2f} ms".format(cat, n / st, 1000 * dt / n)
            for cat, (n, dt, st) in sorted(per.items()) if n and st]
    n, dt, st = total
    rows.append("Summary | Avg Accept Tokens {:.2f} | TPOT {:.2f} ms | {:.1f} tokens/s".format(n / st, 1000 * dt / n, n / dt))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configuration", default="configs/static_8b_code.yaml")
    ap.add_argument("--num-prompts", type=int, default=8)
    ap.add_argument("--verbose", action="store_true", help="stream the decoded ids like the reference does")
    args = ap.parse_args()
    from umbrella_amd.speculation.auto_engine import AutoEngine
    from umbrella_amd.utils import load_config
    config = load_config(args.configuration)
    gen_len = config.pop("generation_length", 256)
    max_turns = config.pop("max_turns", 2)
    config.pop("template", None)
    dtype = torch.float16 if "awq" in config["model"].lower() else torch.bfloat16
    engine = AutoEngine.from_config(device="cuda:0", dtype=dtype, **config)
    engine.initialize()
    lengths = [64, 128, 256, 512]
    g = torch.Generator().manual_seed(0)
    prompts = []
    for idx in range(args.num_prompts):
        P = lengths[idx % len(lengths)]
        turns = [torch.randint(3, 128000, (1, P), generator=g)] + [torch.randint(3, 128000, (1, 32), generator=g)] * (max_turns - 1)
        prompts.append(("prompt {:4d}".format(P), turns))
    for row in report(*run_prompts(engine, prompts, gen_len, args.verbose)):
        print(row)


if __name__ == "__main__":
    main()
