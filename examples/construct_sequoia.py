"""Measure the draft's acceptance vector against the target and grow a Sequoia tree for it
(counterpart of the reference's examples/construct_sequoia.py; same flags).  Offline there is no HumanEval:
the sequences are synthetic token ids (context + "solution"), so with random-init weights the vector is
near zero -- pass --acc to grow a tree for a known vector instead.

    python examples/construct_sequoia.py --model meta-llama/Llama-3.1-8B-Instruct \
        --draft_model meta-llama/Llama-3.2-1B-Instruct --w 5 --d 6 --dst /tmp/sequoia_5x6.json
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umbrella_amd.models import AutoModelLM  # noqa: E402
from umbrella_amd.sequoia_utils import expected_accept_length, generate_sequoia_tree, measure_acceptance_rate  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="meta-llama/Llama-3.1-8B-Instruct")
ap.add_argument("--draft_model", default="meta-llama/Llama-3.2-1B-Instruct")
ap.add_argument("--w", type=int, default=3, help="tree width")
ap.add_argument("--d", type=int, default=4, help="tree depth")
ap.add_argument("--dst", default="sequoia_tree.json")
ap.add_argument("--num-seqs", type=int, default=8)
ap.add_argument("--context-len", type=int, default=192)
ap.add_argument("--solution-len", type=int, default=64)
ap.add_argument("--acc", type=float, nargs="*", help="skip the measurement and use this acceptance vector")
args = ap.parse_args()

if args.acc:
    acc = list(args.acc)
else:
    DEVICE, MAX_LEN = "cuda:0", 2048
    dtype = torch.float16 if "awq" in args.model.lower() else torch.bfloat16
    draft = AutoModelLM.from_pretrained(model_name=args.draft_model, offload=False, cuda_graph=True, batch_size=1,
                                        max_length=MAX_LEN, dtype=dtype, device=DEVICE)
    draft.alloc()
    target = AutoModelLM.from_pretrained(model_name=args.model, offload=False, cuda_graph=False, batch_size=1,
                                         max_length=MAX_LEN, dtype=dtype, device=DEVICE)
    target.alloc()
    g = torch.Generator().manual_seed(0)
    total, counts = 0, torch.zeros(args.w, device=DEVICE)
    for _ in range(args.num_seqs):
        ids = torch.randint(3, 128000, (1, args.context_len + args.solution_len), generator=g)
        c, n = measure_acceptance_rate(draft, target, ids, args.solution_len, args.w)
        counts += c
        total += n
    acc = (counts / total).tolist()
    print("acceptance vector:", [round(a, 4) for a in acc], "over", total, "positions")
    acc = [max(a, 1e-6) for a in acc]                  # log() of an unseen rank
gm = generate_sequoia_tree(width=args.w, depth=args.d, acc=acc, json_file=args.dst)
print(json.dumps({"dst": args.dst, "size": gm["size"], "branches": gm["branches"],
                  "expected_accept_len": round(expected_accept_length(gm, acc), 3)}))
