"""Sequoia growmap re-tuned for MI355X cost ratios (reference: examples/construct_sequoia.py:58-90 +
umbrella/sequoia_utils.py:83-130, which size the tree for a 24/48 GB NVIDIA card).

On this part the verify forward is weight-streaming bound and almost flat in the tree size up to T = 64, while a
draft forward costs the same for 1 or 8 rows: the optimal tree is much larger than the shipped 3x4.  This script
  1. measures, under hipGraph replay on the headline pairing (70B-AWQ target + 1B draft, fp16), the verify time at
     T = w d + 1 and the draft forward time per level width;
  2. evaluates tokens/s = E[accept](tree, acc) / (d draft forwards + verify + fixed) for every w x d with T <= 64;
  3. writes the arg-max tree to umbrella_amd/trees/ and the table to gpurun_out/ (copied to profiles/).
Usage: python scripts/tune_growmap.py [--acc 0.65 0.2 0.1 0.05] [--target ... --draft ...]"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("UMBRELLA_SYNTHETIC", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ge.build()
from umbrella_amd.models import AutoModelLM
from umbrella_amd.sequoia_utils import DEFAULT_ACC, expected_accept_length, generate_budget_tree, generate_sequoia_tree

ap = argparse.ArgumentParser()
ap.add_argument("--target", default="hugging-quants/Meta-Llama-3.1-70B-Instruct-AWQ-INT4")
ap.add_argument("--draft", default="meta-llama/Llama-3.2-1B-Instruct")
ap.add_argument("--acc", type=float, nargs="*", default=DEFAULT_ACC)
ap.add_argument("--max-tree", type=int, default=64)
ap.add_argument("--prefix", type=int, default=256)
ap.add_argument("--fixed-ms", type=float, default=0.35, help="top-k levels, accept scan, compaction, host sync per iteration")
ap.add_argument("--name", default="mi355x_70b_awq_1b")
args = ap.parse_args()
dev, dtype = "cuda:0", torch.float16


def graph_ms(model, T, prefix, head=True, reps=30):
    ids = torch.randint(3, 128000, (T,), dtype=torch.int32, device=dev)
    pos = torch.arange(prefix, prefix + T, dtype=torch.int32, device=dev)
    pre = torch.tensor([prefix], dtype=torch.int32, device=dev)
    run = lambda: model.forward_explicit(ids, pos, pos, pre, head_from=0 if head else T)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.time() - t0) / reps * 1e3


target = AutoModelLM.from_pretrained(args.target, max_length=2048, device=dev, dtype=dtype)
target.alloc()
draft = AutoModelLM.from_pretrained(args.draft, max_length=2048, device=dev, dtype=dtype, cuda_graph=True)
draft.alloc(exit_layer=16)
for m in (target, draft):
    m.prefill_tokens(torch.randint(3, 128000, (args.prefix,), dtype=torch.int32, device=dev), 0)
# measured on both sides of every 16-row token-tile boundary: the cost steps there (MFMA work and activation traffic per
# weight byte grow with the tile count), it is flat inside a tile
verify = {T: graph_ms(target, T, args.prefix) for T in (1, 8, 13, 16, 17, 24, 32, 33, 48, 49, 64)}
dforward = {w: graph_ms(draft, w, args.prefix) for w in (1, 2, 3, 4, 5, 6, 7, 8, 12, 16)}
print("verify ms by T:", {k: round(v, 3) for k, v in verify.items()}, flush=True)
print("draft forward ms by rows:", {k: round(v, 3) for k, v in dforward.items()}, flush=True)


def interp(tbl, x):
    ks = sorted(tbl)
    if x <= ks[0]:
        return tbl[ks[0]]
    for a, b in zip(ks, ks[1:]):
        if x <= b:
            return tbl[a] + (tbl[b] - tbl[a]) * (x - a) / (b - a)
    return tbl[ks[-1]]


acc = list(args.acc)
rows = []


def cost(gm):
    widths = [len(x) for x in gm["roots"]]
    # draft forwards per iteration with the look-back schedule: the root (2 rows) + every further level that has children
    t_draft = interp(dforward, 2) + sum(interp(dforward, w) for w in widths[1:-1])
    return t_draft, interp(verify, gm["size"])


for w in range(1, 17):                                   # the reference's generator: fixed width per level
    for d in range(1, 33):
        T = w * d + 1
        if T > args.max_tree:
            continue
        gm = generate_sequoia_tree(w, d, acc + [1e-9] * max(0, w - len(acc)))
        e = expected_accept_length(gm, acc)
        t_draft, t_ver = cost(gm)
        t = t_draft + t_ver + args.fixed_ms
        rows.append(dict(kind="sequoia", w=w, d=d, T=T, accept=round(e, 3), draft_ms=round(t_draft, 3), verify_ms=round(t_ver, 3),
                         iter_ms=round(t, 3), tokens_s=round(e / t * 1e3, 1)))
for T in list(range(6, 17)) + list(range(20, args.max_tree + 1, 4)):                 # node-budget trees: the T - 1 most probable nodes of depth <= d
    for d in range(2, 17):
        gm = generate_budget_tree(T, d, acc)
        if len(gm["roots"]) - 1 < d:
            continue                                      # the cap did not bind: same tree as a smaller d
        e = expected_accept_length(gm, acc)
        t_draft, t_ver = cost(gm)
        t = t_draft + t_ver + args.fixed_ms
        rows.append(dict(kind="budget", w=max(len(x) for x in gm["roots"]), d=d, T=gm["size"], accept=round(e, 3),
                         draft_ms=round(t_draft, 3), verify_ms=round(t_ver, 3), iter_ms=round(t, 3),
                         tokens_s=round(e / t * 1e3, 1)))
rows.sort(key=lambda r: -r["tokens_s"])
best = rows[0]
base = next(r for r in rows if r["kind"] == "sequoia" and r["w"] == 3 and r["d"] == 4)
print("best:", best, "\nshipped 3x4:", base, flush=True)
gm = generate_budget_tree(best["T"], best["d"], acc) if best["kind"] == "budget" else \
    generate_sequoia_tree(best["w"], best["d"], acc + [1e-9] * max(0, best["w"] - len(acc)))
dst = os.path.join(ROOT, "gpurun_out", f"{args.name}-T{best['T']}d{best['d']}.json")
os.makedirs(os.path.dirname(dst), exist_ok=True)
with open(dst, "w") as f:
    json.dump(gm, f)
with open(os.path.join(ROOT, "gpurun_out", f"growmap_tuning_{args.name}.json"), "w") as f:
    json.dump({"target": args.target, "draft": args.draft, "acc": acc, "prefix": args.prefix, "fixed_ms": args.fixed_ms,
               "verify_ms_by_T": verify, "draft_forward_ms_by_rows": dforward, "best": best, "shipped_3x4": base,
               "top20": rows[:20]}, f, indent=1)
print("wrote", dst)
