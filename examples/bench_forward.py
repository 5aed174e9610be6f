"""Fixed-shape forward timing (counterpart of the reference's examples/bench.py:27-51): prefill P random
tokens, then time R repeats of a D-token decode step of one model."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umbrella_amd.models import AutoModelLM  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="meta-llama/Llama-3.2-1B-Instruct")
ap.add_argument("--P", type=int, default=128)
ap.add_argument("--D", type=int, default=1)
ap.add_argument("--T", type=int, default=100)
ap.add_argument("--M", type=int, default=2048)
args = ap.parse_args()
dtype = torch.float16
m = AutoModelLM.from_pretrained(args.model, max_length=args.M, device="cuda:0", dtype=dtype)
m.alloc()
ids = torch.randint(3, 128000, (args.P + args.D,), dtype=torch.int32, device="cuda:0")
m.prefill_tokens(ids[:args.P], 0)
step = ids[args.P:]
for _ in range(5):
    m.prefill_tokens(step, args.P)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(args.T):
    m.prefill_tokens(step, args.P)
torch.cuda.synchronize()
dt = (time.time() - t0) / args.T
print(f"{args.model}: P={args.P} D={args.D}: {dt*1e3:.3f} ms/forward, weights {m.weight_bytes()/1e9:.2f} GB -> "
      f"{m.weight_bytes()/dt/1e9:.0f} GB/s")
