"""Phase timeline of the persistent chain inside a graph-replayed 16-layer 1B forward (T rows), from a -DUMB_CHAIN_TRACE build:
  bash scripts/r5/build_chain_variant.sh trace "-DUMB_CHAIN_TRACE"
  UMB_LIB_PATH=build/variants/lib_chain_trace.so python scripts/r5/chain_trace.py [T] [chain index]
Every wave stamps the 100 MHz wall clock at its phase boundaries; times below are microseconds after the launch's first
stamp, min / median / max over the 256 workgroups."""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
T = int(sys.argv[1]) if len(sys.argv) > 1 else 3
which = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
trace = torch.zeros(64, 256, 4, 32, dtype=torch.int64, device=dev)
os.environ["UMB_CHAIN_TRACE_PTR"] = hex(trace.data_ptr())
os.environ["UMBRELLA_SYNTHETIC"] = "1"
os.environ["UMB_CHAIN"] = "1"
from umbrella_amd.models.config import KNOWN  # noqa: E402
from umbrella_amd.models.llama import Llama  # noqa: E402

name = "meta-llama/Llama-3.2-1B-Instruct"
cfg = copy.copy(KNOWN[name])
m = Llama(name, max_length=2048, device=dev, dtype=torch.float16, config=cfg)
m.alloc()
m.use_gemv(True)
assert m.chain
ids = torch.randint(3, 128000, (128 + T,), dtype=torch.int32, device=dev)
m.prefill_tokens(ids[:128], 0)
step = ids[128:].contiguous()
pos = torch.arange(128, 128 + T, dtype=torch.int32, device=dev)
pre = torch.tensor([128], dtype=torch.int32, device=dev)
run = lambda: m.forward_explicit(step, pos, pos, pre, head_from=0)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    run()                                   # launches 0..16 (eager)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    run()                                   # launches 17..33 (captured)
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
import time
t0 = time.time()
for _ in range(50):
    g.replay()
torch.cuda.synchronize()
print(f"forward T={T}: {(time.time() - t0) / 50 * 1e3:.3f} ms (trace build)")
print("status", hex(m.chain_status()))
tr = trace[17 + which].cpu().numpy().astype("int64")       # [256][4][32]
base = tr[:, :, 0][tr[:, :, 0] > 0].min()
us = lambda x: (x - base) / 100.0


def stat(col, label):
    v = col[col > 0]
    if len(v) == 0:
        return
    v = us(v)
    import numpy as np
    print(f"  {label:34s} min {v.min():7.2f}  med {np.median(v):7.2f}  max {v.max():7.2f}")


print(f"chain launch #{which} of the forward (front + tail), T={T}")
print(" loader (wave 0):")
stat(tr[:, 0, 0], "start")
n_slots = 29
for i in (1, 2, 3, 6, 7, 8, 12, 18, 19, 20, 26, 27, 28, 29):
    stat(tr[:, 0, i], f"slot {i - 1} issued" if i <= n_slots else "slot")
stat(tr[:, 0, 30], "all landed")
import numpy as np
fn, ft = tr[:, 0, 31] >> 48, tr[:, 0, 31] & ((1 << 48) - 1)
print(f"  ring-full episodes per loader: med {np.median(fn):.0f} max {fn.max()}; time spent in them: med "
      f"{np.median(ft) / 100:.2f} us max {ft.max() / 100:.2f} us")
labels = ["start", "o slots done", "h1 gathered", "gate/up operand built", "gate/up slots done", "act gathered",
          "down slots done", "h2 gathered", "q/k/v done (end)"]
for w in (1, 2, 3):
    print(f" consumer {w - 1} (wave {w}):")
    for i, lab in enumerate(labels):
        stat(tr[:, w, i], lab)
