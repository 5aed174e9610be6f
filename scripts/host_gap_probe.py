import os, sys, time

os.environ.setdefault("UMBRELLA_SYNTHETIC", "1")   # benchmarks run on seeded random weights of the exact shapes (no checkpoints offline)
sys.path.insert(0, '/root/repo')
import torch
import __graft_entry__ as ge
ge.build()
import bench
wl = bench.WORKLOADS["70b-awq+1b"]
eng, gm, acc = bench.build_engine(wl, "cuda:0", torch.float16, 2048, 0)
g = torch.Generator().manual_seed(1)
prompt = torch.randint(3, 128000, (1, 128), generator=g)
assert eng._prefill(prompt)
for _ in range(4):
    eng.step()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(32):
    eng.step()
torch.cuda.synchronize(); a = (time.time() - t0) / 32 * 1e3
torch.cuda.synchronize(); t0 = time.time()
for _ in range(32):
    eng._graph.replay()
torch.cuda.synchronize(); b = (time.time() - t0) / 32 * 1e3
eng._finish_iteration()
print(f"step() loop {a:.3f} ms/iter ; back-to-back graph replays {b:.3f} ms/iter ; host gap {1e3*(a-b):.0f} us")
