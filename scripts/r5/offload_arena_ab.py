"""C3 offload step time with / without the NUMA-local host arena (one process per leg; run twice, alternated):
  UMB_HOST_ARENA=0|1 python scripts/r5/offload_arena_ab.py [num_cache_layers]
Some host memory is churned first (what the headline + secondary legs leave behind in bench.py)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ.setdefault("UMBRELLA_SYNTHETIC", "1")
from umbrella_amd.speculation.auto_engine import AutoEngine  # noqa: E402

ncl = int(sys.argv[1]) if len(sys.argv) > 1 else 40
T70, D1B = "hugging-quants/Meta-Llama-3.1-70B-Instruct-AWQ-INT4", "meta-llama/Llama-3.2-1B-Instruct"
if os.environ.get("CHURN", "1") == "1":
    junk = [torch.empty(1 << 30, dtype=torch.uint8).pin_memory() for _ in range(12)]        # 12 GiB pinned, then released
    del junk
    torch._C._host_emptyCache() if hasattr(torch._C, "_host_emptyCache") else None
eng = AutoEngine.from_config("cuda:0", engine="dynamic", model=T70, draft_model=D1B, dtype=torch.float16, width=16, num_beams=24,
                             depth=16, max_length=4096, offload=True, num_cache_layers=ncl, seed=0)
eng.initialize()
prompt = torch.randint(3, 128000, (1, 128), generator=torch.Generator().manual_seed(1))
assert eng._prefill(prompt)
for _ in range(2):
    eng.step()
torch.cuda.synchronize()
times = []
for _ in range(4):
    t0 = time.time()
    eng.step()
    torch.cuda.synchronize()
    times.append((time.time() - t0) * 1e3)
m = eng.target_model
streamed = sum(1 for h in m.host_slabs if h is not None) * m.slab_bytes
ms = sorted(times)[len(times) // 2]
print(f"UMB_HOST_ARENA={os.environ.get('UMB_HOST_ARENA', '1')} ncl={ncl}: {ms:.1f} ms per step ({[round(t, 1) for t in times]}), "
      f"{streamed / ms / 1e6:.1f} GB/s = {streamed / ms / 1e6 / 63:.3f} of the link; host: {m.host_placement()}", flush=True)
