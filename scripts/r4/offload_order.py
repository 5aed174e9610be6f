"""C3 offload, 40 resident layers: per-step time alone in a fresh process vs after the all-streamed configuration (bench.py's order)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
os.environ.setdefault("UMBRELLA_SYNTHETIC", "1")
import bench
from umbrella_amd.speculation.auto_engine import AutoEngine

def run(ncl):
    eng = AutoEngine.from_config("cuda:0", engine="dynamic", model=bench.T70, draft_model=bench.D1B, dtype=torch.float16, width=16,
                                 num_beams=24, depth=16, max_length=4096, offload=True, num_cache_layers=ncl, seed=0)
    eng.initialize()
    g = torch.Generator().manual_seed(1)
    prompt = torch.randint(3, 128000, (1, 128), generator=g)
    assert eng._prefill(prompt)
    for _ in range(2): eng.step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        t0 = time.time(); eng.step(); torch.cuda.synchronize(); ts.append(round((time.time() - t0) * 1e3, 1))
    del eng
    import gc; gc.collect(); torch.cuda.empty_cache()
    return ts

order = [int(v) for v in sys.argv[1:]] or [40]
print(json.dumps({str(n): run(n) for n in order}), flush=True)
