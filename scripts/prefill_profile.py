"""One 2048-token prompt through the 70B-AWQ target in 1024-token chunks (the engines' prefill path), for rocprofv3:
  rocprofv3 --kernel-trace --output-format csv -d <dir> -- python scripts/prefill_profile.py"""
import os, sys, time
os.environ.setdefault("UMBRELLA_SYNTHETIC", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from umbrella_amd.models import AutoModelLM
name = "hugging-quants/Meta-Llama-3.1-70B-Instruct-AWQ-INT4"
L = int(os.environ.get("LAYERS", "80"))
import copy
from umbrella_amd.models.config import KNOWN
from umbrella_amd.models.llama import Llama
cfg = copy.copy(KNOWN[name]); cfg.num_hidden_layers = L
m = Llama(name, max_length=4096, device="cuda:0", dtype=torch.float16, config=cfg)
m.alloc()
m.reserve(m.PREFILL_CHUNK, logit_rows=64)
P = int(os.environ.get("P", "2048"))
ids = torch.randint(3, 128000, (P,), generator=torch.Generator().manual_seed(0)).int().cuda()
m.clear(); m.prefill_tokens(ids, 0); torch.cuda.synchronize()
for _ in range(int(os.environ.get("REPS", "2"))):
    m.clear(); t0 = time.time(); m.prefill_tokens(ids, 0); torch.cuda.synchronize(); dt = time.time() - t0
    print(f"P={P} L={L}: {dt * 1e3:.1f} ms ({P / dt:.0f} tok/s)")
