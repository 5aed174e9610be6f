"""Prefill (time to first token) of the target for a P-token prompt at different chunk sizes."""
import os, sys, time

os.environ.setdefault("UMBRELLA_SYNTHETIC", "1")   # benchmarks run on seeded random weights of the exact shapes (no checkpoints offline)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from umbrella_amd.models import AutoModelLM
name = sys.argv[1] if len(sys.argv) > 1 else "hugging-quants/Meta-Llama-3.1-70B-Instruct-AWQ-INT4"
dtype = torch.float16 if "awq" in name.lower() else torch.bfloat16
m = AutoModelLM.from_pretrained(name, max_length=4096, device="cuda:0", dtype=dtype)
m.alloc()
g = torch.Generator().manual_seed(0)
for P in (128, 512, 2048):
    ids = torch.randint(3, 128000, (P,), generator=g).int().cuda()
    for chunk in (128, 256, 512, 1024):
        m.PREFILL_CHUNK = chunk
        m.reserve(chunk, logit_rows=64)
        m.clear(); m.prefill_tokens(ids, 0); torch.cuda.synchronize()
        m.clear(); t0 = time.time(); row = m.prefill_tokens(ids, 0); torch.cuda.synchronize(); dt = time.time() - t0
        print(f"{name.split('/')[-1]} P={P:5d} chunk={chunk:4d}: {dt * 1e3:8.1f} ms  ({P / dt:8.0f} tok/s)  argmax {int(row.argmax())}")
