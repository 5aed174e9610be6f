"""Diagnostic: eager 1B-shaped forward under the low-latency schedule with per-launch synchronisation."""
import os, sys, copy
os.environ.setdefault("UMBRELLA_SYNTHETIC", "1")
os.environ["UMB_DEBUG_SYNC"] = "1"
os.environ["UMB_SCHED"] = "ll"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from umbrella_amd.models.config import KNOWN
from umbrella_amd.models.llama import Llama
name = "meta-llama/Llama-3.2-1B-Instruct"
cfg = copy.copy(KNOWN[name]); cfg.num_hidden_layers = 2
m = Llama(name, max_length=2048, device="cuda:0", dtype=torch.float16, config=cfg)
m.alloc()
ids = torch.randint(3, 128000, (128 + 3,), dtype=torch.int32, device="cuda:0")
print("prefill", flush=True)
m.prefill_tokens(ids[:128], 0)
torch.cuda.synchronize()
print("decode", flush=True)
pos = torch.arange(128, 131, dtype=torch.int32, device="cuda:0")
pre = torch.tensor([128], dtype=torch.int32, device="cuda:0")
m.forward_explicit(ids[128:].contiguous(), pos, pos, pre, head_from=0)
torch.cuda.synchronize()
print("ok", float(m.logits_buffer[:3].abs().max()))
