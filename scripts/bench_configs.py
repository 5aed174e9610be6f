"""Secondary BASELINE configs (not the headline bench line): tokens/s + link / HBM rates.
  C2: Llama-3.1-8B target + 1B draft, bf16, static 5x6, on-device
  C3: Llama-3.1-70B-AWQ target (layers streamed from pinned host DRAM) + 1B draft, dynamic w16/b24/d16
"""
import argparse, json, os, sys, time
os.environ.setdefault("UMBRELLA_SYNTHETIC", "1")   # benchmarks run on seeded random weights of the exact shapes (no checkpoints offline)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from umbrella_amd.speculation.auto_engine import AutoEngine
from umbrella_amd.sequoia_utils import generate_sequoia_tree

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2", choices=["c2", "c3", "c3-resident", "c4"])
ap.add_argument("--steps", type=int, default=16)
ap.add_argument("--cache-layers", type=int, default=0)
a = ap.parse_args()
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
prompt = torch.randint(3, 128000, (1, 128), generator=g)
if a.config == "c2":
    eng = AutoEngine.from_config(dev, engine="static", model="meta-llama/Llama-3.1-8B-Instruct",
                                 draft_model="meta-llama/Llama-3.2-1B-Instruct", dtype=torch.bfloat16,
                                 growmap=generate_sequoia_tree(5, 6, [0.5, 0.2, 0.12, 0.08, 0.05, 0.03]), max_length=2048, exit_layer=16)
elif a.config in ("c3", "c3-resident"):
    eng = AutoEngine.from_config(dev, engine="dynamic", model="hugging-quants/Meta-Llama-3.1-70B-Instruct-AWQ-INT4",
                                 draft_model="meta-llama/Llama-3.2-1B-Instruct", dtype=torch.float16, width=16, num_beams=24,
                                 depth=16, max_length=4096, offload=(a.config == "c3"), num_cache_layers=a.cache_layers)
else:
    eng = AutoEngine.from_config(dev, engine="dynamic", model="hugging-quants/Meta-Llama-3.1-70B-Instruct-AWQ-INT4",
                                 draft_model="hugging-quants/Meta-Llama-3.1-8B-Instruct-AWQ-INT4", dtype=torch.float16, width=32,
                                 num_beams=32, depth=24, max_length=4096, offload=False, temperature=0.6, topp=0.9, topk=32,
                                 repetition_penalty=1.05)
t0 = time.time(); eng.initialize(); torch.cuda.synchronize(); t_init = time.time() - t0
assert eng._prefill(prompt)
for _ in range(2):
    eng.step()
torch.cuda.synchronize(); start = eng.num_nodes; t0 = time.time()
for _ in range(a.steps):
    eng.step()
torch.cuda.synchronize(); dt = time.time() - t0
m = eng.target_model
out = {"config": a.config, "tree_size": eng.tree_size, "steps": a.steps, "ms_per_step": round(dt / a.steps * 1e3, 3),
       "tokens_per_s_raw_draft": round((eng.num_nodes - start) / dt, 2), "accept_len_raw_draft": round((eng.num_nodes - start) / a.steps, 3),
       "init_s": round(t_init, 1), "target_weight_GB": round(m.weight_bytes() / 1e9, 2)}
if getattr(m, "_off", None) is not None:
    streamed = sum(1 for h in m.host_slabs if h is not None) * m.slab_bytes
    out["host_link_GBs"] = round(streamed * a.steps / dt / 1e9, 1)
    out["streamed_GB_per_verify"] = round(streamed / 1e9, 2)
    out["cross_forward_prefetch"] = os.environ.get("UMB_OFFLOAD_PREFETCH", "1") != "0"
    out["draft_in_graph"] = eng.graph_scope == "draft" and eng.use_graph
    # the floor: the same slabs copied back to back on the side stream with nothing else running (pure link time)
    torch.cuda.synchronize()
    hs = [h for h in m.host_slabs if h is not None]
    with torch.cuda.stream(m.load_stream):
        for i, h in enumerate(hs[:4]):
            m._dev_slabs[i % len(m._dev_slabs)].copy_(h, non_blocking=True)
    torch.cuda.synchronize(); t0 = time.time()
    with torch.cuda.stream(m.load_stream):
        for i, h in enumerate(hs):
            m._dev_slabs[i % len(m._dev_slabs)].copy_(h, non_blocking=True)
    torch.cuda.synchronize(); t_stream = time.time() - t0
    out["pure_stream_ms_per_verify"] = round(t_stream * 1e3, 2)
    out["pure_stream_GBs"] = round(streamed / t_stream / 1e9, 1)
    out["iter_over_stream"] = round(dt / a.steps / t_stream, 4)
    for i in range(len(m._pf_state)):
        m._pf_state[i] = -1                     # the slabs were overwritten: the next forward must refetch
print(json.dumps(out))
