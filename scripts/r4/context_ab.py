"""Headline engine at several prompt lengths (bench.context_sweep) -- run twice with UMB_ATTN_KBK=2048 / unset for the
key-span A/B of the narrow-launch tree attention."""
import os, sys, json, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

args = argparse.Namespace(max_length=int(os.environ.get("MAXLEN", 2048)), seed=0)
wl = dict(bench.WORKLOADS[os.environ.get("WL", "70b-awq+1b")])
dtype = torch.float16 if wl["dtype"] == "fp16" else torch.bfloat16
eng, gm, acc = bench.build_engine(wl, "cuda:0", dtype, args.max_length, 0)
print(os.environ.get("UMB_ATTN_KBK", "default"), json.dumps(bench.context_sweep(eng, wl, gm, acc, args)), flush=True)
