"""Layer-sharded speculative decoding: the target's layers spread over the GPUs of one node, activations hopping
over RCCL send/recv (BASELINE config 5; capacity, not speed -- SURVEY 8e).  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 \
        examples/spec_generate_pp.py --configuration configs/static_70b_awq_on_device.yaml
"""
import argparse
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umbrella_amd.parallel import build_pipelined_engine, shutdown_pipeline  # noqa: E402
from umbrella_amd.utils import load_config  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--configuration", default="configs/static_70b_awq_on_device.yaml")
ap.add_argument("--prompt-len", type=int, default=128)
args = ap.parse_args()
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
config = load_config(args.configuration)
gen_len = config.pop("generation_length", 256)
config.pop("max_turns", None), config.pop("template", None)
dtype = torch.float16 if "awq" in config["model"].lower() else torch.bfloat16
engine = build_pipelined_engine(f"cuda:{local}", dtype=dtype, **config)
if engine is not None:                                    # rank 0 drives; the other ranks served inside the call
    g = torch.Generator().manual_seed(0)
    assert engine._prefill(torch.randint(3, 128000, (1, args.prompt_len), generator=g))
    print(engine.speculative_decoding(max_new_tokens=gen_len))
    shutdown_pipeline(engine)
dist.barrier()
dist.destroy_process_group()
