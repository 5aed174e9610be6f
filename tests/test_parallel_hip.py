"""Layer-sharded pipeline with REAL HIP stages: two processes share one GPU, gloo carries the hops (through pinned host
buffers -- RCCL needs one GPU per rank, which the test box does not have).  Token ids must equal the single-process
engine's: same kernels, same order, only the transport differs -- for the static greedy engine (BASELINE config 5) and
for the reference's 70B engine: dynamic beam-grown tree with stochastic verification, sampled on the last stage."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu
NAME = "meta-llama/Llama-3.2-1B-Instruct"
PROMPT = [11, 250, 7, 1999, 42, 9001, 345, 77, 5, 12000, 64, 3, 901, 15, 33000, 8]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _pp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), UMBRELLA_SYNTHETIC="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as ge
    ge.build()
    from umbrella_amd.parallel import build_pipelined_engine, shutdown_pipeline
    from umbrella_amd.sequoia_utils import generate_sequoia_tree
    from umbrella_amd.speculation.speculation_utils import IdTokenizer
    eng = build_pipelined_engine("cuda:0", dtype=torch.float16, engine="static", model=NAME, draft_model=NAME,
                                 growmap=generate_sequoia_tree(3, 4), max_length=512, exit_layer=4, safe_buffer=16,
                                 tokenizer=IdTokenizer())
    if eng is not None:
        long_prompt = (PROMPT * 5)[:70]                               # 70 rows: one prompt message, walked as chunk(s)
        a = eng.generate(input_ids=PROMPT, max_new_tokens=20)["generated_tokens"]
        b = eng.generate(input_ids=long_prompt, max_new_tokens=12)["generated_tokens"]     # second request: reset + new plan
        shutdown_pipeline(eng)
        q.put((a, b, eng._stage_model.num_layers))
    dist.barrier()
    dist.destroy_process_group()


def test_two_stage_pipeline_equals_single_process():
    import torch.multiprocessing as mp
    import __graft_entry__ as ge
    ge.build()
    os.environ.setdefault("UMBRELLA_SYNTHETIC", "1")
    from umbrella_amd.sequoia_utils import generate_sequoia_tree
    from umbrella_amd.speculation.speculation_utils import IdTokenizer
    from umbrella_amd.speculation.static_speculation_engine import StaticSpeculationEngine
    ref = StaticSpeculationEngine(NAME, NAME, dtype=torch.float16, device="cuda:0", growmap=generate_sequoia_tree(3, 4),
                                  max_length=512, exit_layer=4, safe_buffer=16, tokenizer=IdTokenizer())
    ref.initialize()
    ra = ref.generate(input_ids=PROMPT, max_new_tokens=20)["generated_tokens"]
    rb = ref.generate(input_ids=(PROMPT * 5)[:70], max_new_tokens=12)["generated_tokens"]
    del ref
    torch.cuda.empty_cache()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        a, b, stage_layers = q.get(timeout=150)
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    assert stage_layers == 8                                          # 16 layers split 8 / 8
    assert a == ra and b == rb


def test_eight_stage_pipeline_equals_single_process():
    """BASELINE config 5's shape -- EIGHT stages -- on the one GPU a test box has: eight processes, two of the 1B's 16
    layers each, seven gloo hops per forward (host staged), per-stage hipGraphs; token ids == the single-process engine's.
    (RCCL send / recv between devices stays unmeasured here: README.)"""
    import torch.multiprocessing as mp
    import __graft_entry__ as ge
    ge.build()
    os.environ.setdefault("UMBRELLA_SYNTHETIC", "1")
    from umbrella_amd.sequoia_utils import generate_sequoia_tree
    from umbrella_amd.speculation.speculation_utils import IdTokenizer
    from umbrella_amd.speculation.static_speculation_engine import StaticSpeculationEngine
    ref = StaticSpeculationEngine(NAME, NAME, dtype=torch.float16, device="cuda:0", growmap=generate_sequoia_tree(3, 4),
                                  max_length=512, exit_layer=4, safe_buffer=16, tokenizer=IdTokenizer())
    ref.initialize()
    ra = ref.generate(input_ids=PROMPT, max_new_tokens=20)["generated_tokens"]
    rb = ref.generate(input_ids=(PROMPT * 5)[:70], max_new_tokens=12)["generated_tokens"]
    del ref
    torch.cuda.empty_cache()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pp_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    try:
        a, b, stage_layers = q.get(timeout=400)
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    assert stage_layers == 2                                          # 16 layers, 2 per stage
    assert a == ra and b == rb


DYN = dict(width=8, num_beams=8, depth=4, temperature=0.7, topp=0.9, topk=16, repetition_penalty=1.05)


def _pp_dynamic_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), UMBRELLA_SYNTHETIC="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as ge
    ge.build()
    from umbrella_amd.parallel import PipelinedDynamicEngine, build_pipelined_engine, shutdown_pipeline
    from umbrella_amd.speculation.speculation_utils import IdTokenizer
    eng = build_pipelined_engine("cuda:0", dtype=torch.float16, seed=11, engine="dynamic", model=NAME, draft_model=NAME,
                                 max_length=512, exit_layer=4, safe_buffer=16, tokenizer=IdTokenizer(), **DYN)
    if eng is not None:
        assert isinstance(eng, PipelinedDynamicEngine)
        a = eng.generate(input_ids=PROMPT, max_new_tokens=20)["generated_tokens"]
        g = eng.generate(input_ids=PROMPT, max_new_tokens=12, temperature=0.0, repetition_penalty=1.0)["generated_tokens"]
        b = eng.generate(input_ids=PROMPT, max_new_tokens=12, temperature=0.7, repetition_penalty=1.05)["generated_tokens"]
        shutdown_pipeline(eng)
        q.put((a, g, b))
    dist.barrier()
    dist.destroy_process_group()


def test_two_stage_dynamic_stochastic_equals_single_process():
    """dynamic w8/b8/d4 (T = 33), temperature 0.7 / top-p 0.9 / top-k 16 / penalty 1.05, seed 11: the last stage samples
    with umb_sample_rows over its own copy of the token history; ids equal the single-process engine's draw for draw.
    Then the knobs change between requests (greedy, back to stochastic): the stages follow through the OP_DECODE plan."""
    import torch.multiprocessing as mp
    import __graft_entry__ as ge
    ge.build()
    os.environ.setdefault("UMBRELLA_SYNTHETIC", "1")
    from umbrella_amd.speculation.dynamic_speculation_engine import DynamicSpeculationEngine
    from umbrella_amd.speculation.speculation_utils import IdTokenizer
    ref = DynamicSpeculationEngine(NAME, NAME, dtype=torch.float16, device="cuda:0", max_length=512, exit_layer=4,
                                   safe_buffer=16, tokenizer=IdTokenizer(), offload=False, seed=11, **DYN)
    ref.initialize()
    ra = ref.generate(input_ids=PROMPT, max_new_tokens=20)["generated_tokens"]
    rg = ref.generate(input_ids=PROMPT, max_new_tokens=12, temperature=0.0, repetition_penalty=1.0)["generated_tokens"]
    rb = ref.generate(input_ids=PROMPT, max_new_tokens=12, temperature=0.7, repetition_penalty=1.05)["generated_tokens"]
    del ref
    torch.cuda.empty_cache()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pp_dynamic_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        a, g, b = q.get(timeout=200)
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    assert a == ra and g == rg and b == rb
    assert len(a) >= 20
