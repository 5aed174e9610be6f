import os, sys, socket
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch


def worker(rank, world, port):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), UMBRELLA_SYNTHETIC="1", UMB_TP_ALLREDUCE="peer", UMB_TP_SPIN="200000")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import load_golden
    from umbrella_amd.models.config import LlamaCfg
    from umbrella_amd.models.synthetic import synth_state_small
    from umbrella_amd.tensor_parallel import TensorParallelLlama, TPComm
    G = load_golden()
    cfg = LlamaCfg(**dict(G["target_cfg"], eos_token_id=[3, 5]))
    sd = synth_state_small(cfg, G["seeds"]["target"])
    comm = TPComm()
    tp = TensorParallelLlama.build(cfg, sd, comm, 256, "cuda:0", torch.float16)
    calls = []
    orig = comm.all_reduce
    comm.all_reduce = lambda t, _o=orig: (calls.append(t.numel()), _o(t))[1]
    print(rank, "check1", tp.peer_self_check(), tp.last_self_check, "hook calls", calls, flush=True)
    print(rank, "peer_max_floats", tp.m._tp.peer_max_floats, "peer ptr", bool(tp.m._tp.peer), "world", tp.m._tp.world, "epoch", tp.peer.words[0].item(), flush=True)
    good = [tp.peer.desc.slot[r] for r in range(world)]
    print(rank, "slots", [hex(x) for x in good], flush=True)
    for r in range(world):
        tp.peer.desc.slot[r] = good[rank]
    print(rank, "bent", [hex(tp.peer.desc.slot[r]) for r in range(world)], flush=True)
    print(rank, "check2", tp.peer_self_check(), tp.last_self_check, tp.allreduce_path[:40], flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, 2, port)) for r in range(2)]
    for p in ps: p.start()
    for p in ps: p.join(timeout=100)
    for p in ps:
        if p.is_alive(): p.kill()
