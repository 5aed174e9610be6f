"""Multi-process CPU tests (gloo, world_size 2 and 3) of the layer-sharding plumbing: control words,
activation hops, returned ids and the commit broadcast.  The stage compute is a stand-in (the real one
is HIP); ordering / shapes / values of everything that crosses ranks is what is checked."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from umbrella_amd.parallel import (OP_CHUNK, OP_COMMIT, OP_STOP, OP_TREE, PipelineComm, split_layers)


def test_split_layers():
    assert split_layers(80, 8) == [(i * 10, i * 10 + 10) for i in range(8)]
    assert split_layers(16, 3) == [(0, 6), (6, 11), (11, 16)]
    assert split_layers(4, 1) == [(0, 4)]
    for L, w in ((80, 4), (32, 8), (7, 2)):
        r = split_layers(L, w)
        assert r[0][0] == 0 and r[-1][1] == L and all(a[1] == b[0] for a, b in zip(r, r[1:]))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, Tmax, max_path = 16, 32, 5
    comm = PipelineComm(rank, world, "cpu", H, torch.float32, Tmax, max_path)
    log = []
    if rank == 0:
        for it in range(3):
            T = 13 if it < 2 else 7
            c = comm.command(OP_TREE if it < 2 else OP_CHUNK, T, 100 + it, 1)
            h = torch.full((T, H), float(it + 1))
            h[:, 0] = torch.arange(T)
            comm.send_activations(h + 1.0)                       # stage 0 "compute": +1
            ids = comm.return_ids(n=(T if it < 2 else 1))
            log.append(ids.tolist())
            comm.command(OP_COMMIT)
            res = torch.tensor([2, 9, 0, 50 + it, 3, 0, 0, 0], dtype=torch.int32)
            comm.share_commit(res, torch.tensor([0, 2, 5, 0, 0], dtype=torch.int32))
        comm.command(OP_STOP)
        q.put(("rank0", log))
    else:
        while True:
            c = comm.command()
            if c[0] == OP_STOP:
                break
            if c[0] in (OP_TREE, OP_CHUNK):
                T = c[1]
                h = comm.recv_activations(T).clone()
                h = h + 1.0                                      # every stage adds 1
                comm.send_activations(h)
                if comm.last:
                    n = T if c[0] == OP_TREE else 1
                    ids = (h[:, 0] + h[:, 1]).int()              # function of the fully processed activations
                    comm.return_ids(ids, n)
                log.append((c[0], T, c[2]))
            elif c[0] == OP_COMMIT:
                res, path = comm.share_commit()
                log.append(("commit", res.tolist(), path.tolist()))
        q.put((f"rank{rank}", log))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pipeline_protocol_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # after `world` stages each adding 1: h[:,0] = t + world, h[:,1] = it+1 + world
    for it, ids in enumerate(got["rank0"]):
        T = 13 if it < 2 else 7
        exp = [t + world + (it + 1) + world for t in range(T)]
        assert ids == (exp if it < 2 else exp[:1])
    for r in range(1, world):
        log = got[f"rank{r}"]
        assert [e for e in log if e[0] != "commit"] == [(OP_TREE, 13, 100), (OP_TREE, 13, 101), (OP_CHUNK, 7, 102)]
        commits = [e for e in log if e[0] == "commit"]
        assert len(commits) == 3 and commits[1][1][:5] == [2, 9, 0, 51, 3] and commits[1][2] == [0, 2, 5, 0, 0]
