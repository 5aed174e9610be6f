"""Multi-process CPU tests (gloo, world_size 2 and 3) of the layer-sharding plumbing: control words,
activation hops, returned ids and the commit broadcast.  The stage compute is a stand-in (the real one
is HIP); ordering / shapes / values of everything that crosses ranks is what is checked."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from umbrella_amd.parallel import OP_DECODE, OP_PREFILL, OP_STOP, PipelineComm, split_layers


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
    """Stand-in stages (every stage adds 1 to the activations) driven through the real protocol: one plan word per mode
    change, a prompt walked in fixed chunks, then the decode loop with the commit of iteration i broadcast together with
    the continue flag once rank 0 knows what follows."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, Tmax, max_path, T = 16, 32, 5, 13
    comm = PipelineComm(rank, world, "cpu", H, torch.float32, Tmax, max_path)
    log = []
    P, start, chunk = 7, 100, 4
    if rank == 0:
        comm.plan(OP_PREFILL, P, start, 1, chunk)
        comm.bcast(torch.arange(700, 700 + P, dtype=torch.int32))   # the prompt's token ids follow their plan
        for lo in range(0, P, chunk):
            rows = min(P, lo + chunk) - lo
            h = torch.full((rows, H), 10.0)
            h[:, 0] = torch.arange(lo, lo + rows)
            comm.send_activations(h + 1.0)                       # stage 0 "compute": +1
        log.append(comm.return_ids(n=1).tolist())
        comm.plan(OP_DECODE)
        for it in range(3):
            if it:                                               # commit of the previous iteration + "one more follows"
                comm.share_commit(torch.tensor([2, 9, 0, 50 + it - 1, 3, 0, 0, 0], dtype=torch.int32),
                                  torch.tensor([0, 2, 5, 0, 0], dtype=torch.int32), cont=1,
                                  newtok=torch.tensor([41, 42, 9], dtype=torch.int32))
            h = torch.full((T, H), float(it + 1))
            h[:, 0] = torch.arange(T)
            comm.send_activations(h + 1.0)
            log.append(comm.return_ids(n=T).tolist())
        comm.share_commit(torch.tensor([2, 9, 0, 52, 3, 0, 0, 0], dtype=torch.int32),
                          torch.tensor([0, 2, 5, 0, 0], dtype=torch.int32), cont=0,
                          newtok=torch.tensor([43, 44, 9], dtype=torch.int32))
        comm.plan(OP_STOP)
        q.put(("rank0", log))
    else:
        own = torch.zeros(Tmax, H)                               # the "stage model's hidden buffer"
        while True:
            c = comm.plan()
            if c[0] == OP_STOP:
                break
            if c[0] == OP_PREFILL:
                Pp, st, want, ch = c[1], c[2], c[3], c[4]
                ids = torch.zeros(Pp, dtype=torch.int32)
                comm.bcast(ids)
                log.append(("ids", ids.tolist()))
                for lo in range(0, Pp, ch):
                    rows = min(Pp, lo + ch) - lo
                    h = comm.recv_activations(rows, into=own)
                    assert h.data_ptr() == own.data_ptr()       # received in place, no staging copy
                    h += 1.0
                    comm.send_activations(h)
                    if comm.last and want and lo + ch >= Pp:
                        comm.return_ids((h[-1:, 0] + h[-1:, 1]).int(), 1)
                    log.append(("chunk", rows, st + lo))
            elif c[0] == OP_DECODE:
                cont = 1
                while cont:
                    h = comm.recv_activations(T, into=own)
                    h += 1.0
                    comm.send_activations(h)
                    if comm.last:
                        comm.return_ids((h[:, 0] + h[:, 1]).int(), T)
                    res, path, newtok, cont = comm.share_commit()
                    assert comm.commit_host[:5] == res.tolist()[:5]
                    log.append(("commit", res.tolist(), path.tolist(), cont, newtok.tolist()))
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
    # after `world` stages each adding 1: h[:,0] = t + world, h[:,1] = fill + world
    log0 = got["rank0"]
    assert log0[0] == [6 + world + 10 + world]                   # last prompt row (index 6) after the chunked prefill
    for it, ids in enumerate(log0[1:]):
        assert ids == [t + world + (it + 1) + world for t in range(13)]
    for r in range(1, world):
        log = got[f"rank{r}"]
        assert [e for e in log if e[0] == "chunk"] == [("chunk", 4, 100), ("chunk", 3, 104)]
        commits = [e for e in log if e[0] == "commit"]
        assert [c[3] for c in commits] == [1, 1, 0]                # two "continue", then leave the decode loop
        assert commits[1][1][:5] == [2, 9, 0, 51, 3] and commits[1][2] == [0, 2, 5, 0, 0]
        assert commits[0][4][:3] == [41, 42, 9] and commits[2][4][:3] == [43, 44, 9] and len(commits[0][4]) == 6
        assert [e for e in log if e[0] == "ids"] == [("ids", list(range(700, 707)))]
