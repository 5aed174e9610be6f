"""bench.py launch plumbing: `--gpus N` must come up as N ranks (self-launched when started as a plain script), report
n_gpus = N on ONE JSON line with the max-over-ranks timing, and refuse to report an N-GPU number from fewer devices."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                          timeout=timeout, env=e, cwd=ROOT)


def _json_lines(stdout):
    return [json.loads(l) for l in stdout.splitlines() if l.startswith("{")]


def test_gpus2_dry_run_self_launches_two_ranks():
    """plain `python bench.py --gpus 2 --dry-run`: re-executes itself under torch.distributed.run, two gloo ranks,
    one line from rank 0 with n_gpus = 2; the slower rank sets the time (max over ranks), the units are summed."""
    r = _run(["--gpus", "2", "--steps", "5", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    out = lines[0]
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["dry_run"] is True
    assert out["accept_len"] == 3.0                              # 3 units per step per rank: summed, then per rank again
    assert out["ms_per_step"] >= 3.9                             # rank 1 sleeps 4 ms per step: the max, not rank 0's 2 ms
    assert "pp" in out and "tp" in out


def test_gpus_must_match_world_size():
    r = _run(["--gpus", "2", "--dry-run"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_refuses_more_gpus_than_the_box_has():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run(["--gpus", str(have + 2), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "refusing" in (r.stderr + r.stdout)
    assert not _json_lines(r.stdout)


@pytest.mark.gpu
def test_gpus2_shared_gpu_runs_replicas_pp_and_tp():
    """Two ranks on this box's one GPU (gloo, host-staged hops), tiny models: the real multi-rank code path of
    `bench.py --gpus 2` end to end -- replicas headline with n_gpus = 2, then the layer-sharded (pp) and tensor-parallel
    (tp) engines over both ranks on the same line."""
    r = _run(["--gpus", "2", "--steps", "6", "--warmup", "2", "--workload", "tiny", "--max-length", "512", "--prompt-len", "24",
              "--sharded-steps", "6", "--phase-timeout", "300"], env={"UMB_BENCH_SHARE_GPU": "1"}, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    out = lines[0]
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert out["pp"].get("n_ranks_rccl") == 2 and out["pp"]["ms_per_step"] > 0, out["pp"]
    assert out["tp"].get("n_ranks_rccl") == 2 and out["tp"]["ms_per_step"] > 0, out["tp"]
    # round 4: the sharded legs run under the headline's acceptance knob (same acc vector and seed), so their
    # ms_per_step / accept_len compare with the N = 1 line; the device census sees that both ranks share ONE device here
    for leg in ("pp", "tp"):
        assert out[leg]["accept_len"] > 1.5 and out[leg]["acc"] == out["config"]["acc"], out[leg]
        assert out[leg]["n_distinct_devices"] == 1 and len(out[leg]["devices"]) == 1, out[leg]
        assert out[leg]["oracle_draft_divergence"] == 0
    assert "hop_us" in out["pp"] and "allreduce_us" in out["tp"] and out["status"] == "ok"
    # the layer-sharded target IS the single-GPU model (same seeded weights, same kernels, same prompt as replica rank 0):
    # its recorded continuation must be the replica's, token for token (round 4: an untied last stage drew another lm_head)
    assert out["pp"]["continuation_head"] == out["continuation_head"], (out["pp"]["continuation_head"], out["continuation_head"])
    assert out["pp"]["accept_len_raw_draft"] == out["accept_len_raw_draft"]
