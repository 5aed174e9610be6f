"""Controllable-acceptance measurement loop shared by bench.py's sharded legs (pp / tp) -- the same procedure the
headline uses (bench.py: headline), so that their `ms_per_step` compare 1 : 1 with the N = 1 line.

Random-weight draft / target pairs accept ~0 drafted tokens, so acceptance is an INPUT: the target's own greedy
continuation is recorded first (raw draft), re-recorded under the knob's own execution pattern until it is a fixed
point (a token verified as a depth-d node sums its attention in another order than as a root: 16-bit near-ties), then
`StaticSpeculationEngine.set_oracle_draft` places it in the tree with the reference's acceptance vector.  Every draft
and verify kernel still runs; only <= depth token ids per iteration change.
"""
from __future__ import annotations

import time

import torch


def _run(eng, n):
    start = eng.num_nodes
    for _ in range(n):
        if not eng.validate_status():
            break
        eng.step()
    return eng.num_nodes - start


def disable_eos(eng):
    """Synthetic-weight measurements only: random-init logits make the EOS ids as likely as any other token, and an
    accepted EOS ends a request (keep = its position: an EOS root keeps 0 tokens, so a fixed-length timing loop would
    never advance).  The accept scan is launched with an empty EOS list instead -- its arguments are part of the captured
    iteration, so call this before the first step()."""
    eng.eos_tokens = []
    eng.eos_dev = torch.tensor([-1], dtype=torch.int32, device=eng.device)
    eng._graph = None


def steered_measure(eng, prompt, acc, seed, warmup, steps, levels, barrier=None):
    """-> dict(ms_per_step, tokens, accept_len, raw_tokens_per_s, raw_accept_len, passes, divergence).
    `barrier`: optional callable bracketing the timed region (multi-rank SPMD engines)."""
    need = (warmup + steps) * levels + 16
    disable_eos(eng)
    assert eng._prefill(prompt)
    start = eng.num_nodes
    torch.cuda.synchronize()
    t0 = time.time()
    raw_steps = 0
    while eng.num_nodes - start < need and eng.validate_status():
        eng.step()
        raw_steps += 1
    torch.cuda.synchronize()
    raw_dt = time.time() - t0
    raw_tokens = eng.num_nodes - start
    truth = eng.tokens[start:eng.num_nodes + 1].tolist()
    passes, div = 0, 0
    for passes in range(1, max(8, min(40, (warmup + steps + 15) // 16)) + 1):
        eng.reset()
        assert eng._prefill(prompt)
        eng.set_oracle_draft(truth, start, acc, seed=seed)
        _run(eng, warmup + steps)
        div = eng.diverged
        while eng.num_nodes - start < need and eng.validate_status():
            eng.step()
        new_truth = eng.tokens[start:eng.num_nodes + 1].tolist()
        if div == 0:
            break
        truth = new_truth
    eng.reset()
    assert eng._prefill(prompt)
    eng.set_oracle_draft(truth, start, acc, seed=seed)
    _run(eng, warmup)
    torch.cuda.synchronize()
    if barrier:
        barrier()
    t0 = time.time()
    tokens = _run(eng, steps)
    torch.cuda.synchronize()
    if barrier:
        barrier()
    dt = time.time() - t0
    return {"ms_per_step": round(dt / steps * 1e3, 4), "tokens": tokens, "tokens_per_s": round(tokens / dt, 2),
            "accept_len": round(tokens / steps, 3), "tokens_per_s_raw_draft": round(raw_tokens / raw_dt, 2),
            "accept_len_raw_draft": round(raw_tokens / max(raw_steps, 1), 3), "oracle_draft_passes": passes,
            "continuation_head": [int(t) for t in truth[:8]],
            "oracle_draft_divergence": getattr(eng, "diverged", div)}


def device_census(device, dist=None):
    """Which physical device does every rank of the group drive?  -> dict(devices = sorted distinct identities,
    n_distinct, n_ranks).  A mis-bound launch (eight ranks on one GPU) must not be able to report an 8-GPU number."""
    p = torch.cuda.get_device_properties(device)
    ident = str(getattr(p, "uuid", "")) or f"{p.name}:{getattr(p, 'pci_bus_id', '?')}:{getattr(p, 'pci_device_id', '?')}"
    ident = f"{ident}|bus{getattr(p, 'pci_bus_id', '?')}"
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        ids = [ident]
    else:
        ids = [None] * dist.get_world_size()
        dist.all_gather_object(ids, ident)
    return {"devices": sorted(set(ids)), "n_distinct": len(set(ids)), "n_ranks": len(ids)}
