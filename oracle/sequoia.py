"""Sequoia growmap generator restated (oracle; test-only).

umbrella/sequoia_utils.py:40-130: greedy level-wise expansion -- at each level
keep the `width` best (parent, rank) candidates by accumulated log-acceptance.
"""
from __future__ import annotations

import torch

DEFAULT_ACC = [0.65, 0.2, 0.1, 0.05]          # sequoia_utils.py:7


def ancestor_mask(successors):
    n = len(successors)
    parent = [-1] * n
    for p, ch in enumerate(successors):
        for c in ch:
            parent[c] = p
    mask = [[0] * n for _ in range(n)]
    for i in range(n):
        j = i
        while j >= 0:
            mask[i][j] = 1
            j = parent[j]
    return mask


def generate(width: int, depth: int, acc=None) -> dict:
    acc = torch.log(torch.tensor(DEFAULT_ACC if acc is None else acc, dtype=torch.float32))
    B = len(acc)
    roots, score, succ, branches, node_depth = [[0]], [[0.0]], [[]], [[0]], [0]
    for i in range(depth):
        roots.append(list(range(i * width + 1, (i + 1) * width + 1)))
        branches.append([0] * width)
        node_depth += [i + 1] * width
        succ += [[] for _ in range(width)]
        cur = torch.tensor(score[i], dtype=torch.float32).repeat_interleave(B)
        cand = acc.repeat(1 if i == 0 else width) + cur
        s, idx = cand.topk(k=width)
        score.append(s.tolist())
        off = 0 if i == 0 else (i - 1) * width + 1
        for child, par in enumerate(sorted((idx // B + off).tolist())):
            succ[par].append(child + i * width + 1)
            branches[i][par - off] += 1
    return {"roots": roots, "branches": branches, "Successors": succ, "mask": ancestor_mask(succ),
            "depth": node_depth, "size": width * depth + 1}
