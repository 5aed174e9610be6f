"""Sequoia growmap construction (replaces umbrella/sequoia_utils.py:40-130).

``generate_sequoia_tree(width, depth, acc)`` grows the tree level by level, keeping at
every level the `width` (parent, rank) candidates with the highest accumulated
log-acceptance; the JSON schema (roots / branches / Successors / mask / depth / size)
is the one the engines and the reference's shipped trees use.
"""
from __future__ import annotations

import heapq
import json
import math

DEFAULT_ACC = [0.65, 0.2, 0.1, 0.05]


def successor_list_to_mask(successors):
    n = len(successors)
    parent = [-1] * n
    for p, kids in enumerate(successors):
        for c in kids:
            parent[c] = p
    rows = []
    for i in range(n):
        row = [0] * n
        j = i
        while j != -1:
            row[j] = 1
            j = parent[j]
        rows.append(row)
    return rows


def generate_sequoia_tree(width: int, depth: int, acc=None, json_file=None):
    if acc is None:
        assert width <= 4, "Using default acceptance rate vector, require width<=4"
        acc = DEFAULT_ACC
    import torch
    lacc = torch.log(torch.tensor(acc, dtype=torch.float32))
    nb = len(lacc)
    roots, branches, succ, tdepth = [[0]], [[0]], [[]], [0]
    scores = torch.zeros(1, dtype=torch.float32)
    for lvl in range(depth):
        first = lvl * width + 1
        roots.append(list(range(first, first + width)))
        branches.append([0] * width)
        tdepth += [lvl + 1] * width
        succ += [[] for _ in range(width)]
        cand = (lacc[None, :] + scores[:, None]).reshape(-1)                          # (parent, rank) flattened
        # torch.topk decides ties (three-way ties occur with the default vector); the shipped
        # growmaps were produced with it, so the selection rule is part of the data format
        new_scores, order = cand.topk(k=width)
        base = 0 if lvl == 0 else (lvl - 1) * width + 1
        pars = sorted(int(i) // nb + base for i in order)
        for child, par in enumerate(pars):
            succ[par].append(first + child)
            branches[lvl][par - base] += 1
        scores = new_scores
    out = {"roots": roots, "branches": branches, "Successors": succ, "mask": successor_list_to_mask(succ),
           "depth": tdepth, "size": width * depth + 1}
    if json_file is not None:
        with open(json_file, "w") as f:
            json.dump(out, f, indent=4)
    return out


def generate_budget_tree(size: int, max_depth: int, acc=None, json_file=None):
    """Growmap with a node budget instead of a fixed width per level: the `size - 1` non-root nodes with the highest
    path acceptance probability among all nodes of depth <= max_depth (rank-r child of a node multiplies the path by
    acc[r]).  For a given budget this maximises the expected accept length under the rank-acceptance model, and the
    levels get the widths the probabilities ask for (narrow near the root, where only len(acc) children exist, wide
    where the mass is).  Depth is capped because every level costs one draft forward while the verify is flat in the
    tree size on MI355X (scripts/tune_growmap.py).  Same JSON schema as generate_sequoia_tree: nodes are numbered level
    by level, within a level by (parent order, rank) -- the order the engines place top-k children in."""
    acc = list(DEFAULT_ACC if acc is None else acc)
    assert size >= 1 and max_depth >= 1
    # best-first expansion; heap entries: (-log p, tie-break counter, parent id, rank, depth)
    nodes = [(-1, 0, 0)]                                  # (parent, rank, depth) in pick order; node 0 = root
    heap, cnt = [], 0
    def push_children(pid, logp, depth):
        nonlocal cnt
        if depth >= max_depth:
            return
        # only the rank-0 child is pushed; a node's rank-(r+1) sibling enters the heap when rank r is picked, so
        # ranks are always taken in order (acc is non-increasing)
        if acc[0] > 0:
            heapq.heappush(heap, (-(logp + math.log(acc[0])), cnt, pid, 0, depth + 1, logp))
            cnt += 1
    push_children(0, 0.0, 0)
    while len(nodes) < size and heap:
        nlp, _, pid, rank, depth, plogp = heapq.heappop(heap)
        nid = len(nodes)
        nodes.append((pid, rank, depth))
        push_children(nid, -nlp, depth)
        if rank + 1 < len(acc) and acc[rank + 1] > 0:
            heapq.heappush(heap, (-(plogp + math.log(acc[rank + 1])), cnt, pid, rank + 1, depth, plogp))
            cnt += 1
    # renumber level-major, within a level by (parent's new index, rank)
    kids = {}
    for nid, (pid, rank, depth) in enumerate(nodes):
        if nid:
            kids.setdefault(pid, []).append((rank, nid))
    order, level = [0], [0]
    levels = [[0]]
    while True:
        nxt = []
        for old in level:
            nxt += [nid for _, nid in sorted(kids.get(old, []))]
        if not nxt:
            break
        levels.append(nxt)
        order += nxt
        level = nxt
    new_id = {old: new for new, old in enumerate(order)}
    n = len(order)
    succ = [[] for _ in range(n)]
    tdepth = [0] * n
    for old in order:
        succ[new_id[old]] = [new_id[c] for _, c in sorted(kids.get(old, []))]
        tdepth[new_id[old]] = nodes[old][2]
    roots = [[new_id[o] for o in lv] for lv in levels]
    branches = [[len(succ[i]) for i in lv] for lv in roots]
    out = {"roots": roots, "branches": branches, "Successors": succ, "mask": successor_list_to_mask(succ),
           "depth": tdepth, "size": n}
    if json_file is not None:
        with open(json_file, "w") as f:
            json.dump(out, f, indent=4)
    return out


def growmap_from_branches(branches, json_file=None):
    """Growmap from its branch table alone: ``branches[lvl][j]`` children for the j-th node of level lvl, nodes numbered
    level by level and children laid out by (parent order, rank) -- the layout every growmap of this format has
    (umbrella/sequoia_utils.py:110-113).  Used for topologies the score-greedy generator cannot reproduce from any
    acceptance vector (a level where a lower-scored node keeps a child its higher-scored sibling lost: produced by
    score ties in the original run)."""
    roots, succ, tdepth = [[0]], [[]], [0]
    for lvl, row in enumerate(branches):
        assert len(row) == len(roots[lvl]), (lvl, row)
        nxt = []
        for j, b in enumerate(row):
            kids = list(range(len(succ), len(succ) + b))
            succ[roots[lvl][j]] = kids
            succ += [[] for _ in kids]
            tdepth += [lvl + 1] * b
            nxt += kids
        if not nxt:
            assert all(sum(r) == 0 for r in branches[lvl:]), "branch rows after an empty level must be zero"
            break
        roots.append(nxt)
    out = {"roots": roots, "branches": [list(r) for r in branches[:len(roots)]], "Successors": succ,
           "mask": successor_list_to_mask(succ), "depth": tdepth, "size": len(succ)}
    if json_file is not None:
        with open(json_file, "w") as f:
            json.dump(out, f, indent=4)
    return out


def expected_accept_length(growmap: dict, acc) -> float:
    """E[#accepted tokens per verify] (root + bonus counted as in the engines' dec_len/steps)
    if the rank-r child of any node is accepted with probability acc[r]."""
    succ = growmap["Successors"]
    reach = [0.0] * len(succ)
    reach[0] = 1.0
    for p, kids in enumerate(succ):
        for r, c in enumerate(kids):
            reach[c] = reach[p] * (acc[r] if r < len(acc) else 0.0)
    return sum(reach)


def measure_acceptance_rate(draft_model, target_model, input_ids, solution_len: int, width: int):
    """Acceptance counts of one sequence (examples/construct_sequoia.py:66-90 of the reference): over the last
    `solution_len` positions, how often the target's arg-max token is the draft's rank-r choice, r < width.
    Both models run one causal forward over `input_ids` (LongTensor [1, P]); the ranks come from the device
    top-k / arg-max kernels on the fp32 logits.  Returns (counts float32 [width] on the device, solution_len)."""
    import torch
    from . import _lib
    P = input_ids.shape[1]
    dev = target_model.device
    pos = torch.arange(P, device=dev)
    mask = torch.tril(torch.ones(P, P, dtype=torch.bool, device=dev))
    lo = P - solution_len - 1
    counts = torch.zeros(width, dtype=torch.float32, device=dev)
    lt = target_model.inference(input_ids, pos[None], mask, pos)[0, lo:P - 1].contiguous()
    V = lt.shape[-1]
    best = torch.empty(solution_len, dtype=torch.int32, device=dev)
    _lib.call("umb_argmax_rows", best, lt, solution_len, V)
    ld = draft_model.inference(input_ids, pos[None], mask, pos)[0, lo:P - 1].contiguous()
    top = torch.empty(solution_len, width, dtype=torch.int32, device=dev)
    _lib.call("umb_topk_rows", top, None, ld, solution_len, V, width, None, None, None, None)
    counts += (top == best[:, None]).float().sum(dim=0)
    target_model.clear()
    draft_model.clear()
    return counts, solution_len
