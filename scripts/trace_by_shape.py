"""Summarise a rocprofv3 kernel trace per (kernel, grid size): calls, average / min / max duration.
The --stats table merges the four layer GEMM shapes that share one template instantiation; this keeps them apart so
bench.py's per-shape roofline durations can be checked against the profiler.
  python scripts/trace_by_shape.py <kernel_trace.csv> <out.csv> [name-substring ...]"""
import collections
import csv
import sys

src, dst, pats = sys.argv[1], sys.argv[2], sys.argv[3:]
agg = collections.defaultdict(list)
with open(src) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if pats and not any(p in name for p in pats):
            continue
        if "Grid_Size_X" in r:      # work-items per dimension; the skinny GEMMs launch (n-tile groups, K splits) grids
            blocks = 1
            for d in "XYZ":
                blocks *= max(1, int(r.get(f"Grid_Size_{d}", 1) or 1)) // max(1, int(r.get(f"Workgroup_Size_{d}", 1) or 1))
        else:
            blocks = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
        agg[(name.split("(")[0][:90], blocks)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = sorted(((k, v) for k, v in agg.items()), key=lambda kv: -sum(kv[1]))
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "blocks", "calls", "avg_us", "min_us", "max_us", "total_ms"])
    for (name, blocks), v in rows:
        w.writerow([name, blocks, len(v), round(sum(v) / len(v) / 1e3, 2), round(min(v) / 1e3, 2), round(max(v) / 1e3, 2),
                    round(sum(v) / 1e6, 3)])
print("wrote", dst, len(rows), "rows")
