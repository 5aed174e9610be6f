#!/bin/bash
# link idle time inside the steady steps of C3 with 40 resident layers: H2D slab copies from a kernel + memory-copy trace
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out/prof_offload40
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$out" -- python "$root/scripts/bench_configs.py" --config c3 --cache-layers 40 --steps 4 > "$out/run.log" 2>&1
tail -1 "$out/run.log" | cut -c1-200
python - "$out" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
mt = glob.glob(os.path.join(out, "**", "*memory_copy_trace.csv"), recursive=True)[0]
kt = glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True)[0]
cp = []
rows = list(csv.DictReader(open(mt)))
print("columns:", list(rows[0].keys()) if rows else None, "rows", len(rows))
for r in rows:                                  # no byte column in this rocprofv3: a 444.5 MB slab copy is the only one that takes milliseconds
    s0, e0 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "HOST_TO_DEVICE" in r["Direction"] and e0 - s0 > 3e6:
        cp.append((s0, e0, 444.5e6))
cp.sort()
cp = cp[-120:]                               # the last three steps' 40 copies each
durs = [(e - s) / 1e6 for s, e, _ in cp]
gaps = [(cp[i + 1][0] - cp[i][1]) / 1e6 for i in range(len(cp) - 1)]
print("copies", len(cp), "dur ms min/med/max", round(min(durs), 2), round(sorted(durs)[len(durs) // 2], 2), round(max(durs), 2),
      "GB/s med", round(cp[0][2] / sorted(durs)[len(durs) // 2] / 1e6, 1))
big = [(i, round(g, 2)) for i, g in enumerate(gaps) if g > 0.5]
print("gaps > 0.5 ms (index in the last 120 copies, ms):", big)
print("sum of gaps per step ms:", round(sum(gaps) / 3, 1))
# which kernels run inside the biggest gap
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50]) for r in csv.DictReader(open(kt))]
if big:
    i = max(big, key=lambda x: x[1])[0]
    g0, g1 = cp[i][1], cp[i + 1][0]
    inside = [k for k in ks if k[0] < g1 and k[1] > g0]
    names = {}
    for k in inside: names[k[2]] = names.get(k[2], 0) + 1
    print("kernels overlapping the largest gap:", sorted(names.items(), key=lambda x: -x[1])[:8])
PY
head -3 $(find "$out" -name "*memory_copy_trace.csv" | head -1); find "$out" -name "*_trace.csv" -delete
