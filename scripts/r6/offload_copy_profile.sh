#!/bin/bash
# C3 with 40 resident layers: every H2D slab copy of the run (count, gaps) and of the last steady step (duration, gap before it).
# Answers: is the link ever idle inside a steady step, and how many slabs cross it per step.
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out/prof_offload40
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$out" -- python "$root/scripts/bench_configs.py" --config c3 --cache-layers ${NCL:-40} --steps 4 > "$out/run.log" 2>&1
grep ms_per_step "$out/run.log" | cut -c1-260
python - "$out" <<'PY'
import csv, glob, os, sys, bisect
out = sys.argv[1]
mt = glob.glob(os.path.join(out, "**", "*memory_copy_trace.csv"), recursive=True)[0]
cp = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(mt))
            if "HOST_TO_DEVICE" in r["Direction"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 3e6)
print("big H2D copies in the whole run:", len(cp), "| span of all ms", (cp[-1][1] - cp[0][0]) / 1e6)
allgaps = [((cp[i + 1][0] - cp[i][1]) / 1e6, i) for i in range(len(cp) - 1)]
print("gaps > 0.1 ms between consecutive copies (ms @ index):", [(round(g, 2), i) for g, i in allgaps if g > 0.1])
n = int(os.environ.get("NSTREAM", "40"))
cp = cp[-n:]
print("idx  gap_ms  dur_ms  GB/s")
for i, (s_, e) in enumerate(cp):
    gap = (s_ - cp[i - 1][1]) / 1e6 if i else 0.0
    print(f"{i:3d} {gap:7.2f} {(e - s_) / 1e6:7.2f} {444.5 / ((e - s_) / 1e6):6.1f}")
print("step span ms", (cp[-1][1] - cp[0][0]) / 1e6, "sum copy ms", sum(e - s for s, e in cp) / 1e6)
PY
find "$out" -name "*_trace.csv" -delete
