#!/bin/bash
# One headline step (70B-AWQ verify + 1B draft, static tree) as the profiler sees it: the kernels of the LAST step in time
# order, run-length compressed by kernel family, with busy time, gaps, and the per-call durations of the lm_head launches.
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out/prof_step
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out" -- python "$root/bench.py" --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > "$out/run.log" 2>&1
tail -1 "$out/run.log" | cut -c1-160
python - "$out" <<'PY' | tee "$root/gpurun_out/r05_step_timeline.txt"
import csv, glob, os, re, sys
out = sys.argv[1]
kt = glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True)[0]
ks = []
for r in csv.DictReader(open(kt)):
    n = r["Kernel_Name"]
    m = re.match(r"(?:void )?([A-Za-z_0-9:]+)(<[^(]*>)?", n)
    short = (m.group(1) + (m.group(2) or ""))[:60] if m else n[:60]
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1))))
ks.sort()
# steps end with accept_scan_kernel: take the kernels between the last two of them
idx = [i for i, k in enumerate(ks) if k[2].startswith("accept_scan_kernel")]
a, b = idx[-2] + 1, idx[-1] + 1
step = ks[a:b]
print(f"last step: {len(step)} kernels, span {(step[-1][1]-step[0][0])/1e3:.1f} us, busy {sum(e-s for s,e,_,_ in step)/1e3:.1f} us")
# run-length compressed
runs = []
for i, (s, e, n, g) in enumerate(step):
    gap = (s - step[i-1][1]) / 1e3 if i else 0.0
    key = (n, g)
    if runs and runs[-1][0] == key:
        runs[-1][1] += 1; runs[-1][2] += (e - s) / 1e3; runs[-1][3] += gap
    else:
        runs.append([key, 1, (e - s) / 1e3, gap])
fam = {}
for (n, g), c, d, gp in runs:
    f = fam.setdefault((n, g), [0, 0.0, 0.0]); f[0] += c; f[1] += d; f[2] += gp
print("per family in this step: calls, busy us, gap-before us")
for (n, g), (c, d, gp) in sorted(fam.items(), key=lambda x: -x[1][1]):
    print(f"  {n:62s} blocks {g:6d} x{c:5d}  busy {d:9.1f}  avg {d/c:7.2f}  gaps {gp:8.1f}")
print("lm_head-like launches (>= 1000 blocks) in the last 3 steps, in order (us):")
w = ks[idx[-4] + 1:b]
print("  ", [round((e - s) / 1e3, 1) for s, e, n, g in w if g >= 1000 and "skinny" in n])
print("chain launches of the last step in order (us):")
print("  ", [round((e - s) / 1e3, 1) for s, e, n, g in step if "chain" in n])
print("first 70B layer vs later layers, gate/up launches (us):", [round((e - s) / 1e3, 1) for s, e, n, g in step if "skinny" in n and ", 8>" in n][:6])
PY
find "$out" -name "*_trace.csv" -delete
