#!/bin/bash
# 70B verify layer at T = 13 (graph-replayed 16-layer forward, split schedule): plan overrides of o / qkv / down, per-kernel durations
# from the profiler.  tb byte: low 6 bits tiles per block, 0x40 = 4 k-blocks per chunk (8-stage weight ring), 0x80 = 8 waves per block.
# NOTE: the 0x40 bit (a deeper weight ring for one linear) was an experiment hook of this round and is not in the product: see
# profiles/r05_plan_sweep_70b_negative.txt; without it the 0x40 values below are rejected as tile counts (the plan stays as shipped).
root=$(cd "$(dirname "$0")/../.." && pwd)
cd /tmp && export TMPDIR=/tmp
run() {
  out=$root/gpurun_out/prof_plan; rm -rf "$out"; mkdir -p "$out"
  UMB_PLAN_OVR="$1" SCHEDS=split rocprofv3 --kernel-trace --output-format csv -d "$out" -- python "$root/scripts/ll_bench.py" fwd70b > "$out/run.log" 2>&1
  trace=$(find "$out" -name "*kernel_trace.csv" | head -1)
  python "$root/scripts/trace_by_shape.py" "$trace" "$out/by_shape.csv" skinny_gemm reduce_ > /dev/null
  echo "== ${1:-plan}  $(grep 'forward' "$out/run.log" | tail -1 | sed 's/.*T=13: //' | cut -c1-40)"
  python - "$out/by_shape.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["calls"]) >= 100:
        print(f"     {r['kernel'][5:52]:48s} blocks {r['blocks']:>5s}  avg {r['avg_us']:>6s}  min {r['min_us']:>6s}")
PY
  rm -f "$trace"
}
run ""
run "8192,8192:1,4,200,0"
run "8192,8192:2,4,64,0"
run "8192,28672:2,8,64,0"
run "10240,8192:1,3,200,0"
run "10240,8192:2,7,72,0"
run "8192,8192:1,4,200,0;8192,28672:2,8,64,0;10240,8192:1,3,200,0"
