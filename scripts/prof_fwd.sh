#!/bin/bash
# rocprofv3 kernel trace of graph-replayed forwards (scripts/ll_bench.py fwd*), summarised per (kernel, grid).
# Usage (on the GPU box): bash scripts/prof_fwd.sh fwd1b   -> gpurun_out/prof_<what>_by_shape.csv
what=${1:-fwd1b}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof_$what
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out" -- python "$root/scripts/ll_bench.py" "$what" > "$out/run.log" 2>&1
trace=$(find "$out" -name "*kernel_trace.csv" | head -1)
python "$root/scripts/trace_by_shape.py" "$trace" "$root/gpurun_out/prof_${what}_by_shape.csv"
tail -3 "$out/run.log"
rm -f "$trace"   # raw traces are large; the summary is what is kept
