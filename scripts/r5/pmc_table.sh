#!/bin/bash
# Round 5, counter evidence below the dominant kernel: for the graph-replayed forwards
#   fwd70b   (70B-AWQ, 16 layers, T = 13: the headline verify layer's kernels)
#   fwd1b    (1B fp16, T = 3: the draft layer's kernels)
#   fwd8bawq (8B-AWQ, T = 32: the dynamic engines' draft levels)
# one kernel-trace pass (durations) and three SEPARATE --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ wait split), each
# summarised per (kernel, workgroups) by scripts/r5/pmc_table.py into gpurun_out/<tag>_pmc_table_<fwd>.csv.
# Usage (GPU box): TAG=r05 bash scripts/r5/pmc_table.sh [fwd70b fwd1b fwd8bawq]
root=$(cd "$(dirname "$0")/../.." && pwd)
tag=${TAG:-r05}
what=${@:-fwd70b fwd1b fwd8bawq}
export SCHEDS=${SCHEDS:-auto} T1B=${T1B:-3} T8B=${T8B:-32} T70=${T70:-13}
cd /tmp && export TMPDIR=/tmp
for w in $what; do
  base=$root/gpurun_out/${tag}_pmc_$w
  rm -rf "$base"; mkdir -p "$base"
  rocprofv3 --kernel-trace --output-format csv -d "$base/trace" -- python "$root/scripts/ll_bench.py" $w > "$base/trace.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$base/fetch" -- python "$root/scripts/ll_bench.py" $w > "$base/fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$base/write" -- python "$root/scripts/ll_bench.py" $w > "$base/write.log" 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d "$base/sq" -- python "$root/scripts/ll_bench.py" $w > "$base/sq.log" 2>&1
  python "$root/scripts/r5/pmc_table.py" "$base" "$root/gpurun_out/${tag}_pmc_table_$w.csv" $w
  tail -2 "$base/trace.log"
  find "$base" -name "*.csv" -size +8M -delete
done
