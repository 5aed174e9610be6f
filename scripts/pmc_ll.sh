#!/bin/bash
# SQ counters of the split-K and low-latency int4 GEMMs on the 70B layer shapes (scripts/ll_bench.py 70b).
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INSTS_MFMA" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  out=$root/gpurun_out/pmc_ll_$i
  rm -rf "$out"; mkdir -p "$out"
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out" -- python "$root/scripts/ll_bench.py" 70b > "$out/run.log" 2>&1
  f=$(find "$out" -name "*counter_collection.csv" | head -1)
  echo "=== set $i: $set"
  if [ -n "$f" ]; then python "$root/scripts/pmc_summary.py" "$f" skinny_gemm ll_gemm; else tail -5 "$out/run.log"; fi
  find "$out" -name "*.csv" -size +20M -delete
done
