#!/bin/bash
# SQ counters of one 70B layer shape (ONLY=gu by default) on both GEMM families: where do the waves spend their time?
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
export ONLY=${ONLY:-gu}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INSTS_MFMA" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE32_INSTS" ; do
  i=$((i+1))
  out=$root/gpurun_out/pmc_gu_$i
  rm -rf "$out"; mkdir -p "$out"
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out" -- python "$root/scripts/ll_bench.py" 70b > "$out/run.log" 2>&1
  f=$(find "$out" -name "*counter_collection.csv" | head -1)
  echo "=== set $i: $set"
  if [ -n "$f" ]; then python "$root/scripts/pmc_summary.py" "$f" skinny_gemm ll_gemm; else tail -5 "$out/run.log"; fi
  find "$out" -name "*.csv" -size +20M -delete
done
