#!/bin/bash
# MFMA-busy of the wide verify GEMM (T = 256 and 769) from a PMC pass of its own (counters + kernel-trace only), summarised
# to gpurun_out/${TAG}_pmc_verify_gemm_T{256,769}_mfma_busy.json WITH the sha256/16 of csrc/gemm.hip the pass ran on
# (bench.py reports the figure only while that hash matches the tree).   TAG=r06 bash scripts/r6/pmc_mfma.sh
root=$(cd "$(dirname "$0")/../.." && pwd)
tag=${TAG:-r06}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
for T in ${TS:-256 769}; do
  cmd="T=$T rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -- python scripts/vgemm_bench.py"
  rm -rf "$out/pmc_T$T"
  ( cd /tmp && export TMPDIR=/tmp && T=$T timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY \
      SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/pmc_T$T" -- \
      python "$root/scripts/vgemm_bench.py" - pmc > "$out/pmc_T$T.log" 2>&1 )
  f=$(find "$out/pmc_T$T" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python scripts/pmc_mfma_busy.py "$f" "$root/gpurun_out/${tag}_pmc_verify_gemm_T${T}_mfma_busy.json" "$cmd" verify_gemm vgemm skinny_gemm
  rm -rf "$out/pmc_T$T"
done
