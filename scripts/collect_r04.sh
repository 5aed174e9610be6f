#!/bin/bash
# Round-4 evidence for the wide (T >= 257) regimes, one GPU call; everything lands in gpurun_out/r04/ (the summaries to be
# judged are copied into profiles/ by hand).  Usage: bash scripts/collect_r04.sh [tag]
root=$(cd "$(dirname "$0")/.." && pwd)
tag=${1:-r04}
out=$root/gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out"
cd "$root"
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
PATS="verify_gemm skinny_gemm ll_gemm gv_kernel reduce_ tree_attn topk beam sample accept kv_compact embed rmsnorm argmax sum_splits"
# 1. kernel-trace stats + by-shape tables of BASELINE configs 3 (target resident) and 4
for cfg in c3-resident c4; do
  steps=6; [ $cfg = c4 ] && steps=3
  ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${cfg}_stats" -- \
      python "$root/scripts/bench_configs.py" --config $cfg --steps $steps > "$out/${cfg}_under_rocprof.log" 2>&1 )
  f=$(find "$out/${cfg}_stats" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$out/${tag}_${cfg}_kernel_stats.csv"
  t=$(find "$out/${cfg}_stats" -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python scripts/trace_by_shape.py "$t" "$out/${tag}_${cfg}_kernels_by_shape.csv" $PATS
  find "$out/${cfg}_stats" -name "*kernel_trace.csv" -delete
  tail -1 "$out/${cfg}_under_rocprof.log" | cut -c1-300
done
# 2. MFMA-busy of the wide verify GEMM (PMC pass of its own: counters + kernel-trace only)
for T in 256 769; do
  cmd="T=$T rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -- python scripts/vgemm_bench.py"
  ( cd /tmp && export TMPDIR=/tmp && T=$T timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY \
      SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/pmc_T$T" -- \
      python "$root/scripts/vgemm_bench.py" - pmc > "$out/pmc_T$T.log" 2>&1 )
  f=$(find "$out/pmc_T$T" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python scripts/pmc_mfma_busy.py "$f" "$out/${tag}_pmc_verify_gemm_T${T}_mfma_busy.json" "$cmd" verify_gemm skinny_gemm
  find "$out/pmc_T$T" -name "*.csv" -size +20M -delete
done
# 3. un-profiled timing of the same microbench (never compare a profiled arm with an un-profiled one)
for T in 256 257 769; do T=$T python scripts/vgemm_bench.py - plain 2>&1 | grep "T=" >> "$out/${tag}_vgemm_bench.txt"; done
cat "$out/${tag}_vgemm_bench.txt"
ls -la "$out"
