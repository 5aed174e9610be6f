#!/bin/bash
# End-of-round-6 evidence, one GPU call; everything lands in gpurun_out/r06f/ (the summaries to be judged are copied into profiles/).
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out/r06f
rm -rf "$out"; mkdir -p "$out"
cd "$root"
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
PATS="vgemm_w verify_gemm skinny_gemm ll_gemm gv_kernel draft_chain draft_head reduce_ tree_attn topk beam sample accept kv_compact embed rmsnorm argmax sum_splits"
# 1. PMC traffic of the dominant kernel (separate FETCH_SIZE / WRITE_SIZE passes), tied to the sha of gemm.hip
TAG=r06 bash scripts/pmc_traffic_tag.sh > "$out/pmc_traffic.log" 2>&1
cp gpurun_out/r06_pmc_gemm70b_traffic.json "$out/" 2>/dev/null
# 2. kernel-trace stats of the bench command (headline only) + by-shape table
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out/bench_stats" -- \
    python "$root/bench.py" --steps 16 --warmup 2 --no-cpu-baseline --no-secondary > "$out/bench_under_rocprof.log" 2>&1 )
f=$(find "$out/bench_stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/r06_bench70b_kernel_stats.csv"
t=$(find "$out/bench_stats" -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python scripts/trace_by_shape.py "$t" "$out/r06_bench70b_kernels_by_shape.csv" $PATS
find "$out/bench_stats" -name "*kernel_trace.csv" -delete
# 3. BASELINE configs 3 (target resident) and 4: kernel stats + by shape
for cfg in c3-resident c4; do
  steps=6; [ $cfg = c4 ] && steps=3
  ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${cfg}_stats" -- \
      python "$root/scripts/bench_configs.py" --config $cfg --steps $steps > "$out/${cfg}_under_rocprof.log" 2>&1 )
  f=$(find "$out/${cfg}_stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/r06_${cfg}_kernel_stats.csv"
  t=$(find "$out/${cfg}_stats" -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python scripts/trace_by_shape.py "$t" "$out/r06_${cfg}_kernels_by_shape.csv" $PATS
  find "$out/${cfg}_stats" -name "*kernel_trace.csv" -delete
done
# 4. per-kernel counter table of the 70B forward at 13 rows (FETCH / WRITE / SQ passes)
TAG=r06 bash scripts/r5/pmc_table.sh fwd70b > "$out/pmc_table.log" 2>&1
cp gpurun_out/r06_pmc_table_*.csv "$out/" 2>/dev/null
# 5. MFMA-busy of the wide verify GEMM, tied to the sha of gemm.hip and vgemm.hip; clock / power trace of the same microbench
TAG=r06 bash scripts/r6/pmc_mfma.sh > "$out/pmc_mfma.log" 2>&1
cp gpurun_out/r06_pmc_verify_gemm_T*_mfma_busy.json "$out/" 2>/dev/null
bash scripts/r6/clock_power_trace.sh "$out/r06_vgemm_clock_power_trace.txt" > /dev/null 2>&1
for T in 256 257 769; do T=$T LOOPS=100 python scripts/vgemm_bench.py - plain 2>&1 | grep "T=" >> "$out/r06_vgemm_bench.txt"; done
# 6. the bench line itself (with secondary configs, tuned-tree line and cpu_baseline)
python bench.py --steps 20 --warmup 5 > "$out/r06_bench_default.json" 2> "$out/bench_default.err"
tail -1 "$out/r06_bench_default.json" | cut -c1-300
ls "$out"
