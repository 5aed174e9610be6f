#!/bin/bash
# End-of-round-5 evidence, one GPU call; everything lands in gpurun_out/r05f/ (copied into profiles/ by hand, then
# `python scripts/make_profile_numbers.py`).
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r05f
rm -rf "$out"; mkdir -p "$out"
cd "$root"
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
# 1. PMC traffic of the dominant kernel (separate passes)
TAG=r05 bash scripts/pmc_traffic_tag.sh > "$out/pmc_traffic.log" 2>&1
cp gpurun_out/r05_pmc_gemm70b_traffic.json "$out/" 2>/dev/null
# 2. kernel-trace stats of the bench command (headline only)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out/bench_stats" -- \
    python "$root/bench.py" --steps 16 --warmup 2 --no-cpu-baseline --no-secondary > "$out/bench_under_rocprof.log" 2>&1 )
f=$(find "$out/bench_stats" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$out/r05_bench70b_kernel_stats.csv"
t=$(find "$out/bench_stats" -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python scripts/trace_by_shape.py "$t" "$out/r05_bench70b_kernels_by_shape.csv" skinny_gemm ll_gemm gv_kernel draft_chain draft_head reduce_ tree_attn topk accept kv_compact embed rmsnorm argmax
find "$out/bench_stats" -name "*kernel_trace.csv" -delete
# 3. per-kernel times of the graph-replayed 1B forward at 3 rows: persistent chain (default) and the five GEMV launches
SCHEDS=auto T1B=3 bash scripts/prof_fwd.sh fwd1b > "$out/prof_fwd1b.log" 2>&1
cp gpurun_out/prof_fwd1b_by_shape.csv "$out/r05_fwd1b_by_shape.csv" 2>/dev/null
UMB_CHAIN=0 SCHEDS=auto T1B=3 bash scripts/prof_fwd.sh fwd1b > "$out/prof_fwd1b_gemv.log" 2>&1
cp gpurun_out/prof_fwd1b_by_shape.csv "$out/r05_fwd1b_gemv_launches_by_shape.csv" 2>/dev/null
# 4. chain vs GEMV launches, un-profiled, alternated; headline A/B
bash scripts/r5/chain_ab.sh > "$out/r05_chain_ab.txt" 2>&1
bash scripts/r5/bench_ab.sh > "$out/r05_bench_ab_chain.txt" 2>&1
# 5. per-kernel counter tables of the three forwards
L8B=8 TAG=r05 bash scripts/r5/pmc_table.sh fwd70b fwd1b fwd8bawq > "$out/pmc_table.log" 2>&1
cp gpurun_out/r05_pmc_table_*.csv "$out/" 2>/dev/null
# 6. phase timeline of the chain (trace build, if it travelled)
if [ -f build/variants/lib_chain_trace.so ]; then
  for T in 3 1; do UMB_LIB_PATH=build/variants/lib_chain_trace.so timeout 300 python scripts/r5/chain_trace.py $T 8 2>&1 | grep -v "WARNING\|amdgpu.ids"; done > "$out/r05_chain_trace.txt"
fi
# 7. the bench line itself (with secondary configs and cpu_baseline)
python bench.py --steps 20 --warmup 5 > "$out/r05_bench_default.json" 2> "$out/bench_default.err"
tail -1 "$out/r05_bench_default.json" | cut -c1-400
ls -la "$out"
