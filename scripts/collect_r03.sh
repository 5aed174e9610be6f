#!/bin/bash
# End-of-round evidence, one GPU call: everything lands in gpurun_out/r03/ (copy what is to be judged into profiles/).
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r03
rm -rf "$out"; mkdir -p "$out"
cd "$root"
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
# 1. PMC traffic of the dominant kernel (separate passes)
bash scripts/pmc_traffic_r03.sh > "$out/pmc_traffic.log" 2>&1
cp gpurun_out/r03_pmc_gemm70b_traffic.json "$out/" 2>/dev/null
# 2. kernel-trace stats of the bench command
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out/bench_stats" -- \
    python "$root/bench.py" --steps 16 --warmup 2 --no-cpu-baseline --no-secondary > "$out/bench_under_rocprof.log" 2>&1 )
f=$(find "$out/bench_stats" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$out/r03_bench70b_kernel_stats.csv"
t=$(find "$out/bench_stats" -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python scripts/trace_by_shape.py "$t" "$out/r03_bench70b_kernels_by_shape.csv" skinny_gemm ll_gemm reduce_ tree_attn topk accept kv_compact embed rmsnorm argmax
find "$out/bench_stats" -name "*kernel_trace.csv" -delete
# 3. per-kernel times of graph-replayed forwards
SCHEDS=split bash scripts/prof_fwd.sh fwd70b > "$out/prof_fwd70b.log" 2>&1
cp gpurun_out/prof_fwd70b_by_shape.csv "$out/r03_fwd70b_by_shape.csv" 2>/dev/null
SCHEDS=ll T1B=3 bash scripts/prof_fwd.sh fwd1b > "$out/prof_fwd1b.log" 2>&1
cp gpurun_out/prof_fwd1b_by_shape.csv "$out/r03_fwd1b_by_shape.csv" 2>/dev/null
# 4. SQ counters of the gate/up launch
bash scripts/pmc_gu.sh > "$out/r03_pmc_gu.txt" 2>&1
# 5. per-block phase stamps of the four layer GEMMs
python scripts/r3/gemm_trace.py 2>&1 | grep -v WARNING | grep -v "^\[" > "$out/r03_gemm_phase_trace.txt"
# 6. the bench line itself (with secondary configs and cpu_baseline)
python bench.py --steps 20 --warmup 5 > "$out/bench_default.json" 2> "$out/bench_default.err"
tail -1 "$out/bench_default.json" | cut -c1-400
ls -la "$out"
