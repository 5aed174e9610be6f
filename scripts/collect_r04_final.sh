#!/bin/bash
# End-of-round-4 evidence, one GPU call: everything lands in gpurun_out/r04f/ (copied into profiles/ by hand, then
# `python scripts/make_profile_numbers.py`).
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04f
rm -rf "$out"; mkdir -p "$out"
cd "$root"
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
# 1. PMC traffic of the dominant kernel (separate passes)
TAG=r04 bash scripts/pmc_traffic_tag.sh > "$out/pmc_traffic.log" 2>&1
cp gpurun_out/r04_pmc_gemm70b_traffic.json "$out/" 2>/dev/null
# 2. kernel-trace stats of the bench command (headline only)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out/bench_stats" -- \
    python "$root/bench.py" --steps 16 --warmup 2 --no-cpu-baseline --no-secondary > "$out/bench_under_rocprof.log" 2>&1 )
f=$(find "$out/bench_stats" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$out/r04_bench70b_kernel_stats.csv"
t=$(find "$out/bench_stats" -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python scripts/trace_by_shape.py "$t" "$out/r04_bench70b_kernels_by_shape.csv" skinny_gemm ll_gemm gv_kernel reduce_ tree_attn topk accept kv_compact embed rmsnorm argmax
find "$out/bench_stats" -name "*kernel_trace.csv" -delete
# 3. per-kernel times of graph-replayed forwards
SCHEDS=ll T1B=3 bash scripts/prof_fwd.sh fwd1b > "$out/prof_fwd1b.log" 2>&1
cp gpurun_out/prof_fwd1b_by_shape.csv "$out/r04_fwd1b_by_shape.csv" 2>/dev/null
# 4. C3-resident / C4 kernel stats + MFMA-busy of the wide GEMM (scripts/collect_r04.sh)
bash scripts/collect_r04.sh r04w > "$out/collect_r04w.log" 2>&1
cp gpurun_out/r04w/r04w_*.csv gpurun_out/r04w/r04w_*.json gpurun_out/r04w/r04w_vgemm_bench.txt "$out/" 2>/dev/null
# 5. prefill by shape (16 layers)
( cd /tmp && export TMPDIR=/tmp && LAYERS=16 rocprofv3 --kernel-trace --output-format csv -d /tmp/pf -- python "$root/scripts/prefill_profile.py" > "$out/prefill.log" 2>&1 )
t=$(find /tmp/pf -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python scripts/trace_by_shape.py "$t" "$out/r04_prefill_by_shape.csv" verify_gemm skinny reduce_ tree_attn embed rmsnorm sum_splits > /dev/null
# 6. the bench line itself (with secondary configs and cpu_baseline)
python bench.py --steps 20 --warmup 5 > "$out/r04_bench_default.json" 2> "$out/bench_default.err"
tail -1 "$out/r04_bench_default.json" | cut -c1-300
ls -la "$out"
