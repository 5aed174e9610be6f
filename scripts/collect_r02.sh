#!/bin/bash
# End-of-round evidence, one GPU call: everything lands in gpurun_out/r02/ (copy what is to be judged into profiles/).
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r02
rm -rf "$out"; mkdir -p "$out"
cd "$root"
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
# 1. PMC traffic of the dominant kernel (separate passes)
bash scripts/pmc_traffic.sh > "$out/pmc_traffic.log" 2>&1
cp gpurun_out/r02_pmc_gemm70b_traffic.json "$out/" 2>/dev/null
# 2. kernel-trace stats of the bench command
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out/bench_stats" -- \
    python "$root/bench.py" --steps 16 --warmup 2 --no-cpu-baseline > "$out/bench_under_rocprof.log" 2>&1 )
f=$(find "$out/bench_stats" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$out/r02_bench70b_kernel_stats.csv"
t=$(find "$out/bench_stats" -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python scripts/trace_by_shape.py "$t" "$out/r02_bench70b_kernels_by_shape.csv" skinny_gemm ll_gemm reduce_ tree_attn topk accept kv_compact embed rmsnorm argmax
find "$out/bench_stats" -name "*kernel_trace.csv" -delete
# 3. per-kernel times of graph-replayed forwards, both schedules
bash scripts/prof_fwd.sh fwd70b > "$out/prof_fwd70b.log" 2>&1
cp gpurun_out/prof_fwd70b_by_shape.csv "$out/r02_fwd70b_by_shape_final.csv" 2>/dev/null
# 4. SQ counters of the gate/up launch
bash scripts/pmc_gu.sh > "$out/r02_pmc_gu_final.txt" 2>&1
# 5. stream rate + lab probes
{ echo "== python scripts/ll_bench.py stream 70b"; python scripts/ll_bench.py stream 70b 2>&1 | grep -E "GB/s";
  echo "== python scripts/probe/run_stream_probe.py"; python scripts/probe/run_stream_probe.py 2>&1 | grep "GB/s";
  echo "== python scripts/probe/run_gemm_probe.py"; python scripts/probe/run_gemm_probe.py 2>&1 | grep "GB/s";
  echo "== python scripts/probe/run_valu_probe.py"; python scripts/probe/run_valu_probe.py 2>&1 | grep "ns/instr"; } > "$out/r02_stream_and_probes.txt"
# 6. the bench line itself (with cpu_baseline)
python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
tail -1 "$out/bench_default.json" | cut -c1-400
ls -la "$out"
