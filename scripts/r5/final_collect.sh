root=/root/repo; out=$root/gpurun_out/r05g; rm -rf $out; mkdir -p $out; cd $root
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out/bench_stats" -- python "$root/bench.py" --steps 16 --warmup 2 --no-cpu-baseline --no-secondary > "$out/bench_under_rocprof.log" 2>&1 )
f=$(find "$out/bench_stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/r05_bench70b_kernel_stats.csv"
t=$(find "$out/bench_stats" -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python scripts/trace_by_shape.py "$t" "$out/r05_bench70b_kernels_by_shape.csv" skinny_gemm ll_gemm gv_kernel draft_chain draft_head reduce_ tree_attn topk accept kv_compact embed rmsnorm argmax
find "$out/bench_stats" -name "*kernel_trace.csv" -delete
python bench.py --steps 20 --warmup 5 > "$out/r05_bench_default.json" 2> "$out/bench_default.err"
tail -1 "$out/r05_bench_default.json" | cut -c1-200
bash scripts/r5/step_timeline.sh 2>&1 | tail -30
