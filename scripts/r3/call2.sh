#!/bin/bash
# round 3, call 2: re-run the tests that failed / did not run in call 1; A/B the L2 warm-up helpers (UMB_PF_MB)
cd $GRAFT_REPO_ROOT
o=gpurun_out/c2; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
timeout 1500 python -m pytest tests/test_full_depth.py tests/test_bench_launch.py tests/test_parallel_hip.py tests/test_tensor_parallel.py -m gpu -q -s > $o/new_tests.log 2>&1
echo "rc=$?" >> $o/new_tests.log
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -k "rope_inplace or kv_append or h2d or qkv_rope or reduce_residual" > $o/ops_tests.log 2>&1
echo "rc=$?" >> $o/ops_tests.log
# A/B: 16-layer 70B-AWQ forward at T = 13 under hipGraph replay, split schedule, by warm-up budget
for mb in 0 8 16 20 24 28 0 20; do
  UMB_PF_MB=$mb SCHEDS=split python scripts/ll_bench.py fwd70b 2>&1 | grep "^forward" >> $o/ab_fwd70b.log
done
# the same on the 8B (dense, bf16, T = 31) -- split schedule
for mb in 0 20; do
  UMB_PF_MB=$mb SCHEDS=split python scripts/ll_bench.py fwd8b 2>&1 | grep "^forward" >> $o/ab_fwd8b.log
done
# whole iteration
for mb in 0 20 0 20; do
  UMB_PF_MB=$mb python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('PF_MB=$mb', d['ms_per_step'], d['value'], d['roofline']['layer_gemms'])" >> $o/ab_bench.log
done
# per-kernel view, on and off
cd /tmp && export TMPDIR=/tmp
for mb in 0 20; do
  d=$GRAFT_REPO_ROOT/$o/prof_pf$mb; rm -rf $d; mkdir -p $d
  UMB_PF_MB=$mb SCHEDS=split rocprofv3 --kernel-trace --output-format csv -d $d -- python $GRAFT_REPO_ROOT/scripts/ll_bench.py fwd70b > $d/run.log 2>&1
  t=$(find $d -name "*kernel_trace.csv" | head -1)
  python $GRAFT_REPO_ROOT/scripts/trace_by_shape.py $t $GRAFT_REPO_ROOT/$o/fwd70b_by_shape_pf$mb.csv > /dev/null
  rm -rf $d
done
cd $GRAFT_REPO_ROOT
tail -4 $o/new_tests.log $o/ops_tests.log; cat $o/ab_fwd70b.log $o/ab_fwd8b.log $o/ab_bench.log
