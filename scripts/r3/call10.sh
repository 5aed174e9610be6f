#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c10; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
timeout 1200 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "gemm_awq or silu_epilogue or full_size" > $o/tests_ops.log 2>&1; echo "ops rc=$?" >> $o/tests.log
timeout 1200 python -m pytest tests/test_full_depth.py -m gpu -q -x -k 70b > $o/tests_fd.log 2>&1; echo "full depth rc=$?" >> $o/tests.log
for v in UMB_PLAN2_OFF=1 UMB_NO_W8=1 A=1 UMB_PLAN2_OFF=1 UMB_NO_W8=1 A=1; do
  env $v SCHEDS=split python scripts/ll_bench.py fwd70b 2>&1 | grep "^forward" | sed "s/$/ $v/" >> $o/ab.log
done
for v in UMB_PLAN2_OFF=1 UMB_NO_W8=1 A=1; do
  env $v python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['layer_gemms']['gu'])" >> $o/ab.log
done
python scripts/r3/gemm_trace.py 2>&1 | grep -v WARNING | grep -A11 "^== gu" | head -14 >> $o/ab.log
cat $o/tests.log $o/ab.log; tail -3 $o/tests_ops.log $o/tests_fd.log
