#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c8; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
timeout 1200 python -m pytest tests/test_hip_ops.py tests/test_lowlat.py -m gpu -q -x -k "gemm or awq" > $o/tests_ops.log 2>&1; echo "ops rc=$?" >> $o/tests.log
timeout 1200 python -m pytest tests/test_hip_parity_r2.py tests/test_full_depth.py -m gpu -q -x > $o/tests_par.log 2>&1; echo "parity rc=$?" >> $o/tests.log
for v in UMB_PLAN2_OFF=1 A=1 UMB_PLAN2_OFF=1 A=1; do
  env $v SCHEDS=split python scripts/ll_bench.py fwd70b 2>&1 | grep "^forward" | sed "s/$/ $v/" >> $o/ab.log
done
for v in UMB_PLAN2_OFF=1 A=1 UMB_PLAN2_OFF=1 A=1; do
  env $v python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['layer_gemms'])" >> $o/ab.log
done
cat $o/tests.log $o/ab.log; tail -3 $o/tests_ops.log $o/tests_par.log
