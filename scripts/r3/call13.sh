#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c13; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
timeout 1200 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "gemm_awq or batch_invariance or full_size" > $o/tests_ops.log 2>&1; echo "ops rc=$?" >> $o/ab.log
for v in UMB_NO_SPLIT8=1 A=1 UMB_NO_SPLIT8=1 A=1; do
  env $v python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'], d['roofline']['frac'], {k:v['us'] for k,v in d['roofline']['layer_gemms'].items()})" >> $o/ab.log
done
for v in UMB_NO_SPLIT8=1 A=1 UMB_NO_SPLIT8=1 A=1; do
  env $v SCHEDS=split python scripts/ll_bench.py fwd70b 2>&1 | grep "^forward" | sed "s/$/ $v/" >> $o/ab.log
done
cat $o/ab.log
