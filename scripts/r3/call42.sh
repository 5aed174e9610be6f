#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c42; mkdir -p $o; rm -f $o/*.log
timeout 900 python -m pytest tests/test_gemv.py -m gpu -q -x > $o/tests.log 2>&1; echo "tests rc=$?" >> $o/tests.log
tail -4 $o/tests.log
for rep in 1 2; do
  SCHEDS=ll T1B=5,8 timeout 300 python scripts/ll_bench.py fwd1b 2>&1 | grep "^forward" | sed 's/^/gemv    /; s/| weights.*//' >> $o/fwd.log
  UMB_GEMV=0 SCHEDS=ll T1B=5,8 timeout 300 python scripts/ll_bench.py fwd1b 2>&1 | grep "^forward" | sed 's/^/no-gemv /; s/| weights.*//' >> $o/fwd.log
done
cat $o/fwd.log
for rep in 1 2; do
  python scripts/bench_configs.py --config c2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 gemv   ', d['ms_per_step'])" >> $o/c2.log
  UMB_GEMV=0 python scripts/bench_configs.py --config c2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 no-gemv', d['ms_per_step'])" >> $o/c2.log
done
cat $o/c2.log
