#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c38; mkdir -p $o; rm -f $o/*.log
timeout 900 python -m pytest tests/test_gemv.py -m gpu -q -x > $o/tests.log 2>&1; echo "tests rc=$?" >> $o/tests.log
tail -12 $o/tests.log
for rep in 1 2; do
  SCHEDS=ll T1B=1,3 timeout 300 python scripts/ll_bench.py fwd1b 2>&1 | grep "^forward" | sed 's/^/gemv    /; s/| weights.*//' >> $o/fwd.log
  UMB_GEMV=0 SCHEDS=ll T1B=1,3 timeout 300 python scripts/ll_bench.py fwd1b 2>&1 | grep "^forward" | sed 's/^/no-gemv /; s/| weights.*//' >> $o/fwd.log
done
cat $o/fwd.log
