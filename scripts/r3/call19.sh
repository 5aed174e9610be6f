#!/bin/bash
# kernarg preload A/B: same sources, library built with -mllvm -amdgpu-kernarg-preload-count=16 vs without
cd $GRAFT_REPO_ROOT
o=gpurun_out/c19; mkdir -p $o
PL=$GRAFT_REPO_ROOT/umbrella_amd/csrc/libumbrella_pl.so
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "gemm" > $o/tests.log 2>&1; echo "tests(default) rc=$?" >> $o/tests.log
UMB_LIB_PATH=$PL timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "gemm" > $o/tests_pl.log 2>&1; echo "tests(preload) rc=$?" >> $o/tests_pl.log
tail -2 $o/tests.log $o/tests_pl.log
for rep in 1 2; do
  SCHEDS=split timeout 300 python scripts/ll_bench.py fwd70b 2>&1 | grep "^forward" | sed 's/^/default /' >> $o/fwd.log
  UMB_LIB_PATH=$PL SCHEDS=split timeout 300 python scripts/ll_bench.py fwd70b 2>&1 | grep "^forward" | sed 's/^/preload /' >> $o/fwd.log
  SCHEDS=ll T1B=3 timeout 300 python scripts/ll_bench.py fwd1b 2>&1 | grep "^forward" | sed 's/^/default /' >> $o/fwd.log
  UMB_LIB_PATH=$PL SCHEDS=ll T1B=3 timeout 300 python scripts/ll_bench.py fwd1b 2>&1 | grep "^forward" | sed 's/^/preload /' >> $o/fwd.log
done
cat $o/fwd.log
