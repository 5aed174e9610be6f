#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c26; mkdir -p $o
ALT=$GRAFT_REPO_ROOT/umbrella_amd/csrc/libumbrella_noearly.so
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_parity_r2.py -m gpu -q -x -k "gemm or awq or full_size" > $o/tests.log 2>&1; echo "tests rc=$?" >> $o/tests.log
tail -3 $o/tests.log
for rep in 1 2 3; do
  SCHEDS=split timeout 300 python scripts/ll_bench.py fwd70b 2>&1 | grep "^forward" | sed 's/^/early    /; s/hugging-quants.*T=13: //; s/| weights.*//' >> $o/fwd.log
  UMB_LIB_PATH=$ALT SCHEDS=split timeout 300 python scripts/ll_bench.py fwd70b 2>&1 | grep "^forward" | sed 's/^/no-early /; s/hugging-quants.*T=13: //; s/| weights.*//' >> $o/fwd.log
done
for rep in 1 2; do
  SCHEDS=split T8B=13 timeout 300 python scripts/ll_bench.py fwd8b 2>&1 | grep "^forward" | sed 's/^/early    /; s/| weights.*//' >> $o/fwd.log
  UMB_LIB_PATH=$ALT SCHEDS=split T8B=13 timeout 300 python scripts/ll_bench.py fwd8b 2>&1 | grep "^forward" | sed 's/^/no-early /; s/| weights.*//' >> $o/fwd.log
done
cat $o/fwd.log
