#!/bin/bash
# attention: query rows loaded before the prefix length is read -- A/B on one box
cd $GRAFT_REPO_ROOT
o=gpurun_out/c47; mkdir -p $o; rm -f $o/*.log
OLD=$GRAFT_REPO_ROOT/umbrella_amd/csrc/libumbrella_before.so
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "attention or attn" > $o/tests.log 2>&1; echo "tests rc=$?" >> $o/tests.log; tail -2 $o/tests.log
for rep in 1 2 3; do
  SCHEDS=split timeout 300 python scripts/ll_bench.py fwd70b 2>&1 | grep "^forward" | sed 's/^/new /; s/hugging-quants.*T=13: //; s/| weights.*//' >> $o/fwd.log
  UMB_LIB_PATH=$OLD SCHEDS=split timeout 300 python scripts/ll_bench.py fwd70b 2>&1 | grep "^forward" | sed 's/^/old /; s/hugging-quants.*T=13: //; s/| weights.*//' >> $o/fwd.log
done
cat $o/fwd.log
