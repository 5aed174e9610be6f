#!/bin/bash
# kernarg preload A/B (all hot kernels' arguments ordered for it): default library vs one built without the flag
cd $GRAFT_REPO_ROOT
o=gpurun_out/c20; mkdir -p $o
NOPL=$GRAFT_REPO_ROOT/umbrella_amd/csrc/libumbrella_nopl.so
for rep in 1 2; do
  SCHEDS=split timeout 300 python scripts/ll_bench.py fwd70b 2>&1 | grep "^forward" | sed 's/^/preload    /' >> $o/fwd.log
  UMB_LIB_PATH=$NOPL SCHEDS=split timeout 300 python scripts/ll_bench.py fwd70b 2>&1 | grep "^forward" | sed 's/^/no-preload /' >> $o/fwd.log
  SCHEDS=ll T1B=3 timeout 300 python scripts/ll_bench.py fwd1b 2>&1 | grep "^forward" | sed 's/^/preload    /' >> $o/fwd.log
  UMB_LIB_PATH=$NOPL SCHEDS=ll T1B=3 timeout 300 python scripts/ll_bench.py fwd1b 2>&1 | grep "^forward" | sed 's/^/no-preload /' >> $o/fwd.log
done
cat $o/fwd.log
timeout 1500 python -m pytest tests -m gpu -q -x > $o/tests.log 2>&1; echo "tests rc=$?" >> $o/tests.log
tail -n 4 $o/tests.log
python bench.py --steps 20 --warmup 5 --no-secondary > $o/bench.json 2> $o/bench.err; cat $o/bench.json
