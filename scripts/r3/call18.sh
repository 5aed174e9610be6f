#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c18; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
for d in 0 1 2 3 4 5 7 8 10; do
  UMB_PRE_DBG=$d SCHEDS=split timeout 300 python scripts/ll_bench.py fwd70b 2>&1 | grep "^forward" | sed "s/^/dbg=$d /" >> $o/fwd.log
done
UMB_NO_PRE=1 SCHEDS=split timeout 300 python scripts/ll_bench.py fwd70b 2>&1 | grep "^forward" | sed 's/^/no-pre /' >> $o/fwd.log
cat $o/fwd.log
