#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c22; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
timeout 600 python scripts/probe/gemv_probe.py > $o/probe.log 2>&1; tail -8 $o/probe.log
T=1 timeout 600 python scripts/probe/gemv_probe.py >> $o/probe.log 2>&1; tail -6 $o/probe.log
for rep in 1 2; do SCHEDS=split timeout 300 python scripts/ll_bench.py fwd70b 2>&1 | grep "^forward" >> $o/fwd.log; done; cat $o/fwd.log
