#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c41; mkdir -p $o; rm -f $o/*.log
for rep in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gemv   ', d['ms_per_step'])" >> $o/ab.log
  UMB_GEMV=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no-gemv', d['ms_per_step'])" >> $o/ab.log
done
cat $o/ab.log
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$o/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/$o/prof.log 2>&1 )
f=$(find $o/prof -name "*kernel_stats.csv" | head -1); head -14 $f | cut -c1-150
find $o/prof -name "*kernel_trace.csv" -delete
