#!/bin/bash
# round 3, call 3: where the 1B draft layer's time goes (per-shape microbench + graph-replayed forward, knob variants)
cd $GRAFT_REPO_ROOT
o=gpurun_out/c3; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
python scripts/ll_bench.py 1b 2>&1 | grep -v WARNING > $o/shapes_1b.log
T=1 python scripts/ll_bench.py 1b 2>&1 | grep -v WARNING > $o/shapes_1b_T1.log
for v in "" "UMB_LL_GU=ll" "UMB_LL_GU=shared" "UMB_LL_PF=2" "UMB_LL_WK=4" "UMB_LL_NW=4"; do
  echo "== $v" >> $o/fwd1b.log
  env $v SCHEDS=ll T1B=1,3 python scripts/ll_bench.py fwd1b 2>&1 | grep "^forward" >> $o/fwd1b.log
done
python scripts/ll_bench.py stream 2>&1 | grep "stream read" > $o/stream.log
STREAM_MB=8,13,33 python scripts/ll_bench.py stream 2>&1 | grep "stream read" >> $o/stream.log
cd /tmp && export TMPDIR=/tmp
d=$GRAFT_REPO_ROOT/$o/prof; rm -rf $d; mkdir -p $d
SCHEDS=ll T1B=3 rocprofv3 --kernel-trace --output-format csv -d $d -- python $GRAFT_REPO_ROOT/scripts/ll_bench.py fwd1b > $d/run.log 2>&1
t=$(find $d -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/scripts/trace_by_shape.py $t $GRAFT_REPO_ROOT/$o/fwd1b_by_shape.csv > /dev/null
rm -rf $d
cd $GRAFT_REPO_ROOT
cat $o/shapes_1b.log $o/fwd1b.log $o/stream.log; head -12 $o/fwd1b_by_shape.csv
