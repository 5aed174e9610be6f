#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c49; mkdir -p $o; rm -rf $o/*
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$o/prof -- python $GRAFT_REPO_ROOT/scripts/bench_configs.py --config c4 --steps 4 > $GRAFT_REPO_ROOT/$o/run.log 2>&1 )
t=$(find $o/prof -name "*kernel_trace.csv" | head -1)
python scripts/trace_by_shape.py $t $o/c4_by_shape.csv > /dev/null
find $o/prof -name "*kernel_trace.csv" -delete
tail -1 $o/run.log | cut -c1-300
head -30 $o/c4_by_shape.csv | cut -c1-130
