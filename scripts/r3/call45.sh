#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c45; mkdir -p $o; rm -f $o/*.log
SCHEDS=ll T1B=3 timeout 600 bash scripts/prof_fwd.sh fwd1b > $o/prof.log 2>&1
cp gpurun_out/prof_fwd1b_by_shape.csv $o/gemv_fwd1b_by_shape.csv; head -9 $o/gemv_fwd1b_by_shape.csv
