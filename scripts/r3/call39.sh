#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c39; mkdir -p $o; rm -f $o/*.log
timeout 900 python -m pytest tests/test_gemv.py -m gpu -q -x -k model_logits > $o/tests.log 2>&1; echo "tests rc=$?" >> $o/tests.log
tail -5 $o/tests.log
SCHEDS=ll T1B=3 timeout 600 bash scripts/prof_fwd.sh fwd1b > $o/prof.log 2>&1
cp gpurun_out/prof_fwd1b_by_shape.csv $o/gemv_fwd1b_by_shape.csv; head -16 $o/gemv_fwd1b_by_shape.csv
