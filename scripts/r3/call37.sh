#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c37; mkdir -p $o
timeout 900 python -m pytest tests/test_gemv.py -m gpu -q -x > $o/tests.log 2>&1; echo "tests rc=$?" >> $o/tests.log
tail -25 $o/tests.log
