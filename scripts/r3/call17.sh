#!/bin/bash
# producer-finishing GEMM (umb_gemm_pre): parity tests, then the 16-layer 70B-AWQ forward with and without it
cd $GRAFT_REPO_ROOT
o=gpurun_out/c17; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
timeout 900 python -m pytest tests/test_gemm_pre.py -m gpu -q -x > $o/tests.log 2>&1; echo "tests rc=$?" >> $o/tests.log
tail -5 $o/tests.log
for rep in 1 2; do
  SCHEDS=split timeout 300 python scripts/ll_bench.py fwd70b 2>&1 | grep "^forward" | sed 's/^/pre    /' >> $o/fwd.log
  UMB_NO_PRE=1 SCHEDS=split timeout 300 python scripts/ll_bench.py fwd70b 2>&1 | grep "^forward" | sed 's/^/no-pre /' >> $o/fwd.log
done
cat $o/fwd.log
SCHEDS=split timeout 600 bash scripts/prof_fwd.sh fwd70b > $o/prof.log 2>&1
cp gpurun_out/prof_fwd70b_by_shape.csv $o/pre_fwd70b_by_shape.csv
cat $o/pre_fwd70b_by_shape.csv | head -20
