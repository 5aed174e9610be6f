#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c25; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
timeout 900 python -m pytest tests/test_gemm_nrm.py -m gpu -q -x > $o/tests.log 2>&1; echo "tests rc=$?" >> $o/tests.log
tail -5 $o/tests.log
for rep in 1 2 3; do
  SCHEDS=split timeout 300 python scripts/ll_bench.py fwd70b 2>&1 | grep "^forward" | sed 's/^/nrm    /' >> $o/fwd.log
  UMB_NO_NRM=1 SCHEDS=split timeout 300 python scripts/ll_bench.py fwd70b 2>&1 | grep "^forward" | sed 's/^/no-nrm /' >> $o/fwd.log
done
cat $o/fwd.log
SCHEDS=split timeout 600 bash scripts/prof_fwd.sh fwd70b > $o/prof.log 2>&1
cp gpurun_out/prof_fwd70b_by_shape.csv $o/nrm_fwd70b_by_shape.csv; head -14 $o/nrm_fwd70b_by_shape.csv
