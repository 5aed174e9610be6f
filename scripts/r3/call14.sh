#!/bin/bash
# full GPU suite + default bench
cd $GRAFT_REPO_ROOT
o=gpurun_out/c14; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
( time timeout 2400 python -m pytest tests -m gpu -q -x ) > $o/tests_all.log 2>&1; echo "all rc=$?" >> $o/sum.log
( time python bench.py --steps 20 --warmup 5 ) > $o/bench.log 2>&1
grep "^{" $o/bench.log | cut -c1-200 >> $o/sum.log
cat $o/sum.log; tail -5 $o/tests_all.log
