#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c24; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x > $o/tests.log 2>&1; echo "tests rc=$?" >> $o/tests.log
tail -n 3 $o/tests.log
bash scripts/collect_r03.sh > $o/collect.log 2>&1; tail -n 25 $o/collect.log
