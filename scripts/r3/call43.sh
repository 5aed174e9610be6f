#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c43; mkdir -p $o; rm -f $o/*.log
timeout 2000 python -m pytest tests -m gpu -q -x > $o/tests.log 2>&1; echo "tests rc=$?" >> $o/tests.log
tail -n 3 $o/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1; tail -1 $o/smoke.log
python bench.py --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err; tail -1 $o/bench.json | cut -c1-300
