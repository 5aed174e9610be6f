#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c44; mkdir -p $o; rm -f $o/*.log
SCHEDS=ll T1B=1,2,3,4 timeout 300 python scripts/ll_bench.py fwd1b 2>&1 | grep "^forward" | sed 's/^/gemv    /; s/| weights.*//' >> $o/fwd.log
UMB_GEMV=0 SCHEDS=ll T1B=1,2,3,4 timeout 300 python scripts/ll_bench.py fwd1b 2>&1 | grep "^forward" | sed 's/^/no-gemv /; s/| weights.*//' >> $o/fwd.log
cat $o/fwd.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gemv   ', d['ms_per_step'])"
UMB_GEMV=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no-gemv', d['ms_per_step'])"
