#!/bin/bash
# BASELINE config 4: the 8B-AWQ draft (T = 32 per level) on the split vs the low-latency schedule
cd $GRAFT_REPO_ROOT
o=gpurun_out/c48; mkdir -p $o; rm -f $o/*.log
for s in auto ll; do
  UMB_SCHED=$s timeout 900 python scripts/bench_configs.py --config c4 --steps 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 sched=$s', d['ms_per_step'])" >> $o/c4.log
done
cat $o/c4.log
