#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c15; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
timeout 1200 python -m pytest tests/test_hip_parity_r2.py tests/test_hip_engine.py -m gpu -q -x -k "offload" > $o/tests.log 2>&1; echo "offload tests rc=$?" >> $o/sum.log
for ns in 2 4 6; do
  UMB_OFFLOAD_SLABS=$ns python scripts/bench_configs.py --config c3 --steps 3 --cache-layers 40 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('slabs=$ns ncl=40', d['ms_per_step'], d['host_link_GBs'], d['pure_stream_ms_per_verify'], d['iter_over_stream'])" >> $o/sum.log
done
python scripts/bench_configs.py --config c3 --steps 3 --cache-layers 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ncl=0', d['ms_per_step'], d['host_link_GBs'], d['pure_stream_ms_per_verify'], d['iter_over_stream'])" >> $o/sum.log
cat $o/sum.log; tail -3 $o/tests.log
