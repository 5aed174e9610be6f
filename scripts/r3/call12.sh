#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c12; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
for v in A=1 UMB_W8_512=1 A=1 UMB_W8_512=1; do
  env $v python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'], d['roofline']['frac'], {k:v['us'] for k,v in d['roofline']['layer_gemms'].items()})" >> $o/ab.log
done
cat $o/ab.log
