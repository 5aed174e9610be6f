#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c31; mkdir -p $o; rm -f $o/*.log
for d in "" "-DUMB_VG_NODEQ" "-DUMB_PP_LOADS_X" "-DUMB_PP_LOADS_X -DUMB_VG_NODEQ"; do
  echo "#### VG_DEFS=$d" >> $o/trace.log
  VG_DEFS="$d" timeout 900 python scripts/r3/vg_trace.py 2>&1 | grep -v "amdgpu.ids\|warning" | tail -9 >> $o/trace.log
done
cat $o/trace.log
