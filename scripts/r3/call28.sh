#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c28; mkdir -p $o
timeout 900 python scripts/r3/vg_trace.py > $o/trace.log 2>&1; grep -v "amdgpu.ids\|warning" $o/trace.log | tail -30
