#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c36; mkdir -p $o
GV=2 timeout 600 python scripts/probe/gemv_probe.py > $o/probe2.log 2>&1; grep -v amdgpu $o/probe2.log | tail -8
GV=2 T=1 timeout 600 python scripts/probe/gemv_probe.py >> $o/probe2.log 2>&1; grep -v amdgpu $o/probe2.log | tail -6
