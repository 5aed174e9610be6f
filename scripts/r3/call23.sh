#!/bin/bash
# in-graph A/B of qkv / o plans on the 70B-AWQ 16-layer forward (T = 13)
cd $GRAFT_REPO_ROOT
o=gpurun_out/c23; mkdir -p $o
run() { env "$@" SCHEDS=split timeout 300 python scripts/ll_bench.py fwd70b 2>&1 | grep "^forward" | sed "s/^/$* | /" | sed 's/hugging-quants.*T=13: //; s/| weights.*//' >> $o/fwd.log; }
run A=base
run UMB_PLAN_OVR="10240,8192:2,4,5,0"
run UMB_PLAN_OVR="10240,8192:2,8,5,0"
run UMB_PLAN_OVR="10240,8192:2,4,138,0"
run UMB_PLAN_OVR="10240,8192:2,8,138,0"
run UMB_PLAN_OVR="10240,8192:1,4,5,0"
run UMB_PLAN_OVR="10240,8192:1,8,5,0"
run A=base
run UMB_PLAN_OVR="8192,8192:1,4,0,4"
run UMB_PLAN_OVR="8192,8192:2,8,144,8"
run UMB_PLAN_OVR="8192,8192:2,8,144,4"
run UMB_PLAN_OVR="8192,8192:1,8,136,8"
run UMB_PLAN_OVR="8192,8192:2,8,0,8"
run A=base
run UMB_PLAN_OVR="8192,28672:2,8,144,8"
run UMB_PLAN_OVR="8192,28672:2,16,144,16"
run UMB_PLAN_OVR="8192,28672:1,8,0,8"
cat $o/fwd.log
