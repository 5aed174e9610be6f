#!/bin/bash
# plan overrides for the 70B-AWQ verify layer at 13 rows (graph-replayed 16-layer forward, split schedule): 8-wave blocks for down / qkv / o
cd "$(dirname "$0")/../.."
run() { echo "== $1: $(UMB_PLAN_OVR="$1" SCHEDS=split T70=13 python scripts/ll_bench.py fwd70b 2>&1 | grep forward | sed 's/hugging-quants.*L=16//' | cut -c1-40)"; }
run ""
run "8192,28672:2,8,144,0"
run "10240,8192:2,4,138,0"
run "10240,8192:2,8,138,0"
run "8192,8192:2,8,144,8"
run "8192,8192:2,4,136,4"
run ""
run "8192,28672:2,16,144,0"
run "8192,28672:2,8,136,0"
