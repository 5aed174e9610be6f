#!/bin/bash
# plan overrides for the 8B-AWQ draft's gate/up at 16 / 32 rows (graph-replayed 32-layer forward, split schedule)
cd "$(dirname "$0")/../.."
run() { echo "== $1"; UMB_PLAN_OVR="$1" SCHEDS=split T8B=16,32 python scripts/ll_bench.py fwd8bawq 2>&1 | grep forward | sed 's/hugging-quants.*L=32//'; }
run ""
run "28672,4096:1,1,4,0"
run "28672,4096:1,1,3,0"
run "28672,4096:2,1,6,0"
run "28672,4096:2,1,4,0"
run "28672,4096:1,1,2,0"
run "4096,14336:1,16,0,0"
run "4096,14336:1,4,0,4"
run "6144,4096:1,4,0,0"
run "6144,4096:1,5,0,0"
