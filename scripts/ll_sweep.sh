#!/bin/bash
# A/B sweep of the low-latency GEMM's launch geometry on the 70B-AWQ layer shapes (one process per setting: the knobs
# are read once).  Usage: bash scripts/ll_sweep.sh > gpurun_out/ll_sweep.log
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" python scripts/ll_bench.py 70b 2>&1 | grep -E "^70b" | sed 's/.*| ll/   ll/'; }
run UMB_LL_LB=4
run UMB_LL_LB=2
run UMB_LL_LB=2 UMB_LL_WK=8
run UMB_LL_LB=4 UMB_LL_WK=8
run UMB_LL_LB=2 UMB_LL_WK=4
run UMB_LL_LB=2 UMB_LL_NW=4
run UMB_LL_LB=2 UMB_LL_NW=4 UMB_LL_WK=4
run UMB_LL_LB=4 UMB_LL_NW=4 UMB_LL_WK=4
run UMB_LL_LB=2 UMB_LL_NW=4 UMB_LL_WK=2
