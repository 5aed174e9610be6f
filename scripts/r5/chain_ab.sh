#!/bin/bash
# Persistent chain vs the five GEMV launches: graph-replayed 16-layer 1B forwards at T = 1, 2, 3 on ONE box, alternated.
# Extra environment for the chain leg: CHAIN_ENV="UMB_CHAIN_HINT=0"
root=$(cd "$(dirname "$0")/../.." && pwd)
cd "$root"
for rep in 1 2; do
  for c in 0 1; do
    echo "== UMB_CHAIN=$c (rep $rep) $CHAIN_ENV"
    env UMB_CHAIN=$c $CHAIN_ENV SCHEDS=auto T1B=${T1B:-1,2,3} timeout 300 python scripts/ll_bench.py fwd1b 2>&1 | grep -E "^forward|Error|error"
  done
done
