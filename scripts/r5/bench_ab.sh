#!/bin/bash
# Headline iteration, persistent chain on / off, alternated on ONE box (boxes differ by +-1.5 %: DESIGN.md section 5).
root=$(cd "$(dirname "$0")/../.." && pwd); cd "$root"
for rep in 1 2 3; do
  for c in 0 1; do
    line=$(UMB_CHAIN=$c python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1)
    echo "UMB_CHAIN=$c rep $rep: $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms_per_step", d["ms_per_step"], "tok/s", d["value"], "raw", d.get("value_raw_draft"))')"
  done
done
