#!/bin/bash
# waves per (kv head, query tile) x query tiles per block of the single-launch tree attention on the fragment-ordered cache
cd "$(dirname "$0")/../.."
run() { python scripts/attn_bench.py "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['us_per_launch(attn+combine)'])"; }
for T in 129 257 385 505 769; do
  for nw in 1 2 4 8; do for nq in 1 2; do
    [ $nw -gt 2 ] && [ $nq -eq 2 ] && continue
    echo "T=$T prefix=128 nw=$nw nq=$nq: $(UMB_ATTN_NW=$nw UMB_ATTN_NQ=$nq run --T $T --prefix 128 --Lmax 4096)"
  done; done
  echo "T=$T default: $(run --T $T --prefix 128 --Lmax 4096)"
done
for nw in 1 2; do for nq in 1 2; do echo "causal 1024@1024 nw=$nw nq=$nq: $(UMB_ATTN_NW=$nw UMB_ATTN_NQ=$nq run --T 1024 --prefix 1024 --Lmax 4096 --causal)"; done; done
for T in 31 64; do for nw in 2 4 8; do echo "T=$T prefix=300 nw=$nw: $(UMB_ATTN_NW=$nw run --T $T --prefix 300 --Lmax 2048)"; done; done
