#!/bin/bash
# A/B of the wide-GEMM kernel generations on one box via UMB_VGW (0 = gemm.hip's, 1 = vgemm.hip VER 1, 2 = VER 2)
root=$(cd "$(dirname "$0")/../.." && pwd); cd "$root"
for T in ${2:-256 257 769}; do
  for v in ${1:-2 1 0}; do
    UMB_VGW=$v T=$T LOOPS=${LOOPS:-100} python scripts/vgemm_bench.py - vgw$v 2>&1 | grep "T="
    [ -n "$ZERO" ] && UMB_VGW=$v XZERO=1 T=$T LOOPS=${LOOPS:-100} python scripts/vgemm_bench.py - vgw${v}z 2>&1 | grep "T="
  done
done
