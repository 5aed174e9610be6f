#!/bin/bash
# A/B of wide-GEMM variant libraries on one box: bash scripts/r6/vgw_ab.sh "<variant names>" "<T list>"
root=$(cd "$(dirname "$0")/../.." && pwd); cd "$root"
for T in ${2:-256 257 769}; do
  for v in default $1; do
    lib=-; [ $v != default ] && lib=build/variants/lib_$v.so
    T=$T LOOPS=${LOOPS:-100} python scripts/vgemm_bench.py $lib $v 2>&1 | grep "T="
    [ -n "$ZERO" ] && XZERO=1 T=$T LOOPS=${LOOPS:-100} python scripts/vgemm_bench.py $lib ${v}0 2>&1 | grep "T="
  done
done
