#!/bin/bash
# Clock / power trace of the wide verify GEMM microbench: rocm-smi sampled every ~100 ms while scripts/vgemm_bench.py loops on
# random activations and again on zero activations (same instruction stream, no operand toggling).  The question it answers
# (VERDICT r5 weak #2): is "power-bound at ~1.8 GHz" a measurement or an assertion?
#   bash scripts/r6/clock_power_trace.sh [outfile]
root=$(cd "$(dirname "$0")/../.." && pwd)
out=${1:-$root/gpurun_out/r06_vgemm_clock_power_trace.txt}
cd "$root"
sample() {   # $1 = label, $2 = pid to follow
  while kill -0 "$2" 2>/dev/null; do
    s=$(rocm-smi --showclocks --showpower --showtemp --json 2>/dev/null | python3 -c '
import json,sys
try:
    d=json.load(sys.stdin); c=d[sorted(d)[0]]
    g=lambda *ks: next((c[k] for k in c for q in ks if q.lower() in k.lower()), "?")
    print("sclk", g("sclk clock speed"), "| mclk", g("mclk clock speed"), "| power_W", g("Socket Graphics Package Power","Average Graphics Package Power","Package Power"), "| temp_C", g("Temperature (Sensor junction)","junction","hotspot"))
except Exception as e:
    print("parse error", e)')
    echo "$(date +%s.%N | cut -c1-14) $1 $s"
    sleep 0.1
  done
}
{
  echo "# rocm-smi while scripts/vgemm_bench.py runs (LOOPS launches per shape); gemm.hip sha256/16 = $(sha256sum umbrella_amd/csrc/gemm.hip | cut -c1-16)"
  echo "# raw json once:"; rocm-smi --showclocks --showpower --showtemp --json 2>/dev/null | head -c 1500; echo; echo "# idle"; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | head -4
  for mode in random zero; do
    for T in 256 769; do
      if [ $mode = zero ]; then export XZERO=1; else unset XZERO; fi
      T=$T LOOPS=${LOOPS:-8000} STAMP=1 python scripts/vgemm_bench.py - "$mode" > /tmp/vg_${mode}_$T.log 2>&1 &
      pid=$!
      sample "$mode/T=$T" $pid | head -400
      wait $pid
      grep "T=\|start\|end" /tmp/vg_${mode}_$T.log
    done
  done
} > "$out" 2>&1
tail -5 "$out"
