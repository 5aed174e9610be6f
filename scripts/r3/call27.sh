#!/bin/bash
# two-phase (ping-pong) verify GEMM: parity + timing vs the two-blocks-per-CU kernel
cd $GRAFT_REPO_ROOT
o=gpurun_out/c27; mkdir -p $o
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_parity_r2.py -m gpu -q -x -k "gemm or silu_epilogue or prefill or wide" > $o/tests.log 2>&1; echo "tests rc=$?" >> $o/tests.log
tail -3 $o/tests.log
for T in 256 257 385 769 1024; do
  UMB_VG_PP=0 T=$T python scripts/r3/vg_ablate.py - old >> $o/time.log 2>&1
  T=$T python scripts/r3/vg_ablate.py - pingpong >> $o/time.log 2>&1
done
grep "layer" $o/time.log
