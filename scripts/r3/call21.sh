#!/bin/bash
# software-pipelined int4 verify GEMM (verify_gemm_p_kernel): parity + timing of its schedule variants vs the old kernel
cd $GRAFT_REPO_ROOT
o=gpurun_out/c21; mkdir -p $o
for m in 0 1 2 3; do
  UMB_VG_MODE=$m timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "gemm_awq or silu_epilogue" > $o/tests_m$m.log 2>&1; echo "mode $m tests rc=$?" >> $o/tests.log
done
cat $o/tests.log
for T in 256 257 769; do
  UMB_VG_P=0 T=$T python scripts/r3/vg_ablate.py - old >> $o/time.log 2>&1
  for m in 0 1 2 3; do UMB_VG_MODE=$m T=$T python scripts/r3/vg_ablate.py - mode$m >> $o/time.log 2>&1; done
done
grep "layer" $o/time.log
