#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c29; mkdir -p $o; rm -f $o/*.log
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_parity_r2.py -m gpu -q -x -k "gemm or silu_epilogue or prefill or wide" > $o/tests.log 2>&1; echo "tests rc=$?" >> $o/tests.log
tail -2 $o/tests.log
for T in 256 257 769; do
  UMB_VG_PP=0 T=$T python scripts/r3/vg_ablate.py - old >> $o/time.log 2>&1
  T=$T python scripts/r3/vg_ablate.py - pingpong >> $o/time.log 2>&1
done
grep "layer" $o/time.log
timeout 900 python scripts/r3/vg_trace.py > $o/trace.log 2>&1; grep -v "amdgpu.ids\|warning" $o/trace.log | tail -9
