#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c7; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
for tb in 7 5 3; do
  UMB_TB=$tb timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "gemm_awq or gemm_dense or silu_epilogue or batch_invariance" > $o/tests_tb$tb.log 2>&1
  echo "tb=$tb rc=$?" >> $o/tests.log
done
b() { echo "== $*" >> $o/shapes.log; env "$@" python scripts/ll_bench.py 70b 2>&1 | grep "^70b" | sed 's/| ll.*//' >> $o/shapes.log; }
b ONLY=gu
b ONLY=gu TB_OLD=7
b ONLY=gu TB_OLD=7 UMB_LDS_KB=56
b ONLY=gu TB_OLD=6
b ONLY=qkv
b ONLY=qkv S_OLD=4 TB_OLD=5
b ONLY=qkv S_OLD=8 TB_OLD=5
b ONLY=qkv S_OLD=8
b ONLY=qkv S_OLD=4
b ONLY=qkv S_OLD=4 R_OLD=1
b ONLY=o
b ONLY=o S_OLD=4 R_OLD=1
b ONLY=o S_OLD=8
b ONLY=o S_OLD=8 R_OLD=1
b ONLY=down
b ONLY=down S_OLD=16
b ONLY=down S_OLD=8 R_OLD=1
cat $o/tests.log $o/shapes.log; tail -3 $o/tests_tb7.log
