#!/bin/bash
# round 3, call 4: the reworked tensor-parallel / layer-sharded paths
cd $GRAFT_REPO_ROOT
o=gpurun_out/c4; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
timeout 1500 python -m pytest tests/test_tensor_parallel.py tests/test_parallel_hip.py tests/test_bench_launch.py -m gpu -q -x > $o/tests.log 2>&1
echo "rc=$?" >> $o/tests.log
timeout 1200 python -m pytest tests/test_hip_engine.py -m gpu -q -x -k "pipelin or stage" > $o/tests2.log 2>&1
echo "rc=$?" >> $o/tests2.log
timeout 600 python bench.py --parallel tp --steps 10 --warmup 3 > $o/bench_tp1.log 2>&1
timeout 600 python bench.py --parallel pp --steps 10 --warmup 3 > $o/bench_pp1.log 2>&1
tail -25 $o/tests.log; tail -5 $o/tests2.log; grep "^{" $o/bench_tp1.log | cut -c1-300; grep "^{" $o/bench_pp1.log | cut -c1-300; tail -3 $o/bench_tp1.log | cut -c1-300
