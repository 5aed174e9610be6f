#!/bin/bash
# round 3, call 1: validate the new tests + the new bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c1
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c1/build.log 2>&1
timeout 1500 python -m pytest tests/test_checkpoint_loading.py tests/test_full_depth.py tests/test_bench_launch.py tests/test_parallel_hip.py tests/test_tensor_parallel.py -m gpu -x -q -s > gpurun_out/c1/new_tests.log 2>&1
echo "rc=$?" >> gpurun_out/c1/new_tests.log
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "rope_inplace or kv_append or h2d" > gpurun_out/c1/ops_tests.log 2>&1
echo "rc=$?" >> gpurun_out/c1/ops_tests.log
timeout 900 python -m pytest tests/test_hip_parity_r2.py -m gpu -x -q > gpurun_out/c1/parity_tests.log 2>&1
echo "rc=$?" >> gpurun_out/c1/parity_tests.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/c1/bench.log 2>&1
echo "rc=$?" >> gpurun_out/c1/bench.log
tail -3 gpurun_out/c1/*.log
