#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c46; mkdir -p $o; rm -f $o/*.log
UMB_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 8 --warmup 2 --workload tiny > $o/n2.json 2> $o/n2.err; echo "rc=$?" >> $o/n2.err
tail -1 $o/n2.json | cut -c1-1500; tail -3 $o/n2.err
