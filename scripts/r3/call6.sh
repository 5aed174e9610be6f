#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c6; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
for v in "" "UMB_CB=4" "UMB_LDS_KB=56" "UMB_LDS_KB=56 UMB_CB=4" "UMB_LDS_KB=84" "UMB_LDS_KB=42"; do
  echo "== $v" >> $o/shapes70b.log
  env $v python scripts/ll_bench.py 70b 2>&1 | grep "^70b" | sed 's/| ll.*//' >> $o/shapes70b.log
done
for v in "" "UMB_LDS_KB=56" "UMB_LDS_KB=56 UMB_CB=4"; do
  echo "== $v" >> $o/fwd70b.log
  env $v SCHEDS=split python scripts/ll_bench.py fwd70b 2>&1 | grep "^forward" >> $o/fwd70b.log
done
cat $o/shapes70b.log $o/fwd70b.log
