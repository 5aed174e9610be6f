#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c16; mkdir -p $o
python -c "import __graft_entry__ as g; g.build()" > $o/build.log 2>&1
python scripts/r3/vg_ablate.py - default 2>&1 | grep "T=" >> $o/abl.log
for v in NOBAR NOGLOAD NODEQ NOLDS NOSTORE NOMFMA; do
  python scripts/r3/vg_ablate.py build/vg/lib_$v.so $v 2>&1 | grep "T=" >> $o/abl.log
done
python scripts/r3/vg_ablate.py - default 2>&1 | grep "T=" >> $o/abl.log
cat $o/abl.log
