#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c30; mkdir -p $o; rm -f $o/*.log
for T in 256 769; do
  UMB_VG_PP=0 T=$T python scripts/r3/vg_ablate.py - old >> $o/time.log 2>&1
  T=$T python scripts/r3/vg_ablate.py - M-bd3 >> $o/time.log 2>&1
  for v in X XBD1 MBD2; do T=$T python scripts/r3/vg_ablate.py umbrella_amd/csrc/libumbrella_pp$v.so $v >> $o/time.log 2>&1; done
  UMB_VG_PP=0 T=$T python scripts/r3/vg_ablate.py - old >> $o/time.log 2>&1
done
grep "layer" $o/time.log
