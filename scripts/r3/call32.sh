#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c32; mkdir -p $o; rm -f $o/*.log
for T in 256 769; do
  UMB_VG_PP=0 T=$T python scripts/r3/vg_ablate.py - old >> $o/time.log 2>&1
  T=$T python scripts/r3/vg_ablate.py - lm7 >> $o/time.log 2>&1
  for m in 0 1 2 3 6; do T=$T python scripts/r3/vg_ablate.py umbrella_amd/csrc/libumbrella_lm$m.so lm$m >> $o/time.log 2>&1; done
  T=$T python scripts/r3/vg_ablate.py - lm7 >> $o/time.log 2>&1
done
grep "layer" $o/time.log | cut -c1-200
