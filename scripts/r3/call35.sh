#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c35; mkdir -p $o; rm -f $o/*.log
for T in 256 769; do
  T=$T python scripts/r3/vg_ablate.py - vgprform >> $o/time.log 2>&1
  T=$T python scripts/r3/vg_ablate.py umbrella_amd/csrc/libumbrella_agpr.so agprform >> $o/time.log 2>&1
  T=$T python scripts/r3/vg_ablate.py - vgprform >> $o/time.log 2>&1
  T=$T python scripts/r3/vg_ablate.py umbrella_amd/csrc/libumbrella_agpr.so agprform >> $o/time.log 2>&1
done
grep "layer" $o/time.log | cut -c1-200
