#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c34; mkdir -p $o; rm -f $o/*.log
for T in 256 769; do
  T=$T python scripts/r3/vg_ablate.py - lm7 >> $o/time.log 2>&1
  for m in nf0 nf7 nf7b3 nf0b3; do T=$T python scripts/r3/vg_ablate.py umbrella_amd/csrc/libumbrella_$m.so $m >> $o/time.log 2>&1; done
  T=$T python scripts/r3/vg_ablate.py - lm7 >> $o/time.log 2>&1
done
grep "layer" $o/time.log | cut -c1-200
