#!/bin/bash
# Built copies of earlier commits under .bis/<sha> (git-ignored; they travel to the GPU box with the snapshot) for tools/drift_bisect.sh / drift_ensemble.sh.
# Usage: bash tools/build_bisect_copies.sh 86d1453 63af9ec 50c21da 80f5da0 c249ac2 48dadc3 66e39b8 3a463c1     (remove .bis/ afterwards: 15 MB per copy)
mkdir -p .bis
for c in "$@"; do
  mkdir -p .bis/$c && git archive $c | tar -x -C .bis/$c
  make -C .bis/$c/r-super_amd/csrc -j16 ARCH=gfx950 > /tmp/bis_$c.log 2>&1; make -C .bis/$c/oracle >> /tmp/bis_$c.log 2>&1
  echo $c $(ls .bis/$c/r-super_amd/csrc/*.so)
done
