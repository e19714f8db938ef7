#!/bin/bash
# Ablation of the role-split pair kernel at the stage-3 shape (probe library; results wrong on purpose): what does a slice wait for?
# usage (GPU box): bash tools/er2_ablate.sh [batch]
cd $GRAFT_REPO_ROOT
A=$PWD/hawq_amd/lib/libhawq_mi355_ablate.so
for bits in 0 1 2 4 8 16 12 28 31; do
  echo -n "HAWQ_DBG=$bits: "
  HAWQ_LIB=$A HAWQ_DBG=$bits ERPROBE_ONLY=14 ERPROBE_TIME=1 python tools/erprobe.py ${1:-64} 2>/dev/null | grep "variant 7" | sed 's/.*variant 7: //'
done
