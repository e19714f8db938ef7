#!/bin/bash
# GPU box: gpurun -- 'bash tools/pmc_calibrate.sh'  ->  gpurun_out/r03_cal_{FETCH_SIZE,WRITE_SIZE}.md
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal; rocprofv3 --kernel-trace --pmc $ctr -d /tmp/cal -o r -- python $R/tools/pmc_calibrate.py run > /dev/null 2>&1
  python $R/tools/pmc_calibrate.py report $(find /tmp/cal -name "*.db" | head -1) > $O/r03_cal_$ctr.md
done
cat $O/r03_cal_FETCH_SIZE.md $O/r03_cal_WRITE_SIZE.md
