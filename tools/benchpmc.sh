#!/bin/bash
# SQ PMC counters of the whole bench (run on the GPU box); prints per-kernel averages.  usage: benchpmc.sh "<counters>" [min_us]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $1 -d /tmp/pm -o r -- python $R/bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 > /dev/null 2>&1
python $R/tools/pmc_summary.py $(find /tmp/pm -name "*.db" | head -1) ${2:-15}
