#!/bin/bash
# Evidence for profiles/: run ON the GPU box (gpurun -- 'bash tools/profile_round.sh r01_c').  Writes gpurun_out/<tag>_*.
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err
HAWQ_CHAINS=1 python bench.py --no-cpu-baseline --per-op $O/${tag}_perop_single_chain.md > $O/${tag}_bench_single_chain.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
# the trace replays the tile / sub-batch choice of the run above, so that it holds no tuning launches
export HAWQ_TILES=$(python -c "import json,sys; print(json.load(open('$O/${tag}_bench.json'))['config']['autotuned_tiles'])")
export HAWQ_CHAINS=$(python -c "import json,sys; print(json.load(open('$O/${tag}_bench.json'))['config']['concurrent_sub_batches'])")
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python $R/bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 5 > $O/${tag}_bench_under_rocprof.json 2>/dev/null
python $R/tools/rocprof_summary.py $(find /tmp/kt -name "*.db" | head -1) > $O/${tag}_kernel_trace.md
for ctr in FETCH_SIZE WRITE_SIZE; do
  for steps in 2 12; do
    rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pm -o r -- python $R/bench.py --no-cpu-baseline --no-extra --steps $steps --warmup 1 > /dev/null 2>&1
    echo "$ctr steps=$steps $(python $R/tools/pmc_total.py $(find /tmp/pm -name '*.db' | head -1))" >> $O/${tag}_pmc_totals.txt
    if [ $steps = 12 ]; then python $R/tools/pmc_summary.py $(find /tmp/pm -name "*.db" | head -1) 20 > $O/${tag}_pmc_${ctr}.md; fi
  done
done
cat $O/${tag}_pmc_totals.txt
