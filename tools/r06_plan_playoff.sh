#!/bin/bash
# GPU box (round 6): n MORE tuning runs of the headline workload, then a play-off: the recorded headline plan (profiles/plans.json) and every
# new plan replayed alternately, 3 rounds x 60 timed steps - a new plan replaces the record only if it wins every round.
#   gpurun -- 'GRAFT_HEAD=<head> bash tools/r06_plan_playoff.sh 3'     -> gpurun_out/r6_playoff.txt, r6_plans_p<i>.json
cd $GRAFT_REPO_ROOT
n=${1:-3}
O=gpurun_out
W="--arch ${ARCH:-resnet50} --scheme ${SCHEME:-uniform8}"   # ARCH= / SCHEME= : any workload of profiles/plans.json
T=${TAG:-r6}
# TUNE_ENV="HAWQ_CHAINS=3": environment of the tuning runs only (the replays take everything from the plan files)
export HAWQ_TUNE_TRIALS=${HAWQ_TUNE_TRIALS:-6}
pr() { python -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('$1', d['value'], 'img/s', d['ms_per_step'], 'ms | gpu', d['timing']['mean_ms'], '+-', d['timing']['std_ms'], '| trials', d['config'].get('plan_trials_ms'), '| tiles', d['config']['autotuned_tiles'].replace('.', ' '), '| variants', d['config']['fused_variants'].replace('.', ' '))"; }
{
for i in $(seq $n); do
  env $TUNE_ENV python bench.py $W --retune --no-cpu-baseline --no-extra --save-plan $O/${T}_plans_p$i.json 2>/dev/null | pr "tuning run p$i:"
done
for rnd in 1 2 3; do
  python bench.py $W --no-extra --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | pr "round $rnd recorded:"
  for i in $(seq $n); do
    python bench.py $W --plan $O/${T}_plans_p$i.json --no-extra --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | pr "round $rnd p$i:     "
  done
done
} > $O/${T}_playoff.txt
cat $O/${T}_playoff.txt
