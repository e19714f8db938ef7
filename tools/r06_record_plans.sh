#!/bin/bash
# GPU box (round 6): candidates for profiles/plans.json.  N complete tuning runs of the default bench (headline + every extra workload,
# `--retune --save-plan`), then the HEADLINE workload replayed from every plan file alternately (two rounds), as profiles/r05_plan_choice.txt did:
# two tuning runs differ by +-1.5 % when replayed - more than the tuner's own trial times resolve -, so the record is "tune N times, replay all, keep the fastest".
#   gpurun -- 'GRAFT_HEAD=<head> bash tools/r06_record_plans.sh 3'     -> gpurun_out/r6_plans_t<i>.json, r6_plan_choice.txt
cd $GRAFT_REPO_ROOT
n=${1:-3}
O=gpurun_out
export HAWQ_TUNE_TRIALS=${HAWQ_TUNE_TRIALS:-4}
pr() { python -c "
import json, sys
d = json.loads(sys.stdin.readline()); e = d.get('extra', {})
print('$1', d['value'], 'img/s', d['ms_per_step'], 'ms | trials', d['config'].get('plan_trials_ms'), '|', ' '.join(f\"{k.replace('resnet50_uniform8_','').replace('_b128','')}:{v.get('images_per_s')}\" for k, v in e.items() if isinstance(v, dict)))"; }
for i in $(seq $n); do
  python bench.py --retune --no-cpu-baseline --save-plan $O/r6_plans_t$i.json 2> $O/r6_tune_t$i.err | tee $O/r6_tune_t$i.json | pr "tuning run t$i:"
done > $O/r6_plan_choice.txt
for rnd in 1 2; do
  for i in $(seq $n); do
    python bench.py --plan $O/r6_plans_t$i.json --no-extra --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | pr "replay $rnd of t$i:"
  done
done >> $O/r6_plan_choice.txt
cat $O/r6_plan_choice.txt
