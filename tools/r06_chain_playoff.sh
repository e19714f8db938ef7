#!/bin/bash
# GPU box (round 6, late): is the chain count the tuner picks the right one?  Per workload: two plans tuned with HAWQ_CHAINS=2 and two with
# HAWQ_CHAINS=3 (HAWQ_TUNE_TRIALS=6), then all four (and the recorded plan) replayed alternately, 3 rounds x 60 timed steps.
#   gpurun -- 'bash tools/r06_chain_playoff.sh "uniform8 uniform4 bops_0.5"'   -> gpurun_out/r6c_<scheme>_playoff.txt, r6c_<scheme>_plans_c<chains>_<i>.json
cd $GRAFT_REPO_ROOT
O=gpurun_out
export HAWQ_TUNE_TRIALS=${HAWQ_TUNE_TRIALS:-6}
pr() { python -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('$1', d['value'], 'img/s', d['ms_per_step'], 'ms | gpu', d['timing']['mean_ms'], '+-', d['timing']['std_ms'], '| parity', d['parity']['gpu_logits_bit_equal_oracle'], '| chains', d['config']['concurrent_sub_batches'], '| variants', d['config']['fused_variants'].replace('.', ' '))"; }
for scheme in ${1:-uniform8}; do
  W="--arch ${ARCH:-resnet50} --scheme $scheme"
  {
  for c in 2 3; do for i in 1 2; do
    HAWQ_CHAINS=$c python bench.py $W --retune --no-cpu-baseline --no-extra --save-plan $O/r6c_${scheme}_plans_c${c}_$i.json 2>/dev/null | pr "$scheme tuned, $c chains, #$i:"
  done; done
  for rnd in 1 2 3; do
    python bench.py $W --no-extra --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | pr "$scheme round $rnd recorded (or tuned, if stale):"
    for c in 2 3; do for i in 1 2; do
      python bench.py $W --plan $O/r6c_${scheme}_plans_c${c}_$i.json --no-extra --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | pr "$scheme round $rnd c$c #$i:"
    done; done
  done
  } > $O/r6c_${scheme}_playoff.txt
  cat $O/r6c_${scheme}_playoff.txt
done
