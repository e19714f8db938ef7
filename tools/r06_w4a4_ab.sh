#!/bin/bash
# GPU box (round 6): does 4-bit pay?  Same box, fresh plans (HAWQ_TUNE_TRIALS=2), interleaved:
#   W8A8 | W4A4 policy 1 (nibble expand inputs, 2 fused pairs) | W4A4 policy 2 (int8 expand inputs: the W8A8 pair structure, nibble 3x3 and
#   reduce convs) | policy 2 without the nibble streaming 1x1 kernel | bops_0.5 policy 1 / 2
#   gpurun -- 'bash tools/r06_w4a4_ab.sh [reps]'
cd $GRAFT_REPO_ROOT
reps=${1:-2}
export HAWQ_TUNE_TRIALS=${HAWQ_TUNE_TRIALS:-2}
run() {  # label scheme env...
  label=$1; scheme=$2; shift 2
  env "$@" timeout 600 python bench.py --scheme $scheme --retune --steps 100 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d['config']
print('$label', d['value'], 'img/s', d['ms_per_step'], 'ms  parity', d['parity']['gpu_logits_bit_equal_oracle'], ' chains', c['concurrent_sub_batches'], ' fused pairs', c['fused_expand_reduce_launches'], ' tiles', c['autotuned_tiles'], ' variants', c['fused_variants'])"
}
for rep in $(seq $reps); do
  run "rep$rep W8A8            " uniform8 HAWQ_EXPAND_IN8=1
  run "rep$rep W4A4 policy1    " uniform4 HAWQ_EXPAND_IN8=1
  run "rep$rep W4A4 policy2    " uniform4 HAWQ_EXPAND_IN8=2
  run "rep$rep W4A4 policy2-g2n" uniform4 HAWQ_EXPAND_IN8=2 HAWQ_NO_GEMM2_NIB=1
  run "rep$rep bops policy1    " bops_0.5 HAWQ_EXPAND_IN8=1
  run "rep$rep bops policy2    " bops_0.5 HAWQ_EXPAND_IN8=2
done
