#!/bin/bash
# GPU box: same-plan A/B of two builds of libhawq_mi355.so (same ABI).  The plan (tile ids, fused variants, chains) is tuned
# ONCE with library B, then replayed - no autotuning noise - alternately with A and B in ONE call on ONE box.
#   gpurun -- 'bash tools/ab_lib.sh <libA.so> <libB.so> [reps]'       (paths relative to the repo root)
R=$GRAFT_REPO_ROOT; A=$R/$1; B=$R/$2; reps=${3:-3}
cd $R
HAWQ_LIB=$B python bench.py --no-extra --no-cpu-baseline --steps 30 --warmup 5 > /tmp/plan.json 2>/dev/null
cfg() { python -c "import json; print(json.loads(open('/tmp/plan.json').readline())['config']['$1'])"; }
export HAWQ_TILES=$(cfg autotuned_tiles) HAWQ_CHAINS=$(cfg concurrent_sub_batches) HAWQ_ER_TILES=$(cfg fused_variants) HAWQ_ER_SPLIT_TILES=$(cfg fused_split_tiles)
echo "plan: chains $HAWQ_CHAINS tiles $HAWQ_TILES fused $HAWQ_ER_TILES"
pr() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['value'], 'img/s; gpu ms', d['timing']['mean_ms'], '+-', d['timing']['std_ms'], 'min', d['timing']['min_ms'], 'parity', d['parity']['gpu_logits_bit_equal_oracle'])"; }
for i in $(seq $reps); do
  HAWQ_LIB=$A python bench.py --no-extra --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | pr A
  HAWQ_LIB=$B python bench.py --no-extra --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | pr B
done
