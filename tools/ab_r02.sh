#!/bin/bash
# GPU box: interleaved A/B of the whole forward, round-2 head (worktree _r02, built beside the repo) vs this tree, ONE box, ONE call.
#   gpurun -- 'bash tools/ab_r02.sh [reps] [extra bench args]'
R=$GRAFT_REPO_ROOT; reps=${1:-2}; shift
pr() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['value'], 'img/s', d['ms_per_step'], 'ms; gpu', d['timing']['mean_ms'], '+-', d['timing']['std_ms'], 'chains', d['config']['concurrent_sub_batches'], 'parity', d['parity']['gpu_logits_bit_equal_oracle'])"; }
for i in $(seq $reps); do
  (cd $R/_r02 && python bench.py --no-extra --no-cpu-baseline --steps 80 --warmup 10 "$@" 2>/dev/null | pr r02)
  (cd $R && python bench.py --no-extra --no-cpu-baseline --steps 80 --warmup 10 "$@" 2>/dev/null | pr r03)
done
