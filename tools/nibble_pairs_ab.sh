# W4A4 / mixed schedules: 4-bit expand inputs stored as int8 so that the fused expand -> reduce launches (which now write the hawq4
# output of the reduce conv themselves) take the stage 1-3 pairs, against the native nibble storage without those pairs
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for scheme in uniform4 bops_0.5; do
for v in 1 0; do
  HAWQ_EXPAND_IN8=$v timeout 300 python bench.py --scheme $scheme --steps 60 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$scheme EXPAND_IN8=$v rep $rep', d['value'], d['ms_per_step'], d['parity']['gpu_logits_bit_equal_oracle'], d['config']['fused_expand_reduce_launches'], d['config'].get('wave_private_solo_launches'))"
done; done; done
