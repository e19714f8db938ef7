# uneven sub-batch split: do two chains that drift apart (different kernels in flight at a time) beat two in lockstep?
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for sp in "64,64" "72,56" "80,48" "96,32"; do
  HAWQ_CHAINS=2 HAWQ_SPLIT=$sp timeout 300 python bench.py --steps 60 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('split $sp rep $rep', d['value'], d['ms_per_step'], d['parity']['gpu_logits_bit_equal_oracle'])"
done; done
