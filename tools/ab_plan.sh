#!/bin/bash
# Same-box A/B of two PLANS of one library build: tune once (or take plans A), derive plans B with a python expression on the
# recorded plan dict `p` (e.g. swap a fused variant), replay both alternately.   usage (GPU box): bash tools/ab_plan.sh "<python stmt on p>" [reps]
O=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-extra --retune --save-plan $O/ab_plans_a.json --steps 60 > $O/ab_tune.json 2>/dev/null
python - "$1" <<PY
import json, sys
d = json.load(open("$O/ab_plans_a.json"))
for k, p in d.items():
    exec(sys.argv[1])
json.dump(d, open("$O/ab_plans_b.json", "w"))
print("A", {k: (v["fused_variants"], v["tiles"]) for k, v in json.load(open("$O/ab_plans_a.json")).items()})
print("B", {k: (v["fused_variants"], v["tiles"]) for k, v in d.items()})
PY
for rep in $(seq 1 ${2:-3}); do for ab in a b; do
  python bench.py --no-cpu-baseline --no-extra --plan $O/ab_plans_$ab.json --steps 100 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$ab', d['value'], d['timing']['mean_ms'], d['timing']['min_ms'], d['parity']['gpu_logits_bit_equal_oracle'], d['config']['plan_source'][:30])"
done; done
