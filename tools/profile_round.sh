#!/bin/bash
# Evidence for profiles/: run ON the GPU box
#   gpurun -- 'GRAFT_HEAD=<git head> bash tools/profile_round.sh r04_c'                      (ResNet50 uniform8, the headline workload)
#   gpurun -- 'GRAFT_HEAD=<git head> ARCH=resnet18 bash tools/profile_round.sh r04_c_resnet18'
#   gpurun -- 'SCHEME=uniform4 LIGHT=1 bash tools/profile_round.sh r04_c_uniform4'            (LIGHT: bench + kernel trace + traffic only)
# Writes gpurun_out/<tag>_*: the bench line (default `python bench.py` for the headline workload: it REPLAYS profiles/plans.json when that file
# has the workload, else it tunes; the plan it ran is saved to <tag>_plans.json and every later pass of this script replays exactly that plan),
# single-chain per-launch table, rocprofv3 kernel trace, HBM traffic (separate FETCH_SIZE / WRITE_SIZE PMC passes, counter totals at --steps 12
# minus --steps 2), MFMA / VALU utilisation counters per kernel, and <tag>_traffic.json (merge into profiles/traffic.json) recorded WITH the plan.
tag=${1:-rXX}
ARCH=${ARCH:-resnet50}
SCHEME=${SCHEME:-uniform8}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
W="--arch $ARCH --scheme $SCHEME"
if [ "$ARCH $SCHEME" = "resnet50 uniform8" ] && [ -z "$LIGHT" ]; then
  python bench.py ${BENCH_ARGS} --save-plan $O/${tag}_plans.json > $O/${tag}_bench.json 2> $O/${tag}_bench.err
else
  python bench.py $W --no-cpu-baseline --no-extra ${BENCH_ARGS} --save-plan $O/${tag}_plans.json > $O/${tag}_bench.json 2> $O/${tag}_bench.err
fi
P="--plan $O/${tag}_plans.json"
if [ -z "$LIGHT" ]; then
  # single chain, TUNED as a single chain (as in rounds 1-3, so the per-launch tables compare across rounds): per-launch table against the FUSED plan's byte model
  HAWQ_CHAINS=1 python bench.py $W --retune --no-cpu-baseline --no-extra --per-op $O/${tag}_perop_single_chain.md > $O/${tag}_bench_single_chain.json 2>/dev/null
fi
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python $R/bench.py $W $P --no-cpu-baseline --no-extra --steps 20 --warmup 5 > $O/${tag}_bench_under_rocprof.json 2>/dev/null
python $R/tools/rocprof_summary.py $(find /tmp/kt -name "*.db" | head -1) > $O/${tag}_kernel_trace.md
CHN=$(python -c "import json; print(json.loads(open('$O/${tag}_bench.json').readline())['config']['concurrent_sub_batches'])")
python $R/tools/rocprof_overlap.py $(find /tmp/kt -name "*.db" | head -1) 10 $CHN > $O/${tag}_kernel_overlap.md 2>&1
rm -f $O/${tag}_pmc_totals.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  for steps in 2 12; do
    rm -rf /tmp/pm; HAWQ_BENCH_SPIN_UP=0 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pm -o r -- python $R/bench.py $W $P --no-cpu-baseline --no-extra --steps $steps --warmup 1 > /dev/null 2>&1
    echo "$ctr steps=$steps $(python $R/tools/pmc_total.py $(find /tmp/pm -name '*.db' | head -1))" >> $O/${tag}_pmc_totals.txt
    if [ $steps = 12 ] && [ -z "$LIGHT" ]; then python $R/tools/pmc_summary.py $(find /tmp/pm -name "*.db" | head -1) 20 > $O/${tag}_pmc_${ctr}.md; fi
  done
done
if [ -z "$LIGHT" ]; then
  # matrix-pipe / VALU utilisation per kernel (north_star: "rocprof HBM GB/s and MFMA utilisation")
  rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/pm -o r -- python $R/bench.py $W $P --no-cpu-baseline --no-extra --steps 6 --warmup 1 > /dev/null 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/pm -name "*.db" | head -1) 8 > $O/${tag}_pmc_MFMA.md
  python $R/tools/pmc_total.py $(find /tmp/pm -name "*.db" | head -1) > $O/${tag}_pmc_MFMA_totals.txt
  python $R/tools/mfma_util.py $O/${tag}_pmc_MFMA.md > $O/${tag}_mfma_utilisation.md
fi
# the dominant kernel of the forward (bench.py: roofline.dominant_kernel) from the kernel trace above; LIGHT runs have no MFMA pass (mfma_busy_frac null)
python $R/tools/dominant_kernel.py $(find /tmp/kt -name "*.db" | head -1) $O/${tag}_pmc_MFMA.md ${ARCH}_${SCHEME}_b128 ${GRAFT_HEAD:-unknown} $CHN > $O/${tag}_dominant_kernel.json
python - <<PY
import json, re
tot = {}
for line in open("$O/${tag}_pmc_totals.txt"):
    m = re.match(r"(\w+) steps=(\d+) \1 ([\d.e+]+) (\d+)", line)
    if m:
        tot[(m[1], int(m[2]))] = float(m[3])
fetch_kb = (tot[("FETCH_SIZE", 12)] - tot[("FETCH_SIZE", 2)]) / 10
write_kb = (tot[("WRITE_SIZE", 12)] - tot[("WRITE_SIZE", 2)]) / 10
b = json.loads(open("$O/${tag}_bench.json").readline())
plan = {"tiles": b["config"]["autotuned_tiles"], "fused_variants": b["config"]["fused_variants"], "chains": b["config"]["concurrent_sub_batches"]}
out = {b["config"]["workload"]: {
    "bytes_per_launch": round((2 * fetch_kb + write_kb) * 1024.0), "fetch_size_kb_raw": fetch_kb, "write_size_kb": write_kb,
    "git_head": "${GRAFT_HEAD:-unknown}", "tag": "$tag", "fused_pairs": b["config"].get("fused_pairs", []),
    # the plan the counters were collected for (replayed from ${tag}_plans.json): bench.py prints whether a later run's plan is the same
    "plan": plan, "plan_source": b["config"].get("plan_source"), "images_per_s_of_the_plan_run": b["value"],
    "upper_bound_note": "FETCH_SIZE / WRITE_SIZE count requests at the L2's memory side; Infinity-Cache hits are included (profiles/r03_pmc_calibration.md), so this is an upper bound on HBM bytes",
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace only), counter summed over ALL dispatches of bench.py --plan <the plan> --no-cpu-baseline --no-extra --warmup 1 at --steps 12 minus the same at --steps 2, divided by 10 forwards (tools/profile_round.sh, tools/pmc_total.py); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B for 16 B/lane streams); the counters report KiB (profiles/r03_pmc_calibration.md: 32 MiB written = 32 768)"}}
json.dump(out, open("$O/${tag}_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
cat $O/${tag}_pmc_totals.txt
# the four passes must have launched the same number of kernels apart from the 10 extra forwards (else the subtraction is void)
python - <<PY
import re
n = {}
for line in open("$O/${tag}_pmc_totals.txt"):
    m = re.match(r"(\w+) steps=(\d+) \1 ([\d.e+]+) (\d+)", line)
    if m:
        n[(m[1], int(m[2]))] = int(m[4])
ok = n[("FETCH_SIZE", 2)] == n[("WRITE_SIZE", 2)] and n[("FETCH_SIZE", 12)] == n[("WRITE_SIZE", 12)]
print("dispatch counts of the counter passes", "agree" if ok else "DIFFER - traffic figure invalid", n)
PY
