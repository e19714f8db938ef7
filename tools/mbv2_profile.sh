#!/bin/bash
# MobileNetV2 evidence for profiles/ (run ON the GPU box):  gpurun -- 'bash tools/mbv2_profile.sh r04_e_mbv2'
# Writes gpurun_out/<tag>_*: the bench.py MobileNetV2 line, the per-launch table (tools/mbv2_perop.py), the rocprofv3 kernel trace of
# 20 single-chain replays and the HBM traffic of one forward (separate FETCH_SIZE / WRITE_SIZE passes, counter totals at 12 replays
# minus 2 replays, divided by 10; FETCH_SIZE doubled per MI355X_MICROARCH.md; counters report KiB).
tag=${1:-rXX_mbv2}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
python -c "import bench, torch, json; print(json.dumps(bench.mobilenet_line(128, torch.device('cuda'), 40)))" 2>/dev/null | tail -1 > $O/${tag}_bench_line.json
python tools/mbv2_perop.py 128 > $O/${tag}_perop.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python $R/tools/mbv2_run.py 128 20 > $O/${tag}_run_under_rocprof.txt 2>/dev/null
python $R/tools/rocprof_summary.py $(find /tmp/kt -name "*.db" | head -1) > $O/${tag}_kernel_trace.md
rm -f $O/${tag}_pmc_totals.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  for steps in 2 12; do
    rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pm -o r -- python $R/tools/mbv2_run.py 128 $steps > /dev/null 2>&1
    echo "$ctr steps=$steps $(python $R/tools/pmc_total.py $(find /tmp/pm -name '*.db' | head -1))" >> $O/${tag}_pmc_totals.txt
  done
done
python - <<PY
import json, re
tot = {}
for line in open("$O/${tag}_pmc_totals.txt"):
    m = re.match(r"(\w+) steps=(\d+) \1 ([\d.e+]+) (\d+)", line)
    if m:
        tot[(m[1], int(m[2]))] = float(m[3])
fetch_kb = (tot[("FETCH_SIZE", 12)] - tot[("FETCH_SIZE", 2)]) / 10
write_kb = (tot[("WRITE_SIZE", 12)] - tot[("WRITE_SIZE", 2)]) / 10
b = json.loads(open("$O/${tag}_bench_line.json").readline())
out = {"mobilenetv2_w1_uniform8_b128": {"bytes_per_forward": round((2 * fetch_kb + write_kb) * 1024.0), "fetch_size_kb_raw": fetch_kb, "write_size_kb": write_kb,
       "plan_bytes_per_forward": b["plan_bytes_per_image"] * 128, "images_per_s": b["images_per_s"], "git_head": "${GRAFT_HEAD:-unknown}",
       "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (--kernel-trace only) over tools/mbv2_run.py 128 <steps>: totals at 12 replays minus 2 replays, / 10; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B for 16 B/lane streams, MI355X_MICROARCH.md); counters in KiB; Infinity-Cache hits included: an upper bound on HBM bytes"}}
json.dump(out, open("$O/${tag}_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
cat $O/${tag}_pmc_totals.txt
