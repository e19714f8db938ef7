"""Build container, after tools/profile_round.sh ran on the GPU box: copy one tag's summaries from gpurun_out/ into profiles/ and merge its
<tag>_traffic.json / <tag>_dominant_kernel.json into profiles/traffic.json / profiles/dominant_kernel.json (what bench.py reads).
usage: python tools/merge_evidence.py <tag> [<tag> ...]"""
import glob
import json
import os
import shutil
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for tag in sys.argv[1:]:
    for kind in ("traffic", "dominant_kernel"):
        src = os.path.join(root, "gpurun_out", f"{tag}_{kind}.json")
        if not os.path.exists(src) or not os.path.getsize(src):
            continue
        dst = os.path.join(root, "profiles", f"{kind}.json")
        cur = json.load(open(dst)) if os.path.exists(dst) else {}
        cur.update(json.load(open(src)))
        json.dump(cur, open(dst, "w"), indent=1, sort_keys=True)
    n = 0
    for f in glob.glob(os.path.join(root, "gpurun_out", f"{tag}_*")):
        if f.endswith(("_plans.json", "_plans_single_chain.json", ".log")) or os.path.getsize(f) > 400_000:
            continue
        shutil.copy(f, os.path.join(root, "profiles", os.path.basename(f)))
        n += 1
    print(tag, n, "files")
