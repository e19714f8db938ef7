"""The dominant rocprof kernel of a profiled bench run, as a small JSON record (bench.py copies the committed one into
`roofline.dominant_kernel`).  usage: python tools/dominant_kernel.py <kernel-trace results.db> <tag>_pmc_MFMA.md <workload> <git head>
Run ON the GPU box by tools/profile_round.sh.  Only kernels of the forward count (tuning / calibration / copy kernels are skipped);
launches per forward = calls / forwards, forwards = stem launches / sub-batch chains (every chain of every forward launches the stem once)."""
import json
import re
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from rocprof_summary import short  # noqa: E402

db, pmc_md, workload, head = sys.argv[1:5]
chains = int(sys.argv[5]) if len(sys.argv) > 5 else 2
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), avg(duration), sum(duration), max(vgpr_count), max(lds_size) from kernels group by name order by sum(duration) desc").fetchall()
fwd = [r for r in rows if re.search(r"conv|expand|band|gemm|stem|bottleneck|depthwise|avgpool", r[0]) and not re.search(r"rocclr|at::native", r[0])]
stem = [r for r in fwd if "stem" in r[0]]
forwards = max(1.0, sum(r[1] for r in stem) / chains)
tot = sum(r[3] for r in fwd)
top = fwd[0]
busy = None
try:   # time-weighted matrix-pipe utilisation of that kernel from the PMC pass (same formula as tools/mfma_util.py)
    hdr, num, den = None, 0.0, 0.0
    for line in open(pmc_md):
        cells = [x.strip() for x in line.strip().strip("|").split("|")]
        if hdr is None:
            hdr = cells if cells and cells[0] == "kernel" else None
            continue
        if len(cells) != len(hdr) or set(cells[0]) <= set("-"):
            continue
        r = dict(zip(hdr, cells))
        if r["kernel"] == short(top[0]):
            n, us, b = float(r["n"]), float(r["avg us"]), float(r["SQ_VALU_MFMA_BUSY_CYCLES"])
            num += n * b
            den += n * us * 1e-6 * 2.1e9 * 1024
    busy = round(num / den, 4) if den else None
except OSError:
    pass
print(json.dumps({workload: {
    "rocprof_name": short(top[0]), "launches_per_forward": round(top[1] / forwards, 2), "avg_us": round(top[2] / 1e3, 2),
    "share_of_forward_kernel_time": round(top[3] / tot, 4), "vgpr": top[4], "lds_bytes": top[5],
    "mfma_busy_frac": busy, "forward_kernels_in_trace": len(fwd), "forwards_in_trace": round(forwards, 1), "git_head": head,
    "source": "rocprofv3 --kernel-trace / --pmc passes of tools/profile_round.sh on the recorded plan (profiles/<tag>_kernel_trace.md, <tag>_mfma_utilisation.md)"}}))
