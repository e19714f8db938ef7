"""How many kernels of the forward are on the GPU at once?  Sweep over the kernel intervals of the LAST `n_forwards` hipGraph replays in a
rocprofv3 --kernel-trace results .db: share of the span with 0 / 1 / 2 / ... kernels in flight, and per kernel the time it ran alone.
usage: python tools/rocprof_overlap.py <results.db> [n_forwards=10]   (run ON the GPU box; tools/profile_round.sh calls it)
A forward is delimited by the stem launches: with c concurrent sub-batch chains every forward starts c stems."""
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from rocprof_summary import short  # noqa: E402

c = sqlite3.connect(sys.argv[1])
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 10
chains = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows = c.execute("select name, start, end from kernels order by start").fetchall()
stems = [i for i, r in enumerate(rows) if "stem" in r[0]]
if len(stems) < chains * (nf + 1):
    sys.exit("trace too short")
first = stems[-chains * nf]
rows = rows[first:]
ev = []
for k, (n, s, e) in enumerate(rows):
    ev += [(s, 1, k), (e, -1, k)]
ev.sort()
depth, t_prev, hist, alone, live = 0, ev[0][0], {}, {}, set()
for t, d, k in ev:
    hist[depth] = hist.get(depth, 0) + t - t_prev
    if depth == 1:
        n = short(rows[next(iter(live))][0])
        alone[n] = alone.get(n, 0) + t - t_prev
    t_prev = t
    depth += d
    (live.add if d > 0 else live.discard)(k)
span = ev[-1][0] - ev[0][0]
busy = sum(e - s for _, s, e in rows)
print(f"last {nf} forwards ({chains} chains): {len(rows)} kernels, span {span / 1e3 / nf:.1f} us per forward, summed kernel time {busy / 1e3 / nf:.1f} us per forward\n")
print("| kernels in flight | us per forward | share of the span |\n|---|---|---|")
for d in sorted(hist):
    print(f"| {d} | {hist[d] / 1e3 / nf:.1f} | {100 * hist[d] / span:.1f} % |")
print("\n| kernel | us per forward it ran with nothing beside it |\n|---|---|")
for n, t in sorted(alone.items(), key=lambda kv: -kv[1])[:12]:
    print(f"| {n} | {t / 1e3 / nf:.1f} |")
