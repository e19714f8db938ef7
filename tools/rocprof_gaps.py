"""Idle gaps between consecutive kernels of the LAST forward (hipGraph replay) in a rocprofv3 kernel-trace .db.
usage: python tools/rocprof_gaps.py <results.db> [n_kernels_per_forward]"""
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from rocprof_summary import short  # noqa: E402

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 53
last = rows[-n:]
busy = sum(e - s for _, s, e in last)
span = last[-1][2] - last[0][1]
gaps = [(last[i + 1][1] - last[i][2]) for i in range(len(last) - 1)]
print(f"last forward: {len(last)} kernels, span {span / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, "
      f"gaps {sum(gaps) / 1e3:.1f} us (mean {sum(gaps) / len(gaps) / 1e3:.2f}, max {max(gaps) / 1e3:.2f})")
print("first / last kernel:", short(last[0][0]), "/", short(last[-1][0]))
