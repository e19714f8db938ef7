"""Per-kernel averages of rocprofv3 --pmc counters from a results .db (run ON the GPU box; the
.db files are too large to copy back).  usage: python tools/pmc_summary.py <results.db> [min_us]
Rows are grouped by (kernel, grid size) so that individual layers stay visible."""
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from rocprof_summary import short  # noqa: E402


def main():
    c = sqlite3.connect(sys.argv[1])
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
    rows = c.execute("select kernel_name, grid_size, counter_name, avg(value), avg(duration), count(*) "
                     "from counters_collection group by kernel_name, grid_size, counter_name").fetchall()
    tab, names = {}, []
    for k, g, cn, v, d, n in rows:
        tab.setdefault((k, g), {"dur": d, "n": n})[cn] = v
        if cn not in names:
            names.append(cn)
    print("| kernel | grid | n | avg us | " + " | ".join(names) + " |")
    print("|---|---|---|---|" + "---|" * len(names))
    for (k, g), d in sorted(tab.items(), key=lambda kv: -kv[1]["dur"] * kv[1]["n"]):
        if d["dur"] / 1e3 < min_us:
            continue
        print(f"| {short(k)} | {g // 256} | {d['n']} | {d['dur'] / 1e3:.1f} | " +
              " | ".join(f"{d.get(n, float('nan')):.4g}" for n in names) + " |")


if __name__ == "__main__":
    main()
