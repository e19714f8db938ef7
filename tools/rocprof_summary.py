"""Summarise a rocprofv3 --kernel-trace results .db (rocpd SQLite) into a per-kernel table.
usage: python tools/rocprof_summary.py <results.db> [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"void conv_kernel<Cfg<(\d+), (\d+), \d+, \d+>, (\d), (true|false)>", name)
    if m:
        epi = ["RAW", "REQUANT", "RESIDUAL", "DEQUANT"][int(m.group(3))]
        return f"conv_kernel<{m.group(1)}x{m.group(2)},{epi}{',DUAL' if m.group(4) == 'true' else ''}>"
    return name.split("(")[0][:90]


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), avg(duration), sum(duration), min(duration), max(duration), "
                     "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name "
                     "order by sum(duration) desc").fetchall()
    tot = sum(r[3] for r in rows)
    lines = ["| kernel | calls | avg us | min us | max us | total ms | % | vgpr | agpr | lds |", "|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        lines.append(f"| {short(r[0])} | {r[1]} | {r[2] / 1e3:.1f} | {r[4] / 1e3:.1f} | {r[5] / 1e3:.1f} | {r[3] / 1e6:.3f} | "
                     f"{100 * r[3] / tot:.1f} | {r[6]} | {r[7]} | {r[8]} |")
    lines.append(f"\ntotal kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
