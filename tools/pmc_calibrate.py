"""Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters this repo's `roofline.traffic` is built from:
elementwise torch kernels of KNOWN byte counts, below and above the 256 MiB Infinity Cache, each launched 4x back to back on the
same buffers (so that the repeats of a small tensor can hit the cache while a 1 GiB tensor cannot).

    cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/cal -o r -- python tools/pmc_calibrate.py run
    python tools/pmc_calibrate.py report /tmp/cal/.../r_results.db      # per-dispatch table, KB counted / bytes moved

Workloads (n fp32 elements):  read-only `x.sum()` (n * 4 B read), write-only `y.fill_(1)` (n * 4 B written), copy-like
`torch.add(x, 1, out=y)` (n * 4 B read + n * 4 B written); n * 4 B = 32 MiB, 128 MiB, 1 GiB."""
import sqlite3
import sys

SIZES = (32 << 20, 128 << 20, 1 << 30)


def run():
    import torch
    dev = torch.device("cuda:0")
    for nbytes in SIZES:
        n = nbytes // 4
        x = torch.ones(n, device=dev)
        y = torch.empty(n, device=dev)
        torch.cuda.synchronize()
        for _ in range(4):
            x.sum()
        torch.cuda.synchronize()
        for _ in range(4):
            y.fill_(1.0)
        torch.cuda.synchronize()
        for _ in range(4):
            torch.add(x, 1.0, out=y)
        torch.cuda.synchronize()
        del x, y
        torch.cuda.empty_cache()


def report(db):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    order = next((k for k in ("dispatch_id", "start", "id") if k in cols), None)   # counters_collection is a view: no rowid
    rows = c.execute("select kernel_name, grid_size, counter_name, value, duration from counters_collection" + (f" order by {order}" if order else "")).fetchall()
    print("| # | kernel | grid (threads) | counter | KB counted | us |")
    print("|---|---|---|---|---|---|")
    for i, (k, g, cn, v, d) in enumerate(rows):
        if g < (1 << 18):   # the tiny second stage of the reduction etc.
            continue
        name = k.split("(")[0].split("<")[0][-40:]
        kind = "sum (read)" if "reduce" in k else ("fill (write)" if "Fill" in k else ("add (read + write)" if "add" in k.lower() or "CUDAFunctor" in k else name))
        print(f"| {i} | {kind} | {g} | {cn} | {v:.6g} | {d / 1e3:.1f} |")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2])
