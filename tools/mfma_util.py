"""<tag>_pmc_MFMA.md (tools/pmc_summary.py table of the MFMA / VALU counter pass) -> matrix-pipe utilisation per kernel.
usage: python tools/mfma_util.py gpurun_out/r02_d_pmc_MFMA.md > gpurun_out/r02_d_mfma_utilisation.md"""
import sys

rows = []
hdr = None
for line in open(sys.argv[1]):
    cells = [c.strip() for c in line.strip().strip("|").split("|")]
    if hdr is None:
        if cells and cells[0] == "kernel":
            hdr = cells
        continue
    if len(cells) != len(hdr) or set(cells[0]) <= set("-"):
        continue
    r = dict(zip(hdr, cells))
    try:
        busy, us = float(r["SQ_VALU_MFMA_BUSY_CYCLES"]), float(r["avg us"])
    except ValueError:
        continue
    if busy <= 0:
        continue
    rows.append((r["kernel"], r["grid"], r["n"], us, busy / 32, busy / (us * 1e-6 * 2.1e9 * 1024) * 100))
print("# Matrix-pipe utilisation per kernel (rocprofv3 --pmc, `" + sys.argv[1].split("/")[-1] + "`)\n")
print("`SQ_VALU_MFMA_BUSY_CYCLES` counts 32 cycles per `v_mfma_i32_32x32x32_i8`.  Utilisation = busy cycles / (kernel duration x 2.1 GHz x\n"
      "1024 SIMDs); 2.1 GHz is the clock an all-CU MFMA load sustains (`r01_ubench_mfma_rate.txt`), so rows of memory-bound kernels (which\n"
      "clock higher) are slightly over-stated; durations are those of the profiled run.  Batch 64 per launch (two concurrent sub-batches\n"
      "of the batch-128 forward).  `grid/256` is the launch's thread count / 256.\n")
print("| kernel | grid/256 | launches | avg us | MFMAs per launch | matrix pipe busy |\n|---|---|---|---|---|---|")
for k, g, n, us, nm, u in rows:
    print(f"| {k} | {g} | {n} | {us:.1f} | {nm:.3g} | {u:.1f} % |")
