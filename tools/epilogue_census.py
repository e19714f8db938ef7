"""Instruction census of the loops of a kernel, from a hipcc -S listing.

usage: python tools/epilogue_census.py <file.s> <mangled-or-demangled-name-fragment> [outputs-per-lane-per-iteration]

For every loop of every kernel whose (demangled) name contains the fragment: instructions by class between the loop header and its back
edge - VALU (by mnemonic), MFMA, LDS, vector memory, scalar, waits / barriers.  With the third argument the VALU count is also divided by
the number of outputs one lane produces per iteration (16 for the 32 x 32 MFMA tile of the residual epilogues): "VALU per output".
A wave64 VALU instruction occupies its SIMD for 4 cycles (v_mad_i64_i32 / v_mul_hi_u32: 8-16), so the table is the issue-side floor of
the loop; latencies are not in it."""
import collections
import re
import subprocess
import sys


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "MFMA"
    if op.startswith("v_accvgpr"):
        return "ACC-MOVE"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop") or op.startswith("s_sleep"):
        return "WAIT"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("v_"):
        return "VALU"
    return "OTHER"


def main():
    path, frag = sys.argv[1], sys.argv[2]
    per_iter = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    lines = open(path).read().split("\n")
    starts = [(i, m[1]) for i, ln in enumerate(lines) if (m := re.match(r"^(_Z\S+):", ln))]
    names = subprocess.run(["c++filt"] + [s[1] for s in starts], capture_output=True, text=True).stdout.split("\n")
    for (i0, sym), dem in zip(starts, names):
        if frag not in sym and frag not in dem:
            continue
        end = next(j for j in range(i0, len(lines)) if "s_endpgm" in lines[j])
        print(f"== {dem[:160]}")
        # loops: LLVM annotates every basic block of a loop ("=>This Inner Loop Header: Depth=d" on the header, "in Loop: Header=BBx_y" on
        # the others, on `.LBBx_y:` labels and on fall-through `; %bb.n:` markers alike) - a rotated loop's blocks need not be contiguous
        blocks, cur = [], None   # (loop key or None, depth, [instruction lines])
        for j in range(i0 + 1, end + 1):
            ln = lines[j]
            if re.match(r"^(\.LBB\S+:|; %bb\.\d+:)", ln):
                mh = re.match(r"^\.LBB(\S+):.*Loop Header: Depth=(\d+)", ln)
                mi = re.search(r"in Loop: Header=BB(\S+) Depth=(\d+)", ln)
                key = (mh[1], int(mh[2])) if mh else ((mi[1], int(mi[2])) if mi else None)
                cur = [key, []]
                blocks.append(cur)
            elif cur is not None:
                cur[1].append(ln)
        headers = {}
        for key, body in blocks:
            if key is not None:
                headers.setdefault(key, []).extend(body)
        for (lab, depth), body in headers.items():
            cls, valu = collections.Counter(), collections.Counter()
            for ln in body:
                ln = ln.strip()
                if not ln or ln.startswith((";", ".")) or ln.endswith(":"):
                    continue
                op = ln.split()[0]
                c = classify(op)
                cls[c] += 1
                if c == "VALU":
                    valu[re.sub(r"_e(32|64)$|_dpp$|_sdwa$", "", op)] += 1
            lab = ".LBB" + lab
            tot = sum(cls.values())
            if cls["VALU"] + cls["MFMA"] < 8:
                continue   # flag-polling / copy loops
            print(f"-- loop {lab} depth {depth}: {tot} instructions: " + ", ".join(f"{k} {v}" for k, v in cls.most_common()))
            if per_iter:
                print(f"   VALU per output: {cls['VALU'] / per_iter / 1.0:.2f}  (+ ACC-MOVE {cls['ACC-MOVE'] / per_iter:.2f}, LDS {cls['LDS'] / per_iter:.2f}) at {per_iter:g} outputs per lane and iteration")
            print("   VALU mix: " + ", ".join(f"{k} {v}" for k, v in valu.most_common(24)))


if __name__ == "__main__":
    main()
