"""Where does a kernel spill?  usage: python tools/spill_sites.py <file.s> <mangled-name-fragment>
Prints, for every kernel of a hipcc -S listing whose symbol contains the fragment and that uses scratch, each scratch_load / scratch_store
with the innermost loop label it sits in ("-" = straight-line code outside every loop)."""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
frag = sys.argv[2]
i = 0
while i < len(lines):
    m = re.match(r"^(_Z\S+):\s", lines[i])
    if not m or frag not in m[1]:
        i += 1
        continue
    name, j, loop, hits = m[1], i + 1, "-", []
    while j < len(lines) and "s_endpgm" not in lines[j]:
        ln = lines[j]
        lm = re.search(r";\s+(?:=>\s*)?(?:This )?(?:Inner |Parent )?Loop (?:Header: Depth=(\d+)|BB\S+ Depth=(\d+))", ln)
        if re.match(r"^\.LBB\S+:", ln):
            loop = "in-loop depth " + (lm[1] or lm[2]) if lm else ("in-loop" if "in Loop" in ln else "-")
        elif re.match(r"^; %bb", ln):
            loop = "in-loop" if "in Loop" in ln else "-"
        if "scratch_" in ln:
            hits.append((j - i, loop, ln.strip().split()[0]))
        j += 1
    if hits:
        inloop = sum(1 for h in hits if h[1] != "-")
        print(f"{name[:120]}: {len(hits)} scratch ops, {inloop} inside loops")
        for h in hits[:12]:
            print("   ", h)
    i = j
