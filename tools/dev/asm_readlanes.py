#!/usr/bin/env python3
"""Where does a kernel fetch parked scalar registers back (v_readlane / v_writelane)?
usage: asm_readlanes.py file.s <kernel-name-substring>
Prints, for the innermost loop with the most basic blocks (the event loop of the tracking
kernels): VALU instructions and v_readlane/v_writelane per basic block."""
import re, sys, collections
src = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith("_ZN") and pat in l and re.match(r"^_ZN\S+:", l))
end = next(i for i in range(start, len(src)) if src[i].strip().startswith("s_endpgm"))
loops = collections.Counter()
for i in range(start, end):
    m = re.search(r"in Loop: Header=(\S+) Depth=2", src[i])
    if m: loops[m.group(1)] += 1
hdr = loops.most_common(1)[0][0]
blk = None; stats = collections.OrderedDict()
for i in range(start, end):
    l = src[i]
    m = re.match(r"^(\.LBB\d+_\d+|; %bb\.\d+):", l)
    if m:
        blk = m.group(1) if ("Header=%s " % hdr) in l or (m.group(1) == "." + hdr[0:] ) else None
        if ("Header=%s " % hdr) in l: stats[blk] = [0, 0]
        continue
    if blk in stats:
        op = l.strip().split()[0] if l.strip() else ""
        if op.startswith("v_"):
            stats[blk][0] += 1
            if op.startswith(("v_readlane", "v_writelane")): stats[blk][1] += 1
tot = [0, 0]
for b, (v, r) in stats.items():
    tot[0] += v; tot[1] += r
    if r: print(f"{b:16s} valu {v:4d} of which readlane/writelane {r}")
print(f"loop {hdr}: {len(stats)} blocks, valu {tot[0]}, readlane/writelane {tot[1]}")
for k in ("sgpr_spill_count", "vgpr_spill_count", ".vgpr_count"):
    for i in range(end, min(end + 400, len(src))):
        if k in src[i]: print(src[i].strip()); break
