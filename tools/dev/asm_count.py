#!/usr/bin/env python3
"""Count instructions per basic block of one kernel in hipcc -S output.
usage: asm_count.py file.s 'k_transportILi3ELb0ELb1ELi2E' [first_line last_line]"""
import re, sys, collections
src = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith("_ZN") and pat in l and re.match(r"^_ZN\S+:", l))
# (a kernel may hold several s_endpgm -- early returns: count up to the end-of-function label)
end = next(i for i in range(start, len(src)) if src[i].startswith(".Lfunc_end"))
blocks = collections.OrderedDict()
cur = "entry"
blocks[cur] = collections.Counter()
for i in range(start + 1, end + 1):
    l = src[i].strip()
    if not l or l.startswith(";") or l.startswith("."):
        m = re.match(r"^(\.LBB\d+_\d+):", l) or re.match(r"^; (%bb\.\d+):", l)
        if m:
            cur = m.group(1); blocks[cur] = collections.Counter(); blocks[cur]["_line"] = i - start
        continue
    m = re.match(r"^; %bb\.(\d+):", l)
    op = l.split()[0]
    if op.startswith("v_"):
        kind = "valu"
        if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")): blocks[cur]["trans"] += 1
        if op.startswith(("v_mul_lo_u32", "v_mad_u64_u32", "v_mul_hi")): blocks[cur]["imul"] += 1
        if op.startswith(("v_cndmask", "v_mov")): blocks[cur]["mov/sel"] += 1
        if op.startswith("v_cmp"): blocks[cur]["cmp"] += 1
        if "f64" in op: blocks[cur]["f64"] += 1
    elif op.startswith("s_"):
        kind = "salu"
    elif op.startswith("ds_"):
        kind = "lds"
    elif op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        kind = "vmem"
    else:
        kind = "other"
    blocks[cur][kind] += 1
for b, c in blocks.items():
    if sum(v for k, v in c.items() if k != "_line") == 0: continue
    print(f"{b:14s} line {c.get('_line',0):5d}  valu {c['valu']:4d} (f64 {c['f64']:3d} trans {c['trans']:2d} imul {c['imul']:2d} sel/mov {c['mov/sel']:3d} cmp {c['cmp']:3d})  salu {c['salu']:3d} lds {c['lds']:2d} vmem {c['vmem']:2d}")
tot = collections.Counter()
for c in blocks.values(): tot.update(c)
print("total", dict(tot))
