#!/usr/bin/env python3
"""Instruction counts of the event loop (the Depth=2 loop holding the class-record reads) of a k_ddmc_q kernel in
hipcc -S output, all paths together: usage: asm_loop_valu.py file.s k_ddmc_qILi1ELb1ELb1E"""
import re, sys
src = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith("_ZN") and pat in l and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(src)) if src[i].startswith(".Lfunc_end"))
# loops at depth 2: from an "Inner Loop Header: Depth=2" label to the first following block that is not "in Loop: Header=<that>"
i = start
while i < end:
    if "This Inner Loop Header: Depth=2" in src[i]:
        hdr = re.match(r"^(\.LBB\d+_\d+):", src[i - 1]).group(1)[2:]
        j = i + 1
        cnt = {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "wait/nop": 0, "branch": 0}
        has = False
        while j < end:
            l = src[j].strip()
            m = re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", l)
            if m and ("Header=" + hdr + " ") not in l and ("Header=" + hdr) not in l:
                break
            if l and not l.startswith((";", ".")):
                op = l.split()[0]
                if op.startswith("ds_read_b128"): has = True
                if op.startswith("v_"): cnt["valu"] += 1
                elif op.startswith(("s_waitcnt", "s_nop")): cnt["wait/nop"] += 1
                elif op.startswith(("s_cbranch", "s_branch")): cnt["branch"] += 1
                elif op.startswith("s_"): cnt["salu"] += 1
                elif op.startswith("ds_"): cnt["lds"] += 1
                else: cnt["vmem"] += 1
            j += 1
        if has:
            print(pat, "loop", hdr, "lines", i - start, "-", j - start, cnt, "total", sum(cnt.values()))
        i = j
    else:
        i += 1
