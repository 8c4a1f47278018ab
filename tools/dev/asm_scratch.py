#!/usr/bin/env python3
"""Where does a kernel touch scratch memory?  usage: asm_scratch.py file.s
Prints every scratch_load / scratch_store with the basic block and loop nest it sits in."""
import re, sys
cur, info = "entry", ""
for i, l in enumerate(open(sys.argv[1])):
    m = re.match(r"^(\.LBB\d+_\d+):\s*;?(.*)", l) or re.match(r"^; (%bb\.\d+):\s*;?(.*)", l)
    if m:
        cur, info = m.group(1), m.group(2).strip()
    elif "scratch_" in l:
        print(f"{i+1:6d} {cur:12s} {info:45s} {l.strip()[:70]}")
