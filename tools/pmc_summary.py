#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv for the transport kernel.
usage: pmc_summary.py <counter_collection.csv> <events in the launch>"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_transport" in r["Kernel_Name"]]
ev = float(sys.argv[2])
c = {}
for r in rows:
    c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
dur = (int(rows[0]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) * 1e-9
we = ev / 64
print(f"kernel {dur*1e3:.2f} ms  VGPR {rows[0]['VGPR_Count']} SGPR {rows[0]['SGPR_Count']}  waves {c.get('SQ_WAVES')}")
for k, v in sorted(c.items()):
    print(f"  {k:28s} {v:.4e}   per wave-event {v / we:10.1f}")
if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
    print("  lane utilisation of VALU cycles:", c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"]))
if "SQ_WAVE_CYCLES" in c:
    for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
        if k in c:
            print(f"  {k}/SQ_WAVE_CYCLES = {c[k] / c['SQ_WAVE_CYCLES']:.3f}")
