#!/usr/bin/env python3
"""Prints the kernels of the last N dispatches of a rocprofv3 kernel trace with start
offsets, durations and the gaps between them (usage: timeline.py <kernel_trace.csv> [N])."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    name = name[name.find("k_"):][:50] if "k_" in name else name[:50]
    print(f"{(s - t0) / 1e3:10.1f}us  dur {(e - s) / 1e3:9.1f}us  gap {(s - prev_end) / 1e3:8.1f}us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8s} {name}")
    prev_end = max(prev_end, e)
