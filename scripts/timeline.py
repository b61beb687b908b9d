"""Prints the kernels of the last step of a traced bench run in start order, with the gap
before each: python3 scripts/timeline.py <dir with *_kernel_trace.csv> [n_last]"""
import csv, glob, sys

d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
rows = rows[-n_last:]
prev = None
for s, e, name in rows:
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{gap:9.1f} us gap  {(e - s) / 1e3:9.1f} us  {name}")
    prev = e
