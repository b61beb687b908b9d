#!/usr/bin/env python3
"""Prints step time and per-kernel averages of bench.py JSON lines."""
import json
import sys

for path in sys.argv[1:]:
    j = json.loads(open(path).read().strip().splitlines()[-1])
    r = j.get("roofline") or {}
    print(path, f"{j['ms_per_step']:.3f} ms/step", f"{j['value'] / 1e9:.1f} G rows/s", "frac", r.get("frac"))
    print("   ", {k: round(v["avg_ms"], 3) for k, v in (j.get("kernels") or {}).items()})
