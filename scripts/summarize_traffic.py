#!/usr/bin/env python3
"""Per-kernel summary of one scripts/profile_workload.sh run: calls and average duration from
the kernel trace, HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes (KiB in the
CSV; FETCH_SIZE doubled: gfx950 tallies a 128-byte fabric read request as 64 bytes).
Writes <dir>/traffic.json and prints the table."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name[:48]


def main(root: str, tag: str, bench_args: str) -> None:
    kernels = defaultdict(lambda: {"calls": 0, "total_ns": 0.0})
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = kernels[short(row["Name"])]
            k["calls"] += int(row["Calls"])
            k["total_ns"] += float(row["TotalDurationNs"])
    counters = {}
    for which in ("fetch", "write"):
        acc = defaultdict(list)
        for f in glob.glob(os.path.join(root, which, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
        counters[which] = acc
    bench = {}
    try:
        bench = json.loads(open(os.path.join(root, "trace.json")).read().strip().splitlines()[-1])
    except (OSError, ValueError, IndexError):
        pass
    out = {"tag": tag, "command": "python3 bench.py " + bench_args, "bench_line_under_trace": bench, "kernels": {},
           "note": "avg_ns from rocprofv3 --kernel-trace --stats; fetch/write bytes per launch from separate --pmc passes of the "
                   "same command (FETCH_SIZE and WRITE_SIZE are KiB; FETCH_SIZE x 2: gfx950 counts 64 B per 128-B read request)"}
    total = sum(k["total_ns"] for k in kernels.values()) or 1.0
    print(f"== {tag}: python3 bench.py {bench_args}")
    print(f"{'kernel':28s} {'calls':>7s} {'avg_us':>10s} {'pct':>6s} {'fetch_MB/launch':>16s} {'write_MB/launch':>16s}")
    for name, k in sorted(kernels.items(), key=lambda kv: -kv[1]["total_ns"]):
        if name.startswith(("k_synth", "k_col_stats")) or k["calls"] == 0:
            continue
        fe, wr = counters["fetch"].get(name), counters["write"].get(name)
        fetch = 2 * 1024 * sum(fe) / len(fe) if fe else None
        write = 1024 * sum(wr) / len(wr) if wr else None
        avg = k["total_ns"] / k["calls"]
        out["kernels"][name] = {"calls": k["calls"], "avg_ns": avg, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write}
        print(f"{name:28s} {k['calls']:7d} {avg / 1e3:10.1f} {100 * k['total_ns'] / total:6.2f} "
              f"{(fetch or 0) / 1e6:16.2f} {(write or 0) / 1e6:16.2f}")
    json.dump(out, open(os.path.join(root, "traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
