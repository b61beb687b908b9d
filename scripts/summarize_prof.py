#!/usr/bin/env python3
"""Summarises the CSVs written by scripts/rocprof_bench.sh: per-kernel time from the
kernel trace and per-kernel counter averages from the PMC passes."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name: str) -> str:
    for key in ("k_part_hist", "k_part_scatter", "k_part_agg", "k_part_offsets", "k_part_plan", "k_group_by", "k_cmp_leaf", "k_agg_global", "k_emit", "k_fill", "k_contains", "k_str_leaf", "k_lut",
                "k_col_stats", "k_synth", "k_bitmap", "k_tile_counts", "k_scan_counts", "k_write_selvec", "k_validity",
                "k_distinct_fold"):
        if key in name:
            return key
    return name[:60]


def main(root: str) -> None:
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
        print("== kernel stats:", os.path.relpath(f, root))
        for row in csv.DictReader(open(f)):
            print(f"  {short(row['Name']):18s} calls={row['Calls']:>6s} total_ns={row['TotalDurationNs']:>14s} "
                  f"avg_ns={row['AverageNs']:>12s} pct={row['Percentage']}")
    for f in sorted(glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("== counters:", os.path.relpath(f, root))
        for k, cs in acc.items():
            if k in ("k_synth", "k_col_stats"):
                continue
            print("  " + k + ": " + "  ".join(f"{c}={sum(v) / len(v):.4g}(n={len(v)})" for c, v in sorted(cs.items())))


if __name__ == "__main__":
    main(sys.argv[1])
