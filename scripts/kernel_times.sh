#!/bin/bash
# Per-kernel average durations (rocprofv3 --kernel-trace --stats) of one bench.py configuration
# on the GPU box, printed as "calls avg_us name".   usage: scripts/kernel_times.sh <bench args...>
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ktimes_$$
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$R/bench.py" "$@" > "$OUT.json" 2> "$OUT.log" || { echo "profile failed"; tail -5 "$OUT.log"; rm -rf "$OUT"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(f"{int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:10.1f}  {r['Name'][:90]}")
PY
python3 -c "
import json,sys
d=json.load(open('$OUT.json')); print('ms_per_step', d['ms_per_step'])"
rm -rf "$OUT" "$OUT.json" "$OUT.log"
