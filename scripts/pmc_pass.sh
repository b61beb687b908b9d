#!/bin/bash
# One rocprofv3 counter pass over a bench.py configuration on the GPU box; prints the counters
# per kernel (averaged over the launches).   usage: scripts/pmc_pass.sh "<counters>" <bench args...>
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$$
CTRS=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT" -- python3 "$R/bench.py" "$@" > "$OUT.json" 2> "$OUT.log" || { echo "pass failed"; tail -5 "$OUT.log"; rm -rf "$OUT"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in acc:
    if "synth" in k or "col_stats" in k: continue
    print(k, "launches", len(n[k]))
    for c, v in sorted(acc[k].items()): print(f"    {c:28s} {v / len(n[k]):16.0f}")
PY
rm -rf "$OUT" "$OUT.json" "$OUT.log"
