#!/bin/bash
# Kernel trace + HBM traffic of one bench.py configuration on the GPU box: three rocprofv3
# runs (kernel trace with stats; FETCH_SIZE; WRITE_SIZE -- the two counters do not fit one
# pass), then a summary and a traffic file under gpurun_out/prof_<tag>/ to be copied into
# profiles/.   usage: scripts/profile_workload.sh <tag> <bench args...>
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS=("$@")
run() { # name, rocprof args...
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 "$R/bench.py" "${ARGS[@]}" > "$OUT/$name.json" 2> "$OUT/$name.log" || { echo "$TAG/$name failed"; tail -5 "$OUT/$name.log"; return 1; }
  echo "$TAG/$name done"
}
run trace --kernel-trace --stats || exit 1
run fetch --pmc FETCH_SIZE || exit 1
run write --pmc WRITE_SIZE || exit 1
python3 "$R/scripts/summarize_traffic.py" "$OUT" "$TAG" "${ARGS[*]}" > "$OUT/summary.txt" || exit 1
# the CSVs are large; the summary is what is kept
rm -rf "$OUT/trace" "$OUT/fetch" "$OUT/write"
cat "$OUT/summary.txt"
