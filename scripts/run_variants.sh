#!/bin/bash
# Runs bench.py once per tuning variant built by scripts/build_variants.sh (on the GPU box).
# usage: scripts/run_variants.sh "<bench args>" name...
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS=$1; shift
for name in "$@"; do
  lib=$R/prism_amd/lib/variants/libprism_gpu_$name.so
  [ "$name" = base ] && lib=$R/prism_amd/lib/libprism_gpu.so
  PRISM_GPU_LIB=$lib timeout -k 10 300 python3 -u $R/bench.py $ARGS --no-cpu-baseline 2>/dev/null \
    | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],3), round(d['roofline']['kernel_avg_ms'],4), round(d['roofline']['frac'],4))" || exit 1
done
