#!/bin/bash
# PMC passes over bench.py on the GPU box, one rocprofv3 run per pass (rocprofv3 does not
# split counters over passes).  usage: scripts/rocprof_pmc.sh <tag> <bench args...>
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { # name, rocprof args...
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 "$R/bench.py" "${ARGS[@]}" > "$OUT/$name.log" 2>&1 || { echo "$name failed"; tail -5 "$OUT/$name.log"; return 1; }
  echo "$name done"
}
ARGS=("$@")
run trace --kernel-trace --stats || exit 1
run pmc1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU || exit 1
run pmc2 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY || exit 1
run pmc3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE || exit 1
run pmc4 --pmc WRITE_SIZE || exit 1
