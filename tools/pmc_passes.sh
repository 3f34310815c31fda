#!/bin/bash
# Collect PMC counters for bench.py in separate rocprofv3 passes (no tracing options: the pool
# refuses --pmc together with trace domains).  Usage: tools/pmc_passes.sh <tag> [bench args...]
# Output: gpurun_out/pmc_<tag>/pass*/..._counter_collection.csv
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
ARGS="--steps 4 --warmup 2 --repeats 1 --no-cpu-baseline $*"
run() {
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python3 bench.py $ARGS > "$OUT/$name.json" 2> "$OUT/$name.err" || { echo "pass $name failed"; tail -3 "$OUT/$name.err"; return 1; }
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU &&
run sq2 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE GRBM_GUI_ACTIVE
echo done
