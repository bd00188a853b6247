#!/bin/bash
# one SQ counter pass of the bench per library: tools/r3_pmc_ab.sh <tag> lib...   (instruction counts and waits of the kernels, side by side)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for so in "$@"; do
  n=$(basename $so .so)
  export PTX_LIB=$REPO/$so
  timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/$n" -- python3 "$REPO/bench.py" --steps 1 --warmup 1 --spp-per-step 96 --no-cpu-baseline --no-config3 --no-events ${BENCH_ARGS:-} > "$OUT/$n.log" 2>&1 || { echo "pass $n failed"; tail -5 "$OUT/$n.log"; exit 1; }
  echo "== $n"
  python3 "$REPO/tools/pmc_summary.py" "$OUT/$n" | grep "k_traverse\|k_shade" | cut -c1-400
done
