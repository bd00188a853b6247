#!/bin/bash
# Collects PMC counters for the bench in separate rocprofv3 passes (one pass per counter group; never
# combined with tracing).  Usage on the GPU box: bash tools/pmc_passes.sh <outdir> [bench args...]
set -u
OUT=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$REPO/bench.py" --steps 1 --warmup 1 --spp-per-step 96 --no-cpu-baseline --no-config3 --no-events ${BENCH_ARGS:-} > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
}
if [ "${PMC_ONLY:-}" != mem ]; then
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU || exit 1
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_VMEM_WR TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE || exit 1
fi
run mem TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE || exit 1
run lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_REQ_sum TCC_BUSY_sum SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE || exit 1
if [ "${PMC_ONLY:-}" = mem ]; then echo passes done; exit 0; fi
run fetch FETCH_SIZE || exit 1
run write WRITE_SIZE || exit 1
echo passes done
