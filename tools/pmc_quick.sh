#!/bin/bash
# quick counter passes after a kernel change: L2 hit/miss and HBM bytes of the 96-spp bench.  usage: bash tools/pmc_quick.sh <outdir>
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/$1; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$REPO/bench.py" --steps 1 --warmup 1 --spp-per-step 96 --no-cpu-baseline --no-config3 --no-events > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
}
run l2 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE || exit 1
run fetch FETCH_SIZE || exit 1
run write WRITE_SIZE || exit 1
cd "$REPO" && python3 tools/pmc_summary.py "$OUT/l2" "$OUT/fetch" "$OUT/write" > "$OUT/summary.txt" && cut -c1-220 "$OUT/summary.txt"
