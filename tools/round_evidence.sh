#!/bin/bash
# Round evidence on the GPU box: full GPU test suite, smoke(), the default bench line, the rocprofv3
# kernel-trace summary of the bench, and the FETCH_SIZE / WRITE_SIZE passes (each in a run of its own).
# usage: bash tools/round_evidence.sh <tag>      (outputs under gpurun_out/<tag>/)
set -u
TAG=$1
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > "$OUT/pytest_gpu.log" 2>&1 || { tail -20 "$OUT/pytest_gpu.log"; exit 1; }
tail -2 "$OUT/pytest_gpu.log"
timeout -k 10 200 python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1 || { tail -20 "$OUT/smoke.log"; exit 1; }
tail -1 "$OUT/smoke.log"
timeout -k 10 400 python -u bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
python3 -c "
import json; d=json.load(open('$OUT/bench.json'))
print('samples/s %.1fM ms/step %.1f extend %.0f Mrays/s frac %.3f cpu %.2fM'%(d['value']/1e6,d['ms_per_step'],d['extend_mrays_per_s'],d['roofline']['frac'],d['cpu_baseline']['value']/1e6))"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-config3 > "$OUT/trace_bench.json" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
find "$OUT/trace" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT/trace" -name '*kernel_trace.csv' -delete
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -- python3 "$REPO/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-config3 --no-events > "$OUT/pmc_$c.json" 2> "$OUT/pmc_$c.err" || { tail -5 "$OUT/pmc_$c.err"; exit 1; }
done
cd "$REPO"
python3 tools/pmc_summary.py "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" > "$OUT/pmc_traffic.txt"
python3 tools/pmc_traffic_json.py cornell_teapot@1080x1080 "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" "$OUT/pmc_FETCH_SIZE.json" "profiles/${TAG}_pmc_fetch_write.txt" > "$OUT/traffic_entry.txt" && cp profiles/hbm_traffic.json "$OUT/hbm_traffic.json"
cat "$OUT/pmc_traffic.txt" | cut -c1-200
head -8 "$OUT/kernel_stats.csv"
