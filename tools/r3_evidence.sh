#!/bin/bash
# Round-3 evidence on the GPU box (outputs under gpurun_out/<tag>/): tools/round_evidence.sh (GPU tests, smoke, bench,
# kernel-trace summary, FETCH/WRITE passes of the headline workload), the full PMC passes for cornell_teapot 1080^2 and for
# sr2 3840x2160 (config 5), the per-config GPU rows, the per-round table, and the other forms of bench.py.
# usage: bash tools/r3_evidence.sh <tag>
set -u
TAG=$1
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
bash tools/round_evidence.sh $TAG || exit 1
echo "== PMC passes, cornell_teapot"
BENCH_ARGS= bash tools/pmc_passes.sh "$OUT/pmc_teapot" || exit 1
cd "$REPO"
python3 tools/pmc_summary.py "$OUT"/pmc_teapot/sq1 "$OUT"/pmc_teapot/sq2 "$OUT"/pmc_teapot/mem "$OUT"/pmc_teapot/lat "$OUT"/pmc_teapot/fetch "$OUT"/pmc_teapot/write > "$OUT/pmc_teapot_summary.txt"
echo "== PMC passes, sr2 3840x2160"
BENCH_ARGS="--scene sr2 --size 3840x2160 --spp-per-step 32" bash tools/pmc_passes.sh "$OUT/pmc_sr2" || exit 1
cd "$REPO"
python3 tools/pmc_summary.py "$OUT"/pmc_sr2/sq1 "$OUT"/pmc_sr2/sq2 "$OUT"/pmc_sr2/mem "$OUT"/pmc_sr2/lat "$OUT"/pmc_sr2/fetch "$OUT"/pmc_sr2/write > "$OUT/pmc_sr2_summary.txt"
timeout -k 10 200 python -u bench.py --scene sr2 --size 3840x2160 --spp-per-step 32 --steps 1 --warmup 1 --no-cpu-baseline --no-config3 --no-events > "$OUT/pmc_sr2_bench.json" 2> "$OUT/pmc_sr2_bench.err" || exit 1
python3 tools/pmc_traffic_json.py sr2@3840x2160 "$OUT/pmc_sr2/fetch" "$OUT/pmc_sr2/write" "$OUT/pmc_sr2_bench.json" "profiles/${TAG}_pmc_sr2_summary.txt" "dependent-load latency: waves waiting ~80 % of their cycles, VALU busy under 40 % (profiles/${TAG}_pmc_sr2_summary.txt)" > "$OUT/traffic_entry_sr2.txt" && cp profiles/hbm_traffic.json "$OUT/hbm_traffic.json"
find "$OUT" -name '*.csv' -size +1M -delete
echo "== rows"
timeout -k 10 300 python tools/baseline_rows.py --no-cpu > "$OUT/rows.jsonl" 2> "$OUT/rows.err" || exit 1
echo "== per round"
timeout -k 10 120 python tools/round_times.py cornell_teapot 1080x1080 640 > "$OUT/per_round_teapot_640spp.txt" 2>&1 || exit 1
timeout -k 10 120 python tools/round_times.py sr2 3840x2160 128 > "$OUT/per_round_sr2_128spp.txt" 2>&1 || exit 1
echo "== bench forms"
timeout -k 10 200 python bench.py --gpus 1 --force-dist --steps 2 --warmup 1 --no-cpu-baseline --no-config3 > "$OUT/bench_force_dist.json" 2> "$OUT/bench_force_dist.err" || exit 1
timeout -k 10 200 python bench.py --gpus 1 --in-process --steps 2 --warmup 1 > "$OUT/bench_inproc1.json" 2> "$OUT/bench_inproc1.err" || exit 1
timeout -k 10 200 python bench.py --gpus 2 --in-process --share-device --steps 2 --warmup 1 --spp-per-step 256 > "$OUT/bench_inproc2_shared.json" 2> "$OUT/bench_inproc2_shared.err" || exit 1
timeout -k 10 200 python tools/launch_floor.py cornell > "$OUT/launch_floor.txt" 2>&1 || exit 1
echo evidence done
