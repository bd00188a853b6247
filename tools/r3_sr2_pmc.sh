#!/bin/bash
# sr2 at 3840x2160 (config 5): kernel split of the bench with two libraries (A = $2, B = production), then the PMC
# passes of tools/pmc_passes.sh for both (fetch / write only for B).  usage: bash tools/r3_sr2_pmc.sh <tag> <A.so>
set -e
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
out=$REPO/gpurun_out/$1; mkdir -p $out
A=$REPO/$2
for scene in sr2 cubesplosion; do
  PTX_LIB=$A timeout -k 10 200 python -u bench.py --scene $scene --size 3840x2160 --spp-per-step 128 --steps 3 --warmup 1 --no-cpu-baseline --no-config3 > $out/bench_A_$scene.json 2> $out/bench_A_$scene.err
  timeout -k 10 200 python -u bench.py --scene $scene --size 3840x2160 --spp-per-step 128 --steps 3 --warmup 1 --no-cpu-baseline --no-config3 > $out/bench_B_$scene.json 2> $out/bench_B_$scene.err
done
python3 - <<PY
import json
for scene in ("sr2", "cubesplosion"):
    for v in "AB":
        d = json.load(open("$out/bench_%s_%s.json" % (v, scene))); k = d["kernel_ms"]
        print("%s %-13s %7.1f M samples/s  extend %6.1f shadow %6.1f shade %6.1f start %5.1f resolve+acc %5.1f ms  ext %.0f sh %.0f Mrays/s" % (v, scene, d["value"] / 1e6, k["extend"], k["shadow"], k["shade"], k["generate"], k["resolve_accumulate"], d["extend_mrays_per_s"], d["shadow_mrays_per_s"]))
PY
export BENCH_ARGS="--scene sr2 --size 3840x2160 --spp-per-step 32"
PMC_SKIP_MEMSIZE=1 PTX_LIB=$A bash tools/pmc_passes.sh $out/pmc_A
bash tools/pmc_passes.sh $out/pmc_B
cd $REPO
python3 tools/pmc_summary.py $out/pmc_A/sq1 $out/pmc_A/sq2 $out/pmc_A/mem $out/pmc_A/lat $out/pmc_A/fetch $out/pmc_A/write > $out/pmc_A_summary.txt
python3 tools/pmc_summary.py $out/pmc_B/sq1 $out/pmc_B/sq2 $out/pmc_B/mem $out/pmc_B/lat $out/pmc_B/fetch $out/pmc_B/write > $out/pmc_B_summary.txt
find $out -name '*.csv' -size +1M -delete
