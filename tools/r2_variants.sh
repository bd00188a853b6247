#!/bin/bash
# round-2 experiments on the GPU box: default bench for every variant library.  usage: [BENCH_ARGS="--scene sr2 --size 3840x2160 --spp-per-step 96"] bash tools/r2_variants.sh <outdir> [reps]
OUT=$1; mkdir -p $OUT
V=$PWD/pathtracer_amd/lib/variants
for rep in $(seq 1 ${2:-2}); do
for so in $V/libptx_*.so; do
  n=$(basename $so .so); n=${n#libptx_}
  PTX_LIB=$so timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config3 ${BENCH_ARGS:-} > $OUT/bench_$n.json 2> $OUT/bench_$n.err || { echo "$n FAILED"; tail -3 $OUT/bench_$n.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$OUT/bench_$n.json')); k=d['kernel_ms']
print('%-16s samples/s %6.1fM  extend %6.1f (%5.0f Mr/s) shadow %6.1f (%5.0f) shade %6.1f start %5.1f resolve+acc %5.1f ms'%('$n',d['value']/1e6,k['extend'],d['extend_mrays_per_s'],k['shadow'],d['shadow_mrays_per_s'],k['shade'],k['generate'],k['resolve_accumulate']))" | tee -a $OUT/variants.txt
done
done
