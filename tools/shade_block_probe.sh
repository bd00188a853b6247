#!/bin/bash
# probe: k_shade workgroup size (-DPTX_BLOCK variants built with tools/build_variants.sh) at a stated number of
# workgroups per CU.  usage (GPU box): bash tools/shade_block_probe.sh "name lib per_cu" ...
OUT=gpurun_out/r2blk; mkdir -p $OUT
V=$PWD/pathtracer_amd/lib/variants
run() { # name lib per_cu
  PTX_LIB=$V/libptx_$2.so PTX_SHADE_PER_CU=$3 timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config3 > $OUT/bench_$1.json 2> $OUT/bench_$1.err || { echo "$1 FAILED"; tail -3 $OUT/bench_$1.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$OUT/bench_$1.json')); k=d['kernel_ms']
print('%-16s samples/s %6.1fM  extend %6.1f shadow %6.1f shade %6.1f start %5.1f resolve+acc %5.1f ms'%('$1',d['value']/1e6,k['extend'],k['shadow'],k['shade'],k['generate'],k['resolve_accumulate']))" | tee -a $OUT/variants.txt
}
for spec in "$@"; do run $spec || exit 1; done
