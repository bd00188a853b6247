#!/bin/bash
# on the GPU box: short bench for every variant library; usage: tools/run_variants.sh [bench args]
for so in pathtracer_amd/lib/variants/libptx_*.so pathtracer_amd/lib/libptx_mi355x.so; do
  n=$(basename $so .so)
  PTX_LIB=$PWD/$so timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --spp-per-step 96 --batch-samples 96 --no-cpu-baseline "$@" > gpurun_out/var_$n.json 2> gpurun_out/var_$n.err || { echo "$n FAILED"; tail -3 gpurun_out/var_$n.err; continue; }
  python3 - <<PY
import json
d=json.load(open('gpurun_out/var_$n.json'))
print('%-28s samples/s %6.1fM extend %5.0f shadow %5.0f Mrays/s  ms/step %6.2f'%('$n',d['value']/1e6,d['extend_mrays_per_s'],d['shadow_mrays_per_s'],d['ms_per_step']))
PY
done
