#!/bin/bash
# default-mode (2 pipelines) bench for every variant library, 2 repetitions; usage: tools/run_variants2.sh
for rep in 1 2; do
for so in pathtracer_amd/lib/variants/libptx_*.so pathtracer_amd/lib/libptx_mi355x.so; do
  n=$(basename $so .so)
  PTX_LIB=$PWD/$so timeout -k 10 200 python -u bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-events > gpurun_out/var2_$n.json 2> gpurun_out/var2_$n.err || { echo "$n FAILED"; tail -3 gpurun_out/var2_$n.err; continue; }
  python3 -c "
import json
d=json.load(open('gpurun_out/var2_$n.json'))
print('%-28s samples/s %6.1fM ms/step %6.2f'%('$n',d['value']/1e6,d['ms_per_step']))"
done
done
