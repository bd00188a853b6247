#!/bin/bash
# default bench for every variant library + the production one, with the kernel split; usage: tools/run_variants3.sh
for rep in 1 2; do
for so in pathtracer_amd/lib/variants/libptx_*.so pathtracer_amd/lib/libptx_mi355x.so; do
  n=$(basename $so .so)
  PTX_LIB=$PWD/$so timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/var3_$n.json 2> gpurun_out/var3_$n.err || { echo "$n FAILED"; tail -3 gpurun_out/var3_$n.err; continue; }
  python3 -c "
import json
d=json.load(open('gpurun_out/var3_$n.json')); k=d['kernel_ms']
print('%-22s samples/s %6.1fM  extend %6.1f shadow %6.1f shade %6.1f start %5.1f resolve+acc %5.1f ms'%('$n',d['value']/1e6,k['extend'],k['shadow'],k['shade'],k['generate'],k['resolve_accumulate']))"
done
done
