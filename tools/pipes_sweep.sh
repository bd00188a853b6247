#!/bin/bash
for cfg in "2 0" "3 0" "4 0"; do
  set -- $cfg
  PTX_PIPES=$1 timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --batch-samples $2 --no-cpu-baseline > gpurun_out/p.json 2> gpurun_out/p.err || { echo "pipes=$1 batch=$2 FAILED"; tail -3 gpurun_out/p.err; continue; }
  python3 - <<PY
import json
d=json.load(open('gpurun_out/p.json'))
print('pipes=$1 batch=$2  samples/s %6.1fM extend %5.0f shadow %5.0f Mrays/s (per-kernel)  total rays/s %.2fG ms/step %6.2f'%(d['value']/1e6,d['extend_mrays_per_s'],d['shadow_mrays_per_s'],d['rays_per_s']/1e9,d['ms_per_step']))
PY
done
