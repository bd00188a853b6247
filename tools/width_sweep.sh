#!/bin/bash
# samples/s against the width of a batch (spp per step = spp per batch); usage: tools/width_sweep.sh
for cfg in "768 960" "1024 1280" "1280 1600"; do
  set -- $cfg
  PTX_SLOTS_MI=$2 timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --spp-per-step $1 --batch-samples $1 --no-cpu-baseline > gpurun_out/width_$1.json 2> gpurun_out/width_$1.err || { echo "width $1 FAILED"; tail -3 gpurun_out/width_$1.err; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/width_$1.json'))
print('spp/batch %4d  slots %4d Mi  samples/s %7.1fM  extend %6.0f shadow %6.0f Mrays/s'%($1,$2,d['value']/1e6,d['extend_mrays_per_s'],d['shadow_mrays_per_s']))"
done
