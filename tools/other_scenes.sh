#!/bin/bash
# 4K sanity runs of the other shipped scenes with wide batches; usage (GPU box): bash tools/other_scenes.sh
for n in sr2 cubesplosion cornell_textured cornell; do
  timeout -k 10 300 python -u bench.py --scene $n --size 3840x2160 --steps 2 --warmup 1 --spp-per-step 64 --no-cpu-baseline --save gpurun_out/big_$n.png > gpurun_out/big_$n.json 2> gpurun_out/big_$n.err || { echo "$n FAILED"; tail -5 gpurun_out/big_$n.err; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/big_$n.json'))
print('%-18s %7.1f M samples/s  extend %6.0f shadow %6.0f Mrays/s  %s'%('$n',d['value']/1e6,d['extend_mrays_per_s'],d['shadow_mrays_per_s'] or 0,d['config']['workload']))"
done
