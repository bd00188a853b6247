#!/bin/bash
# where do the 5 ms of config 1 (cornell 256x256, 16 spp) go?  kernel-trace summary of a few such steps
OUT=$PWD/gpurun_out/small; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --scene cornell --size 256x256 --spp-per-step 16 --steps 20 --warmup 2 --no-cpu-baseline --no-config3 > $OUT/bench.json 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
find $OUT/trace -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/trace -name '*kernel_trace.csv' -exec cp {} $OUT/kernel_trace.csv \;
head -12 $OUT/kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
python3 -c "
import json; d=json.load(open('$OUT/bench.json')); print('ms/step', d['ms_per_step'], 'kernel ms total over steps', d['kernel_ms'])"
