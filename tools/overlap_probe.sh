#!/bin/bash
OUT=gpurun_out/r2overlap; mkdir -p $OUT
run() {
  env $2 timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config3 > $OUT/bench_$1.json 2> $OUT/bench_$1.err || { echo "$1 FAILED"; tail -3 $OUT/bench_$1.err; return 1; }
  python3 -c "
import json
d=json.load(open('$OUT/bench_$1.json')); k=d['kernel_ms']
print('%-12s samples/s %6.1fM  ms/step %.1f extend %6.1f shadow %6.1f shade %6.1f'%('$1',d['value']/1e6,d['ms_per_step'],k['extend'],k['shadow'],k['shade']))"
}
run ref PTX_X=0 && run overlap PTX_OVERLAP=1 && run ref2 PTX_X=0 && run overlap2 PTX_OVERLAP=1
