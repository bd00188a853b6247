#!/bin/bash
# parity tests then a short bench; usage: bash tools/gpu_check.sh <tag> [bench args]
TAG=$1; shift
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t_$TAG.log 2>&1
rc=$?; tail -3 gpurun_out/t_$TAG.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { tail -5 gpurun_out/bench_$TAG.err; exit 1; }
python3 - <<PY
import json
d=json.load(open('gpurun_out/bench_$TAG.json'))
print('samples/s %.1fM  extend %.0f Mrays/s  shadow %.0f Mrays/s  ms/step %.2f  kernel_ms %s  frac %.3f'%(d['value']/1e6,d['extend_mrays_per_s'],d['shadow_mrays_per_s'],d['ms_per_step'],{k:round(v,1) for k,v in d['kernel_ms'].items()},d['roofline']['frac']))
PY
