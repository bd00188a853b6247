#!/bin/bash
# queue segments per resident k_shade workgroup slot: PTX_SEG_ROUNDS sweep on the bench kernel split
for cfg in "cornell_teapot 1080x1080 640" "cornell_textured 1080x1080 640" "sr2 3840x2160 128" "cornell 256x256 16"; do
set -- $cfg
for r in ${ROUNDS:-1 2 3 4 8}; do
  PTX_SEG_ROUNDS=$r timeout -k 10 200 python -u bench.py --scene $1 --size $2 --spp-per-step $3 --steps 3 --warmup 1 --no-cpu-baseline --no-config3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('$1 seg_rounds=$r  %7.1f M samples/s  extend %6.1f shadow %6.1f shade %6.1f start %5.1f resolve+acc %5.1f'%(d['value']/1e6,k['extend'],k['shadow'],k['shade'],k['generate'],k['resolve_accumulate']))"
done
done
