#!/bin/bash
# sweep of one environment knob of the library on the bench kernel split: tools/r3_envsweep.sh VAR "v1 v2 ..." [configs]
VAR=$1; VALS=$2
for cfg in ${CONFIGS:-cornell_teapot:1080x1080:640 cornell_textured:1080x1080:640 sr2:3840x2160:128 cubesplosion:3840x2160:128}; do
sc=${cfg%%:*}; rest=${cfg#*:}; sz=${rest%%:*}; spp=${rest#*:}
for v in $VALS; do
  env $VAR=$v timeout -k 10 200 python -u bench.py --scene $sc --size $sz --spp-per-step $spp --steps 3 --warmup 1 --no-cpu-baseline --no-config3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('$sc $VAR=$v  %7.1f M samples/s  extend %6.1f shadow %6.1f shade %6.1f start %5.1f resolve+acc %5.1f'%(d['value']/1e6,k['extend'],k['shadow'],k['shade'],k['generate'],k['resolve_accumulate']))"
done
done
