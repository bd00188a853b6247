#!/bin/bash
# bench kernel split for every library given (paths relative to the repo) on teapot 1080^2 and sr2 4K; usage: tools/r3_variants.sh <tag> lib...
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
out=$REPO/gpurun_out/$1; mkdir -p $out; shift
for rep in 1 2; do
for so in "$@"; do
  n=$(basename $so .so)
  for cfg in ${CONFIGS:-cornell_teapot:1080x1080:640 sr2:3840x2160:128 cubesplosion:3840x2160:128}; do
    sc=${cfg%%:*}; rest=${cfg#*:}; sz=${rest%%:*}; spp=${rest#*:}
    PTX_LIB=$REPO/$so timeout -k 10 200 python -u bench.py --scene $sc --size $sz --spp-per-step $spp --steps 3 --warmup 1 --no-cpu-baseline --no-config3 > $out/${n}_${sc}_$rep.json 2> $out/${n}_${sc}_$rep.err || { echo "$n $sc FAILED"; tail -3 $out/${n}_${sc}_$rep.err; exit 1; }
    python3 -c "
import json
d=json.load(open('$out/${n}_${sc}_$rep.json')); k=d['kernel_ms']
print('%-22s %-14s %7.1f M samples/s  extend %6.1f shadow %6.1f shade %6.1f start %5.1f resolve+acc %5.1f ms  camera %.0f bounced %.0f shadow %.0f Mrays/s'%('$n','$sc',d['value']/1e6,k['extend'],k['shadow'],k['shade'],k['generate'],k['resolve_accumulate'],d.get('extend_camera_mrays_per_s',0),d.get('extend_bounced_mrays_per_s',0),d['shadow_mrays_per_s']))"
  done
done
done
