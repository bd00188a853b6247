#!/bin/bash
# the same library with the short and with the long division in the node step (PTX_LONG_DIV=1); usage: tools/r3_longdiv_ab.sh <tag> [lib]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
out=$REPO/gpurun_out/$1; mkdir -p $out
lib=${2:-pathtracer_amd/lib/libptx_mi355x.so}
for rep in 1 2; do
for mode in 0 1; do
  for cfg in ${CONFIGS:-cornell_teapot:1080x1080:640}; do
    sc=${cfg%%:*}; rest=${cfg#*:}; sz=${rest%%:*}; spp=${rest#*:}
    PTX_LONG_DIV=$mode PTX_LIB=$REPO/$lib timeout -k 10 200 python -u bench.py --scene $sc --size $sz --spp-per-step $spp --steps 3 --warmup 1 --no-cpu-baseline --no-config3 > $out/ld${mode}_${sc}_$rep.json 2> $out/ld${mode}_${sc}_$rep.err || { echo "$mode $sc FAILED"; tail -3 $out/ld${mode}_${sc}_$rep.err; exit 1; }
    python3 -c "
import json
d=json.load(open('$out/ld${mode}_${sc}_$rep.json')); k=d['kernel_ms']
print('long_div=%s %-14s %7.1f M samples/s  extend %6.1f shadow %6.1f shade %6.1f ms  camera %.0f bounced %.0f shadow %.0f Mrays/s'%('$mode','$sc',d['value']/1e6,k['extend'],k['shadow'],k['shade'],d.get('extend_camera_mrays_per_s',0),d.get('extend_bounced_mrays_per_s',0),d['shadow_mrays_per_s']))"
  done
done
done
