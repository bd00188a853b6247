#!/bin/bash
# A/B of two libraries on the per-config GPU rows: tools/r3_ab_rows.sh <tag> <variant.so> (the production library is B)
set -e
out=gpurun_out/${1}; mkdir -p $out
for rep in 1 2; do
PTX_LIB=$PWD/$2 timeout -k 10 300 python tools/baseline_rows.py --no-cpu > $out/rows_A_$rep.jsonl 2> $out/rows_A_$rep.err
timeout -k 10 300 python tools/baseline_rows.py --no-cpu > $out/rows_B_$rep.jsonl 2> $out/rows_B_$rep.err
done
python3 - <<PY
import json
for rep in (1,2):
  for v in "AB":
    for l in open("$out/rows_%s_%d.jsonl"%(v,rep)):
        r=json.loads(l); g=r['gpu']
        print(v, rep, r['run'], '%-16s %.3f G samples/s ext %8.1f sh %8.1f Mrays/s'%(r['scene'], g['samples_per_s']/1e9, g['extend_mrays_per_s'], g['shadow_mrays_per_s']))
PY
