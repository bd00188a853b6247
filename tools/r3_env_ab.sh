#!/bin/bash
# alternating A/B of environment settings with the product library: tools/r3_env_ab.sh <tag> <reps> "VAR=a" "VAR=b" ...  (CONFIGS as in r3_ab_many.sh)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
out=$REPO/gpurun_out/$1; mkdir -p $out; reps=$2; shift 2
for rep in $(seq 1 $reps); do
for setting in "$@"; do
  n=$(echo "$setting" | tr ' =' '__')
  for cfg in ${CONFIGS:-cornell_teapot:1080x1080:640}; do
    sc=${cfg%%:*}; rest=${cfg#*:}; sz=${rest%%:*}; spp=${rest#*:}
    env $setting timeout -k 10 200 python -u bench.py --scene $sc --size $sz --spp-per-step $spp --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-config3 > $out/${n}_${sc}_$rep.json 2> $out/${n}_${sc}_$rep.err || { echo "$n $sc FAILED"; tail -3 $out/${n}_${sc}_$rep.err; exit 1; }
  done
done
done
python3 - "$out" "$@" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for setting in sys.argv[2:]:
    n = setting.replace(' ', '_').replace('=', '_')
    by = {}
    for f in sorted(glob.glob('%s/%s_*_*.json' % (out, n))):
        sc = os.path.basename(f)[len(n) + 1:].rsplit('_', 1)[0]
        d = json.load(open(f)); k = d['kernel_ms']
        by.setdefault(sc, []).append((d['value'] / 1e6, k['extend'], k['shadow'], k['shade']))
    for sc, rows in by.items():
        m = [sum(r[i] for r in rows) / len(rows) for i in range(4)]
        print('%-22s %-16s n=%d  %8.1f M samples/s (min %.1f max %.1f)  extend %7.2f shadow %7.2f shade %7.2f ms' % (setting, sc, len(rows), m[0], min(r[0] for r in rows), max(r[0] for r in rows), m[1], m[2], m[3]))
PY
