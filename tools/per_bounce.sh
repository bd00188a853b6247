#!/bin/bash
# usage (GPU box): bash tools/per_bounce.sh <tag> [lib.so]   -> gpurun_out/pb_<tag>/table.txt
TAG=$1; REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/pb_$TAG; mkdir -p $OUT
[ -n "${2:-}" ] && export PTX_LIB=$REPO/$2
export PB_OUT=$OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $REPO/tools/per_bounce.py cornell_teapot ${PB_SPP:-256} > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
cd $REPO && python3 tools/per_bounce.py --parse $OUT > $OUT/table.txt && rm -rf $OUT/trace && cat $OUT/table.txt
