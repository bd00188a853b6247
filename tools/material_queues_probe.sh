#!/bin/bash
# f-1 second half: material-keyed shade queues (PTX_FLAG_MATERIAL_QUEUES) against queue order; times, then the VALU lane
# utilisation of k_shade from one PMC pass each.  usage (GPU box): bash tools/material_queues_probe.sh <outdir>
OUT=$PWD/$1; mkdir -p $OUT; REPO=$PWD
for scene in cornell_textured cornell_teapot sr2; do
  for f in "" "--material-queues"; do
    timeout -k 10 120 python -u tools/lobe_probe.py $scene 256 $f 2>/dev/null | tee -a $OUT/times.txt || exit 1
  done
done
cd /tmp && export TMPDIR=/tmp
for tag in plain keyed; do
  f=""; [ $tag = keyed ] && f="--material-queues"
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc_$tag -- python3 $REPO/tools/lobe_probe.py cornell_textured 64 $f > $OUT/pmc_$tag.log 2>&1 || { tail -3 $OUT/pmc_$tag.log; exit 1; }
done
cd $REPO && python3 tools/pmc_summary.py $OUT/pmc_plain $OUT/pmc_keyed | grep "==\|k_shade\|k_bucket" | tee $OUT/pmc.txt
