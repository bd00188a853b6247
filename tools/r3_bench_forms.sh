#!/bin/bash
# Round 3: the three forms of bench.py on a one-GPU box (default, one-rank RCCL rehearsal, in-process ptx_render_multi
# with one handle and with two handles sharing the GPU), then the per-config GPU rows.  Output: gpurun_out/$1/
set -e
out=gpurun_out/${1:-r3a}
mkdir -p $out
timeout -k 10 300 python bench.py --steps 4 --warmup 1 > $out/bench.json 2> $out/bench.err
timeout -k 10 200 python bench.py --gpus 1 --force-dist --steps 2 --warmup 1 --no-cpu-baseline --no-config3 > $out/bench_force_dist.json 2> $out/bench_force_dist.err
timeout -k 10 200 python bench.py --gpus 1 --in-process --steps 2 --warmup 1 > $out/bench_inproc1.json 2> $out/bench_inproc1.err
timeout -k 10 200 python bench.py --gpus 2 --in-process --share-device --steps 2 --warmup 1 --spp-per-step 256 > $out/bench_inproc2_shared.json 2> $out/bench_inproc2_shared.err
timeout -k 10 300 python tools/baseline_rows.py --no-cpu > $out/rows.jsonl 2> $out/rows.err
