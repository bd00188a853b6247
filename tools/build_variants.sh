#!/bin/bash
# builds tuning variants of the HIP library: tools/build_variants.sh name "DEFS" [name "DEFS" ...]
set -e
cd "$(dirname "$0")/../pathtracer_amd/csrc"
while [ $# -gt 1 ]; do
  name=$1; defs=$2; shift 2
  make -s OUT=../lib/variants/libptx_$name.so OBJDIR=../lib/variants/obj_$name DEFS="$defs" &
done
wait
ls ../lib/variants/*.so
