#!/bin/bash
# Host code of the library (loader with the parallel OBJ parse, threaded SAH builder, tree JSON / statistics, device tree layout, PNG,
# checkpoint, tone map, scalar traversal) under AddressSanitizer + UBSan and under ThreadSanitizer, on every shipped
# scene.  CPU only (the GPU pool offers no device sanitizer); run from the repository root: bash tools/host_sanitize.sh
set -e
SRC="tools/host_sanitize.cpp tools/host_sanitize_stubs.cpp pathtracer_amd/csrc/host_scene.cpp pathtracer_amd/csrc/host_build.cpp pathtracer_amd/csrc/host_png.cpp pathtracer_amd/csrc/host_format.cpp pathtracer_amd/csrc/host_devtree.cpp"
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off $SRC -o /tmp/ptx_host_asan -lz -pthread
/tmp/ptx_host_asan | tail -1
g++ -std=c++17 -O1 -g -fsanitize=thread -ffp-contract=off $SRC -o /tmp/ptx_host_tsan -lz -pthread
/tmp/ptx_host_tsan --quick | tail -1
