#!/bin/bash
# A/B of a probe library against the production one with correctness first: the image of a 1080x1080 / 8 spp render must
# have the production library's bytes.  usage: tools/r3_probe_ab.sh <tag> <probe.so> [scene ...]
set -e
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; PROBE=$2; shift 2
mkdir -p $REPO/gpurun_out/$TAG
for scene in ${@:-cornell_teapot cornell_textured sr2}; do
  for lib in "" "$REPO/$PROBE"; do
    PTX_LIB=$lib timeout -k 10 100 python - $scene <<'PY'
import os, sys, hashlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import pathtracer_amd as pt
name = sys.argv[1]
scene = pt.LoadedScene("scenes/%s.obj" % name)
dev = scene.upload()
w, h = (640, 360) if name in ("sr2", "cubesplosion") else (512, 512)
img = dev.render(scene.pinhole(w, h), 8, seed=3)
print(name, os.environ.get("PTX_LIB") or "production", hashlib.sha256(img.tobytes()).hexdigest()[:16])
PY
  done
done
CONFIGS="cornell_teapot:1080x1080:640 cornell_textured:1080x1080:640 sr2:3840x2160:128" bash tools/r3_variants.sh $TAG pathtracer_amd/lib/libptx_mi355x.so $PROBE
