#!/bin/bash
# Stages an UNTRACKED copy of the reference's Python (its scripts and pure-Python packages only -- none of the CUDA / C++
# extension sources this repo replaces) at reference_scratch/ (git-ignored), so that a gpurun snapshot carries it to the GPU box,
# where /root/reference does not exist.  Used by tools/gpu_job_reference_train.sh; nothing under reference_scratch/ is ever
# committed.      usage: tools/stage_reference_scratch.sh [/path/to/Relightable3DGaussian]
set -e
REF="${1:-/root/reference}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
DST="$ROOT/reference_scratch"
rm -rf "$DST"
mkdir -p "$DST/bvh" "$DST/env_map"
cp "$REF"/*.py "$DST"/
for d in arguments gaussian_renderer lpipsPyTorch utils scene; do
    (cd "$REF" && find "$d" -name '*.py' -print0 | xargs -0 -I{} cp --parents {} "$DST"/)
done
cp "$REF/bvh/__init__.py" "$DST/bvh/"
cp "$REF/env_map/envmap3.png" "$DST/env_map/"
find "$DST" -type f | wc -l
du -sh "$DST"
