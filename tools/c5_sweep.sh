#!/bin/bash
# Development aid (ON THE GPU BOX): config 5 over PTGPU_READY (work sharing: lanes without work before subtrees change hands) and PTGPU_DRAIN
# on the -DPT_DEVKNOBS build in _build_dev, then the section shares of the -DPT_SECTIONS build in _build_sec.
B="python tools/bq.py --no-extras --preset perlin_spheres --bvh --width 1920 --height 1080 --samples 128 --steps 3 --warmup 1"
export PTGPU_BUILD_DIR=_build
export PTGPU_BUILD_DIR=_build_dev
for r in ${READY:-1 2 4 8 12 16 24 32 100}; do echo "ready $r: $(PTGPU_READY=$r $B)"; done
for r in ${DRAIN:-2 3}; do echo "drain $r: $(PTGPU_DRAIN=$r $B)"; done
export PTGPU_BUILD_DIR=_build_sec
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --preset perlin_spheres --bvh --width 1920 --height 1080 --samples 128 2>&1 | grep -v "^{" | tail -2
export PTGPU_BUILD_DIR=_build
