#!/bin/bash
# Development aid (ON THE GPU BOX): config 5 on the shipped library and on the variants in pathtrace-rs_amd/_build_*/ named on the command line, alternating.
B="python tools/bq.py --no-extras --preset perlin_spheres --bvh --width 1920 --height 1080 --samples 128 --steps 3 --warmup 1"
export PTGPU_BUILD_DIR=_build
for rep in 1 2; do
  for d in cur "$@"; do
    if [ $d = cur ]; then export PTGPU_BUILD_DIR=_build; else export PTGPU_BUILD_DIR=_build_$d; fi
    echo "$d: $($B | cut -c100-200)"
  done
done
export PTGPU_BUILD_DIR=_build
