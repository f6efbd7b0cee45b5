#!/bin/bash
# Development aid: config 3 / config 5 over PTGPU_REFILL (lanes that must be waiting before a wave refills) on a -DPT_DEVKNOBS
# build in pathtrace-rs_amd/_build_dev (PTGPU_BUILD_DIR).
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],3))'
export PTGPU_BUILD_DIR=_build
export PTGPU_BUILD_DIR=_build_dev   # (make -C pathtrace-rs_amd B=_build_dev DEFS=-DPT_DEVKNOBS)
for r in 4 6 8 10 12 16 20 24; do
  PTGPU_REFILL=$r python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "$P" c3_refill_$r
done
for r in 2 4 6 8 12; do
  PTGPU_REFILL=$r python bench.py --preset perlin_spheres --bvh --width 1920 --height 1080 --samples 128 --steps 4 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "$P" c5_refill_$r
done
export PTGPU_BUILD_DIR=_build
