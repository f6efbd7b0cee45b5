#!/bin/bash
# Development aid: config 5 over PTGPU_READY (finished lanes a wave waits for before it leaves the traversal loop) and PTGPU_DRAIN
# (queued leaf candidates that trigger a drain) on a -DPT_DEVKNOBS build in pathtrace-rs_amd/_build_dev (PTGPU_BUILD_DIR).
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],2))'
B="python bench.py --preset perlin_spheres --bvh --width 1920 --height 1080 --samples 128 --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
export PTGPU_BUILD_DIR=_build
export PTGPU_BUILD_DIR=_build_dev   # (make -C pathtrace-rs_amd B=_build_dev DEFS=-DPT_DEVKNOBS)
$B 2>/dev/null | python -c "$P" c5_default
for r in 24 32 40 48 60 64; do PTGPU_READY=$r $B 2>/dev/null | python -c "$P" c5_ready_$r; done
for r in 1 2 3 4; do PTGPU_DRAIN=$r $B 2>/dev/null | python -c "$P" c5_drain_$r; done
export PTGPU_BUILD_DIR=_build
