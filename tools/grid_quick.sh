#!/bin/bash
# Development aid (ON THE GPU BOX): the uniform cell grid on config 5 -- parity slice, bench lines, visits per ray, and (with the
# -DPT_GRID_ROUNDS build in _build_dev) wave-level rounds per call of grid_trace.
timeout 900 python -m pytest tests -m gpu -x -q -k "golden_fixture or noise_parity or tree or bvh_world or far_ray or fuzz" 2>&1 | tail -4
for i in 1 2; do timeout 200 python tools/bq.py --no-extras --preset perlin_spheres --bvh --width 1920 --height 1080 --samples 128 --steps 4 --warmup 1; done
timeout 200 python tools/bq.py --no-extras --preset perlin_spheres --width 1920 --height 1080 --samples 128 --steps 4 --warmup 1
timeout 200 python tools/bq.py --no-extras --preset smallpt --bvh --steps 4 --warmup 1
timeout 120 python tools/tree_stats.py 2>&1 | tail -3
if [ -f pathtrace-rs_amd/_build_dev/libptgpu.so ]; then
  export PTGPU_BUILD_DIR=_build
  export PTGPU_BUILD_DIR=_build_dev
  echo "dev build (nodes = wave rounds, sphere tests = calls, both per RAY: rounds per call = ratio):"
  timeout 120 python tools/tree_stats.py perlin_spheres 960 540 8 1 2>&1 | tail -2
  export PTGPU_BUILD_DIR=_build
fi
