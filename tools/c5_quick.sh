#!/bin/bash
# Development aid (ON THE GPU BOX): config 5 in one go -- a parity slice of the tree kernels, the bench line, visits per ray.
timeout 600 python -m pytest tests -m gpu -x -q -k "golden_fixture or noise_parity or tree or bvh_world or far_ray" 2>&1 | tail -4
for i in 1 2; do timeout 200 python tools/bq.py --no-extras --preset perlin_spheres --bvh --width 1920 --height 1080 --samples 128 --steps 4 --warmup 1; done
timeout 200 python tools/bq.py --no-extras --preset perlin_spheres --width 1920 --height 1080 --samples 128 --steps 4 --warmup 1
timeout 200 python tools/bq.py --no-extras --preset smallpt --bvh --steps 4 --warmup 1
timeout 200 python tools/bq.py --no-extras --preset random_spheres --bvh --steps 4 --warmup 1
timeout 120 python tools/tree_stats.py 2>&1 | tail -3
