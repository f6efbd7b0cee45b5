#!/bin/bash
# The DESIGN.md table in one go (through gpurun, from the repo root): every BASELINE config and the other presets.
Q="python tools/bq.py --no-extras --steps 8 --warmup 2"
$Q --preset small --width 200 --height 100 --samples 4
$Q --preset aras --width 1280 --height 720 --samples 16
$Q
$Q --bvh
$Q --samples 256
$Q --width 2400 --height 1600
$Q --preset perlin_spheres --bvh --width 1920 --height 1080 --samples 128 --steps 4 --warmup 1
$Q --preset perlin_spheres --width 1920 --height 1080 --samples 128 --steps 4 --warmup 1
for p in random cornell cornell_smoke simple_light smallpt; do $Q --preset $p; $Q --preset $p --bvh; done
