#!/bin/bash
# Development aid: shipped library vs one built with other compiler flags (directory given), over the kernel families.
d=${1:-_build_t}
for args in "--steps 20 --warmup 3" "--bvh --steps 10 --warmup 2" "--preset random --steps 10 --warmup 2" "--preset aras --width 1280 --height 720 --samples 16 --steps 20 --warmup 3" "--samples 256 --steps 4 --warmup 1" \
            "--preset perlin_spheres --bvh --width 1920 --height 1080 --samples 128 --steps 3 --warmup 1" "--preset cornell_smoke --steps 6 --warmup 2" "--preset cornell --steps 6 --warmup 2" "--preset simple_light --steps 6 --warmup 2" \
            "--preset smallpt --steps 6 --warmup 2" "--preset two_perlin_spheres --steps 6 --warmup 2"; do
  bash tools/ab.sh "$args" $d | cut -c1-140
done
