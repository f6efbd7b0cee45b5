#!/bin/bash
# Development aid (ON THE GPU BOX): every workload that runs on the 4-wide tree kernels, on the shipped library and on the variants in
# pathtrace-rs_amd/_build_*/ named on the command line.
export PTGPU_BUILD_DIR=_build
run() { python tools/bq.py --no-extras --steps 3 --warmup 1 "$@" | awk '{for(i=1;i<=NF;i++) if($i=="Mrays/s") printf "%9.1f Mrays/s %7.2f ms", $(i-1), $(i+1)}'; }
for d in cur "$@"; do
  if [ $d = cur ]; then export PTGPU_BUILD_DIR=_build; else export PTGPU_BUILD_DIR=_build_$d; fi
  echo "== $d"
  echo "c5 -B        $(run --preset perlin_spheres --bvh --width 1920 --height 1080 --samples 128)"
  echo "c5 list      $(run --preset perlin_spheres --width 1920 --height 1080 --samples 128)"
  echo "smallpt -B   $(run --preset smallpt --bvh)"
  echo "small -B     $(run --preset small --bvh)"
  echo "two_perlin -B $(run --preset two_perlin_spheres --bvh)"
  echo "c3 tree -B   $(PTGPU_VARIANT=256 run --bvh)"
done
export PTGPU_BUILD_DIR=_build
