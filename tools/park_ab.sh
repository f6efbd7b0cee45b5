#!/bin/bash
# Development aid (ON THE GPU BOX): parked walks of the cell-grid kernels (csrc/pt_grid.h) on BASELINE config 5 -- off, and over the two knobs.
# Needs a -DPT_DEVKNOBS build in pathtrace-rs_amd/_build_dev (selected through PTGPU_BUILD_DIR).
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-pipeline --preset perlin_spheres --bvh --width 1920 --height 1080 --samples 128 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('   %.1f Mrays/s  %.3f ms/step  kernel %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"; }
export PTGPU_BUILD_DIR=_build_dev
for rep in 1 2 3; do
  echo "no parking"; PTGPU_PARK_MAX=0 run
  for m in 6 8 12 16; do for a in 1 2; do echo "park <= $m lanes after $a rounds"; PTGPU_PARK_MAX=$m PTGPU_PARK_AFTER=$a run; done; done
done
