#!/bin/bash
# Development aid: refill threshold / workgroups per CU of the general-world kernel on a -DPT_DEVKNOBS library (ON THE GPU BOX).
export PTGPU_BUILD_DIR=_build
export PTGPU_BUILD_DIR=_build_dev
out=gpurun_out/world_sweep.log
: > $out
for p in "cornell_smoke" "simple_light" "cornell_smoke --bvh" "simple_light --bvh"; do
  echo "default: $(timeout 300 python tools/bq.py --preset $p --steps 6 --warmup 2 --no-extras)" >> $out
  for r in 1 2 8 16 32; do echo "refill $r: $(PTGPU_REFILL=$r timeout 300 python tools/bq.py --preset $p --steps 6 --warmup 2 --no-extras)" >> $out; done
  for b in 2 3 4 5 6; do echo "bpc $b: $(PTGPU_BLOCKS_PER_CU=$b timeout 300 python tools/bq.py --preset $p --steps 6 --warmup 2 --no-extras)" >> $out; done
  echo "occ3: $(PTGPU_WORLD_OCC3=1 timeout 300 python tools/bq.py --preset $p --steps 6 --warmup 2 --no-extras)" >> $out
done
export PTGPU_BUILD_DIR=_build
cat $out
