#!/bin/bash
# Development aid: the general-world kernel of the shipped library vs the one in _build_dev, alternating (ON THE GPU BOX).
out=gpurun_out/world_ab.log
: > $out
export PTGPU_BUILD_DIR=_build
for rep in 1 2; do
for args in "--preset simple_light" "--preset simple_light --bvh" "--preset cornell_smoke" "--preset cornell_smoke --bvh" "--preset cornell" "--preset cornell --bvh" ${EXTRA}; do
  for d in cur dev; do
    if [ $d = cur ]; then export PTGPU_BUILD_DIR=_build; else export PTGPU_BUILD_DIR=_build_dev; fi
    echo "$d: $(timeout 300 python tools/bq.py $args --steps 6 --warmup 2 --no-extras)" >> $out
  done
done; done
export PTGPU_BUILD_DIR=_build_dev
if [ -n "$TESTS" ]; then timeout 1200 python -m pytest tests -m gpu -x -q -k "$TESTS" 2>&1 | tail -5 >> $out; fi
export PTGPU_BUILD_DIR=_build
cat $out
