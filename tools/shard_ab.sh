#!/bin/bash
# Development aid (ON THE GPU BOX): the N-GPU projection (tools/shard_times.py) and the headline on the shipped library and on the variants
# in pathtrace-rs_amd/_build_*/ named on the command line.
export PTGPU_BUILD_DIR=_build
for d in cur "$@"; do
  if [ $d = cur ]; then export PTGPU_BUILD_DIR=_build; else export PTGPU_BUILD_DIR=_build_$d; fi
  echo "== $d"
  python tools/shard_times.py --counts ${COUNTS:-1,2,4,8} --reps 2 2>&1 | grep "^N="
  for i in 1 2; do python tools/bq.py --no-extras --steps 8 --warmup 2 | cut -c40-110; done
  python tools/bq.py --no-extras --steps 4 --warmup 1 --width 8 --height 8 --samples 4096 | cut -c1-120
done
export PTGPU_BUILD_DIR=_build
