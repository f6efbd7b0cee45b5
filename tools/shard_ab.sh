#!/bin/bash
# Development aid (ON THE GPU BOX): the N-GPU projection (tools/shard_times.py) and the headline on the shipped library and on the variants
# in pathtrace-rs_amd/_build_*/ named on the command line.
cp pathtrace-rs_amd/_build/libptgpu.so /tmp/cur.so
for d in cur "$@"; do
  if [ $d = cur ]; then cp /tmp/cur.so pathtrace-rs_amd/_build/libptgpu.so; else cp pathtrace-rs_amd/_build_$d/libptgpu.so pathtrace-rs_amd/_build/libptgpu.so; fi
  echo "== $d"
  python tools/shard_times.py --counts ${COUNTS:-1,2,4,8} --reps 2 2>&1 | grep "^N="
  for i in 1 2; do python tools/bq.py --no-extras --steps 8 --warmup 2 | cut -c40-110; done
  python tools/bq.py --no-extras --steps 4 --warmup 1 --width 8 --height 8 --samples 4096 | cut -c1-120
done
cp /tmp/cur.so pathtrace-rs_amd/_build/libptgpu.so
