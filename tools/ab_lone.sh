#!/bin/bash
# Development aid: latency-bound workloads on each library variant in _ab/ -- ONE wave alone on the GPU (8x8 pixels, 4096 spp) and
# a frame with fewer pixels than the GPU has lanes (1200x100 at 256 spp: an eighth of config 4, like a rank's shard at N = 8).
X='s/.* ([0-9.]+) ms\/step.*/\1/'
cp pathtrace-rs_amd/_build/libptgpu.so /tmp/cur.so
for rep in 1 2; do
for f in cur $(ls _ab | sed 's/libptgpu_//; s/.so//'); do
  if [ $f = cur ]; then cp /tmp/cur.so pathtrace-rs_amd/_build/libptgpu.so; else cp _ab/libptgpu_$f.so pathtrace-rs_amd/_build/libptgpu.so; fi
  echo "$f lone $(python tools/bq.py --width 8 --height 8 --samples 4096 --steps 2 --warmup 1 --no-extras | sed -E "$X") ms; eighth $(python tools/bq.py --width 1200 --height 100 --samples 256 --steps 4 --warmup 1 --no-extras | sed -E "$X") ms"
done; done
cp /tmp/cur.so pathtrace-rs_amd/_build/libptgpu.so
