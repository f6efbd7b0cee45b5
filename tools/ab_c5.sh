#!/bin/bash
# Development aid: config 5 (perlin_spheres BVH 1920x1080x128) on each library variant in _ab/ (A/B of tree-kernel changes).
B="python bench.py --preset perlin_spheres --bvh --width 1920 --height 1080 --samples 128 --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],2), round(d["roofline"]["kernel_ms"],2))'
cp pathtrace-rs_amd/_build/libptgpu.so /tmp/cur.so
for rep in 1 2; do
  for f in cur $(ls _ab | sed 's/libptgpu_//; s/.so//'); do
    if [ $f = cur ]; then cp /tmp/cur.so pathtrace-rs_amd/_build/libptgpu.so; else cp _ab/libptgpu_$f.so pathtrace-rs_amd/_build/libptgpu.so; fi
    $B 2>/dev/null | python -c "$P" $f
  done
done
cp /tmp/cur.so pathtrace-rs_amd/_build/libptgpu.so
