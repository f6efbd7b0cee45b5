#!/bin/bash
# Development aid: the general-world presets on each library variant in _ab/ (A/B of pt_world.h changes).
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], sys.argv[2], round(d["value"],1), round(d["ms_per_step"],3))'
cp pathtrace-rs_amd/_build/libptgpu.so /tmp/cur.so
for rep in 1 2; do
  for f in cur $(ls _ab | sed 's/libptgpu_//; s/.so//'); do
    if [ $f = cur ]; then cp /tmp/cur.so pathtrace-rs_amd/_build/libptgpu.so; else cp _ab/libptgpu_$f.so pathtrace-rs_amd/_build/libptgpu.so; fi
    for p in "cornell_smoke" "cornell" "simple_light" "cornell_smoke --bvh"; do
      python bench.py --preset $p --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "$P" $f "$p"
    done
  done
done
cp /tmp/cur.so pathtrace-rs_amd/_build/libptgpu.so
