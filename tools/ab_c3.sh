#!/bin/bash
# Development aid: config 3 (the metric) on each library variant in _ab/ (A/B of list-kernel changes), two rounds.
B="python bench.py $AB_ARGS --steps 12 --warmup 3 --no-cpu-baseline --no-extras"
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],3), round(d["roofline"]["kernel_ms"],3))'
cp pathtrace-rs_amd/_build/libptgpu.so /tmp/cur.so
for rep in 1 2 3; do
  for f in cur $(ls _ab | sed 's/libptgpu_//; s/.so//'); do
    if [ $f = cur ]; then cp /tmp/cur.so pathtrace-rs_amd/_build/libptgpu.so; else cp _ab/libptgpu_$f.so pathtrace-rs_amd/_build/libptgpu.so; fi
    $B 2>/dev/null | python -c "$P" $f
  done
done
cp /tmp/cur.so pathtrace-rs_amd/_build/libptgpu.so
