#!/bin/bash
# Development aid: config 3 (and presets given as arguments) over PTGPU_CULL_STRIPS -- strips of the sort axis that are sorted along
# the second axis (1 = slabs) -- on -DPT_DEVKNOBS builds kept as _ab/libptgpu_dk.so and _ab/libptgpu_cs.so (+ -DPT_CULLSTATS).
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],3), round(d["roofline"]["kernel_ms"],3))'
cp pathtrace-rs_amd/_build/libptgpu.so /tmp/cur.so
for st in 1 2 3 4 5 8; do
  cp _ab/libptgpu_cs.so pathtrace-rs_amd/_build/libptgpu.so
  PTGPU_CULL_STRIPS=$st python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras "$@" 2>&1 | grep "ptgpu cull" | tail -1
  cp _ab/libptgpu_dk.so pathtrace-rs_amd/_build/libptgpu.so
  for rep in 1 2; do PTGPU_CULL_STRIPS=$st python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "$P" strips_$st; done
done
cp /tmp/cur.so pathtrace-rs_amd/_build/libptgpu.so
