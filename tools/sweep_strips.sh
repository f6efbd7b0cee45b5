#!/bin/bash
# Development aid: config 3 (and presets given as arguments) over PTGPU_CULL_STRIPS -- strips of the sort axis that are sorted along
# the second axis (1 = slabs) -- on -DPT_DEVKNOBS builds in pathtrace-rs_amd/_build_dev (PTGPU_BUILD_DIR) and _ab/libptgpu_cs.so (+ -DPT_CULLSTATS).
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],3), round(d["roofline"]["kernel_ms"],3))'
export PTGPU_BUILD_DIR=_build
for st in 1 2 3 4 5 8; do
  export PTGPU_BUILD_DIR=_build_cs   # (make -C pathtrace-rs_amd B=_build_cs DEFS="-DPT_DEVKNOBS -DPT_CULLSTATS")
  PTGPU_CULL_STRIPS=$st python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras "$@" 2>&1 | grep "ptgpu cull" | tail -1
  export PTGPU_BUILD_DIR=_build_dev   # (make -C pathtrace-rs_amd B=_build_dev DEFS=-DPT_DEVKNOBS)
  for rep in 1 2; do PTGPU_CULL_STRIPS=$st python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "$P" strips_$st; done
done
export PTGPU_BUILD_DIR=_build
