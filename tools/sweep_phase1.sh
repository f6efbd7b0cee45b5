#!/bin/bash
# Development aid: config 3 over PTGPU_PHASE1_REFILL (lanes that must be waiting before a wave of the MEASURING launch refills) on a
# -DPT_DEVKNOBS build in pathtrace-rs_amd/_build_dev (PTGPU_BUILD_DIR); last column: pass_ms - kernel_ms = measuring launch + order kernel.
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],3), round(d["roofline"]["kernel_ms"],3), round(d["roofline"]["pass_ms"]-d["roofline"]["kernel_ms"],3))'
export PTGPU_BUILD_DIR=_build
export PTGPU_BUILD_DIR=_build_dev   # (make -C pathtrace-rs_amd B=_build_dev DEFS=-DPT_DEVKNOBS)
for r in 16 32 40 48 56 60 64; do for rep in 1 2; do PTGPU_PHASE1_REFILL=$r python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "$P" p1_refill_$r; done; done
export PTGPU_BUILD_DIR=_build
