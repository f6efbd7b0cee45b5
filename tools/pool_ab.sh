#!/bin/bash
# Development aid (run ON the GPU box: gpurun -- 'bash tools/pool_ab.sh'): the per-wave pixel pools of the 1024-thread frame kernels
# (csrc/pt_kernel.h POOL) against the batched refill, and their knobs. Needs a -DPT_DEVKNOBS build in pathtrace-rs_amd/_build_dev
#   make -C pathtrace-rs_amd -j4 B=_build_dev DEFS=-DPT_DEVKNOBS
# (selected through PTGPU_BUILD_DIR: the shipped _build is never overwritten). Optional reference builds, e.g. round 5's: _build_r5.
set -e
run() { python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras --no-pipeline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('   %.1f Mrays/s  %.3f ms/step  kernel %.3f ms  lds %d block %d' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['lds_bytes'], d['config']['block']))"; }
export PTGPU_BUILD_DIR=_build_dev
for rep in 1 2; do
  echo "batched refill (tuning bit 1048576)"; PTGPU_VARIANT=1048576 run
  for n in 8 16 32; do echo "pool of $n"; PTGPU_POOL=$n run; done
  for t in 0 2 4 16 64; do echo "pool of 32, exact claims below a fair share of $t items per wave"; PTGPU_POOL_TAIL=$t run; done
  echo "aras 16 spp: batched / pool"; PTGPU_VARIANT=1048576 run --preset aras --width 1280 --height 720 --samples 16; run --preset aras --width 1280 --height 720 --samples 16
  echo "256 spp: batched / pool"; PTGPU_VARIANT=1048576 run --samples 256 --steps 4; run --samples 256 --steps 4
done
