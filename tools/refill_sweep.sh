#!/bin/bash
# Development aid: config 3 over PTGPU_REFILL (lanes that must be waiting before a wave refills) on the -DPT_DEVKNOBS library in _build_dev,
# after the shipped library at its default.
export PTGPU_BUILD_DIR=_build
for rep in 1 2; do
  export PTGPU_BUILD_DIR=_build
  echo "shipped: $(timeout 300 python tools/bq.py --steps 12 --warmup 3 --no-extras | cut -c40-120)"
  export PTGPU_BUILD_DIR=_build_dev
  for r in ${REFILLS:-2 4 6 8 12 16}; do echo "dev refill $r: $(PTGPU_REFILL=$r timeout 300 python tools/bq.py --steps 12 --warmup 3 --no-extras | cut -c40-120)"; done
done
export PTGPU_BUILD_DIR=_build_dev
timeout 900 python -m pytest tests -m gpu -x -q -k "full_frames or exact_parity or handover or work_order or shard or progressive" 2>&1 | grep -E "passed|failed"
export PTGPU_BUILD_DIR=_build
