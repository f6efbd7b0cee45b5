#!/bin/bash
# Development aid: measuring launch over every tile (tuning bit 262144) vs one colour of a checkerboard (default), on the library in _build_dev.
export PTGPU_BUILD_DIR=_build
export PTGPU_BUILD_DIR=_build_dev
out=gpurun_out/checker_ab.log
: > $out
for rep in 1 2 3; do
for args in "--steps 20 --warmup 3" "--bvh --steps 10 --warmup 2" "--preset random --steps 10 --warmup 2" "--preset aras --width 1280 --height 720 --samples 16 --steps 20 --warmup 3" "--samples 16 --steps 20 --warmup 3" "--samples 256 --steps 5 --warmup 1"; do
  for v in 262144 0; do
    echo "variant $v: $(PTGPU_VARIANT=$v timeout 300 python tools/bq.py $args --no-extras)" >> $out
  done
done; done
timeout 900 python -m pytest tests -m gpu -x -q -k "full_frames or exact_parity or handover or work_order or progressive or shard" 2>&1 | tail -4 >> $out
export PTGPU_BUILD_DIR=_build
cat $out
