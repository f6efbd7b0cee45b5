#!/bin/bash
# Development aid: cooperative hand-over on (default) / off (tuning bit 65536), alternating, on the shipped library.
out=gpurun_out/coop_ab.log
: > $out
for rep in 1 2; do
for args in "--steps 20 --warmup 3" "--bvh --steps 10 --warmup 2" "--preset random --steps 10 --warmup 2" "--preset aras --width 1280 --height 720 --samples 16 --steps 20 --warmup 3" \
            "--samples 16 --steps 20 --warmup 3" "--samples 256 --steps 5 --warmup 1" "--width 2400 --height 1600 --steps 5 --warmup 1"; do
  for v in 0 65536; do
    echo "variant $v: $(PTGPU_VARIANT=$v timeout 300 python tools/bq.py $args --no-extras)" >> $out
  done
done; done
cat $out
