#!/bin/bash
# Development aid: hand-over policy sweep on a -DPT_DEVKNOBS library (make B=_build_dev DEFS=-DPT_DEVKNOBS), copied over the shipped one ON THE BOX
cp pathtrace-rs_amd/_build_dev/libptgpu.so pathtrace-rs_amd/_build/libptgpu.so
out=gpurun_out/coop_sweep.log
: > $out
for args in "--samples 16 --steps 20 --warmup 3" "--samples 32 --steps 20 --warmup 3" "--steps 20 --warmup 3" "--width 1200 --height 100 --samples 256 --steps 4 --warmup 1"; do
  echo "off: $(PTGPU_VARIANT=65536 timeout 300 python tools/bq.py $args --no-extras)" >> $out
  for cfg in ${COOP_CFGS:-"4:2" "4:1000" "0:1000" "8:1000" "4:4" "1:2"}; do
    echo "live<=${cfg%%:*} streak>=${cfg##*:}: $(PTGPU_COOP_LIVE=${cfg%%:*} PTGPU_COOP_STREAK=${cfg##*:} timeout 300 python tools/bq.py $args --no-extras)" >> $out
  done
done
cat $out
