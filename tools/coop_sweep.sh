#!/bin/bash
# Development aid: hand-over policy sweep on a -DPT_DEVKNOBS library (make B=_build_dev DEFS=-DPT_DEVKNOBS), copied over the shipped one ON THE BOX.
# COOP_CFGS: "live:streak:period_mask:min_est" ...
export PTGPU_BUILD_DIR=_build_dev
out=gpurun_out/coop_sweep.log
: > $out
for args in "--steps 20 --warmup 3" "--width 1200 --height 100 --samples 256 --steps 4 --warmup 1" "--width 1200 --height 200 --samples 256 --steps 4 --warmup 1"; do
  echo "off: $(PTGPU_VARIANT=65536 timeout 300 python tools/bq.py $args --no-extras)" >> $out
  for cfg in ${COOP_CFGS:-"4:2:3:24"}; do
    IFS=: read l s p e <<< "$cfg"
    echo "live<=$l streak>=$s period&$p est>=$e: $(PTGPU_COOP_LIVE=$l PTGPU_COOP_STREAK=$s PTGPU_COOP_PERIOD=$p PTGPU_COOP_EST=$e timeout 300 python tools/bq.py $args --no-extras)" >> $out
  done
done
cat $out
