#!/bin/bash
# Development aid: A/B two builds of libptgpu.so ON THE GPU BOX. The alternative library is built into its own directory
# (make -C pathtrace-rs_amd B=_build_dev DEFS=-D...), which travels with the snapshot; this script copies each candidate over the
# shipped one in the box's scratch copy of the repo, alternating, and runs tools/bq.py with the given arguments.
#   usage: tools/ab.sh "<bq.py args>" [dir ...]      (default dirs: _build_dev; "cur" = the shipped _build is always first)
args="$1"; shift
dirs="${@:-_build_dev}"
cp pathtrace-rs_amd/_build/libptgpu.so /tmp/cur.so
for rep in 1 2 3; do
  for d in cur $dirs; do
    if [ $d = cur ]; then cp /tmp/cur.so pathtrace-rs_amd/_build/libptgpu.so; else cp pathtrace-rs_amd/$d/libptgpu.so pathtrace-rs_amd/_build/libptgpu.so; fi
    echo "$d: $(python tools/bq.py $args --no-extras)"
  done
done
cp /tmp/cur.so pathtrace-rs_amd/_build/libptgpu.so
