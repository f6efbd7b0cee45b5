#!/bin/bash
# Development aid: A/B two builds of libptgpu.so ON THE GPU BOX. The alternative library is built into its own directory
# (make -C pathtrace-rs_amd B=_build_dev DEFS=-D...), which travels with the snapshot; this script points the loader at each candidate
# (PTGPU_BUILD_DIR: the shipped _build is never overwritten), alternating, and runs tools/bq.py with the given arguments.
#   usage: tools/ab.sh "<bq.py args>" [dir ...]      (default dirs: _build_dev; "cur" = the shipped _build is always first)
args="$1"; shift
dirs="${@:-_build_dev}"
export PTGPU_BUILD_DIR=_build
for rep in 1 2 3; do
  for d in cur $dirs; do
    if [ $d = cur ]; then export PTGPU_BUILD_DIR=_build; else export PTGPU_BUILD_DIR=$d; fi
    echo "$d: $(python tools/bq.py $args --no-extras)"
  done
done
export PTGPU_BUILD_DIR=_build
