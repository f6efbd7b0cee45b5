#!/bin/bash
# Development aid (ON THE GPU BOX): run a slice of the GPU tests on the library in _build_dev.   usage: tools/dev_tests.sh "<pytest -k expression>"
export PTGPU_BUILD_DIR=_build
export PTGPU_BUILD_DIR=_build_dev
timeout 1500 python -m pytest tests -m gpu -x -q -k "$1" 2>&1 | tail -25
export PTGPU_BUILD_DIR=_build
