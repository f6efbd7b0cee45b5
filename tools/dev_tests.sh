#!/bin/bash
# Development aid (ON THE GPU BOX): run a slice of the GPU tests on the library in _build_dev.   usage: tools/dev_tests.sh "<pytest -k expression>"
cp pathtrace-rs_amd/_build/libptgpu.so /tmp/cur.so
cp pathtrace-rs_amd/_build_dev/libptgpu.so pathtrace-rs_amd/_build/libptgpu.so
timeout 1500 python -m pytest tests -m gpu -x -q -k "$1" 2>&1 | tail -25
cp /tmp/cur.so pathtrace-rs_amd/_build/libptgpu.so
