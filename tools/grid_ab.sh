export PTGPU_BUILD_DIR=_build
export PTGPU_BUILD_DIR=_build_dev
timeout 800 python tools/grid_ab.py 2>&1 | grep -v "ptgpu grid\]" | tail -40
export PTGPU_BUILD_DIR=_build
