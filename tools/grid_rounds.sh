export PTGPU_BUILD_DIR=_build
export PTGPU_BUILD_DIR=_build_dev
timeout 120 python tools/tree_stats.py perlin_spheres 960 540 8 1 2>&1 | grep -v binary | tail -8
export PTGPU_BUILD_DIR=_build
