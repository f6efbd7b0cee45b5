cp pathtrace-rs_amd/_build/libptgpu.so /tmp/cur.so
cp pathtrace-rs_amd/_build_dev/libptgpu.so pathtrace-rs_amd/_build/libptgpu.so
timeout 120 python tools/tree_stats.py perlin_spheres 960 540 8 1 2>&1 | grep -v binary | tail -8
cp /tmp/cur.so pathtrace-rs_amd/_build/libptgpu.so
