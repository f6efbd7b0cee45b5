cp pathtrace-rs_amd/_build/libptgpu.so /tmp/cur.so
cp pathtrace-rs_amd/_build_dev/libptgpu.so pathtrace-rs_amd/_build/libptgpu.so
timeout 800 python tools/grid_ab.py 2>&1 | grep -v "ptgpu grid\]" | tail -40
cp /tmp/cur.so pathtrace-rs_amd/_build/libptgpu.so
