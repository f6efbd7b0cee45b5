run() { python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-pipeline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('   %.1f Mrays/s  %.3f ms/step  kernel %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for rep in 1 2 3; do for b in _build_prev _build; do export PTGPU_BUILD_DIR=$b; echo "$b c5"; run --preset perlin_spheres --bvh --width 1920 --height 1080 --samples 128; done; done
unset PTGPU_BUILD_DIR
timeout 600 python -m pytest tests -m gpu -x -q -k "cell_grid or golden or config5 or noise_parity or away_from" 2>&1 | tail -2
