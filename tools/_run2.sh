run() { python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extras --no-pipeline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('   %.1f Mrays/s  %.3f ms/step  kernel %.3f ms rays %d' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['rays_per_step']))"; }
export PTGPU_BUILD_DIR=_build_dk
for v in 0 4; do export PTGPU_VARIANT=$v; echo "variant $v aras"; run --preset aras --width 1280 --height 720 --samples 16; run --preset aras --width 1280 --height 720 --samples 16; done
