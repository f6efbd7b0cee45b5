run() { python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-extras --no-pipeline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('   %.1f Mrays/s  %.3f ms/step  kernel %.3f ms rays %d' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['rays_per_step']))"; }
export PTGPU_BUILD_DIR=_build_dev
for rep in 1 2 3; do for r in 0 7; do echo "resv $r"; PTGPU_POOL_RESV=$r run; done; done
for r in 0 7; do echo "256spp resv $r"; PTGPU_POOL_RESV=$r run --samples 256 --steps 4;  echo "aras resv $r"; PTGPU_POOL_RESV=$r run --preset aras --width 1280 --height 720 --samples 16; done
