run() { python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extras --no-pipeline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('   %.1f Mrays/s  %.3f ms/step  kernel %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for rep in 1 2 3; do for b in _build_prev _build; do export PTGPU_BUILD_DIR=$b
echo "$b c3"; run
echo "$b aras"; run --preset aras --width 1280 --height 720 --samples 16
echo "$b c4"; run --samples 256 --steps 4
done; done
