run() { python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-extras --no-pipeline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('   %.1f Mrays/s  %.3f ms/step  kernel %.3f ms pass %.3f  measuring+order %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['pass_ms'], d['roofline']['pass_ms']-d['roofline']['kernel_ms']))"; }
for rep in 1 2 3; do for b in _build_prev _build; do export PTGPU_BUILD_DIR=$b
echo "$b c3"; run
echo "$b aras"; run --preset aras --width 1280 --height 720 --samples 16
done; done
unset PTGPU_BUILD_DIR
timeout 900 python -m pytest tests -m gpu -x -q -k "full_frames or exact_parity or progressive or pixel_pool or golden" 2>&1 | tail -2
