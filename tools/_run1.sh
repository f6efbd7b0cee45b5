export PTGPU_BUILD_DIR=_build_dev
for env in "PTGPU_PARK_MAX=0" "PTGPU_PARK_MAX=4" "PTGPU_PARK_MAX=4 PTGPU_PARK_AFTER=100"; do
echo "== $env"
env $env timeout 60 python - <<PY 2>&1 | tail -4
import sys, importlib.util, numpy as np
sys.path.insert(0,'tests')
from conftest import load_ptgpu
ptgpu = load_ptgpu()
spec = importlib.util.spec_from_file_location("pthost", "pathtrace-rs_amd/pthost.py"); pthost = importlib.util.module_from_spec(spec); spec.loader.exec_module(pthost)
for (W,H,S) in ((96,54,2),(200,120,2),(320,200,16)):
    hs = pthost.HostScene("perlin_spheres", W, H, samples=S, use_bvh=True, device=0)
    sc = hs.device_scene()
    out = np.zeros((H,W,3), np.float32)
    rays = sc.update(ptgpu.PtParams(W,H,S,10,0,1), hs.camera, 0, out)
    print(W,H,S,rays, sc.last_kernel_choice()["name"], sc.last_kernel_choice()["lds_bytes"], flush=True)
PY
done
