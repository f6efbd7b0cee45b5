"""Ad-hoc GPU-vs-oracle check of the general-world presets (development aid; the tests are in tests/)."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import oracle_binding as ob  # noqa: E402

pthost = importlib.import_module("pathtrace-rs_amd.pthost")
ptgpu = pthost.ptgpu

W, H, SPP = 240, 160, 16
L = ob.lib(ob.build_native(os.path.join(ROOT, "gpurun_out", "ora_native")))
for name in sys.argv[1:] or ["smallpt", "simple_light", "cornell", "cornell_smoke", "random"]:
    for bvh in (False, True):
        sc = ob.OracleScene(name, W, H, use_bvh=bvh, library=L)
        want, rays_want = sc.update(SPP)
        hs = pthost.HostScene(name, W, H, samples=SPP, use_bvh=bvh, device=0)
        dev = hs.device_scene()
        params = ptgpu.PtParams(W, H, SPP, 10, 0, 1 if bvh else 0)
        got = np.zeros((H, W, 3), np.float32)
        if os.environ.get("WC_VARIANT"):
            dev.set_tuning(0, int(os.environ["WC_VARIANT"]))
        t = time.time()
        rays = dev.update(params, hs.camera, 0, got)
        dt = time.time() - t
        bad = np.argwhere((got != want).any(axis=2))
        print("%-14s %-4s rays gpu %d oracle %d  %s  differing pixels %d  maxabs %.3g  kernel %.2f ms" % (
            name, "bvh" if bvh else "list", rays, rays_want, "RAYS-OK" if rays == rays_want else "RAYS-DIFF",
            len(bad), float(np.abs(got - want).max()), dev.last_kernel_ms()))
        if len(bad):
            y, x = bad[0]
            print("   first diff at", (x, y), got[y, x], want[y, x])
