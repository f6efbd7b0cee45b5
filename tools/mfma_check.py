#!/usr/bin/env python3
"""Ad-hoc GPU check of the MFMA prefilter: verify-mode miss counter, equality with the exact scan."""
import os, sys, importlib.util
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel)); m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m; spec.loader.exec_module(m); return m
ptgpu = _load("pathtrace_rs_amd_ptgpu", "pathtrace-rs_amd/ptgpu.py")
pthost = _load("pathtrace_rs_amd_pthost", "pathtrace-rs_amd/pthost.py")
for preset, W, H, S in (("random_spheres", 120, 80, 4), ("random_spheres", 600, 400, 8), ("aras", 320, 180, 8), ("small", 200, 100, 4)):
    hs = pthost.HostScene(preset, W, H, samples=S, device=0)
    sc = hs.device_scene()
    p = ptgpu.PtParams(W, H, S, 10, 0, 0)
    outs = {}
    for variant in (4, 0, 8):
        sc.set_tuning(0, variant)
        buf = np.zeros((H, W, 3), np.float32)
        rays = sc.update(p, hs.camera, 0, buf)
        outs[variant] = (buf, rays)
        grid, block, lds = sc.last_launch_info()
        print(preset, "variant", variant, "rays", rays, "kernel ms %.3f" % sc.last_kernel_ms(), "lds", lds, end=" ")
        if variant == 8:
            print(sc.debug_counters(), end="")
        print()
    same = all(np.array_equal(outs[4][0], outs[v][0]) and outs[4][1] == outs[v][1] for v in (0, 8))
    print(preset, "MFMA == exact scan:", same)
    for v in (0, 8):
        d = (outs[4][0] != outs[v][0]).any(axis=2)
        if d.any():
            ys, xs = np.nonzero(d)
            print("  variant", v, "differs at", list(zip(xs[:5].tolist(), ys[:5].tolist())), "count", d.sum(), "rays", outs[v][1], "vs", outs[4][1])
