#!/usr/bin/env python3
"""Development aid: the work order never changes a pixel -- seeded sphere worlds at random frame sizes and sample counts (>= 12: the
two-launch path), rendered with the checkerboard measuring launch (tuning 8192), with every tile measured (8192 | 262144), in natural order
(32) and with the hand-over off (8192 | 65536); all four frames and ray counts must be equal. Usage: order_soak.py [first_seed] [count]"""
import importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import oracle_binding as ob
from conftest import load_ptgpu
spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
tgp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tgp)
ptgpu = load_ptgpu()
first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
n_ordered = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    W, H, S = int(rng.integers(200, 900)), int(rng.integers(150, 700)), int(rng.choice([12, 13, 16, 24, 40]))
    n = int(rng.choice([40, 150, 400, 700]))
    bvh = bool(seed % 3 == 0)
    w = tgp._random_sphere_world(ob, seed, n, W, H, float(rng.uniform(3, 12)), float(rng.uniform(0.2, 1.0)))
    osc = ob.OracleScene.from_world(w["hitables"], w["transforms"], w["materials"], w["textures"], w["camera"], W, H, sky=w["sky"], use_bvh=bvh)
    ex = osc.export()
    frames = {}
    for variant in (8192, 8192 | 262144, 32, 8192 | 65536):
        sc = ptgpu.Scene(ob.to_ptgpu_world_desc(ptgpu, ex), 0)
        sc.set_tuning(0, variant)
        out = np.zeros((H, W, 3), np.float32)
        rays = [sc.update(ptgpu.PtParams(W, H, S, 10, 0, 1 if bvh else 0), ptgpu.PtCamera.from_floats(ex["camera"]), f, out) for f in range(2)]
        ch = sc.last_kernel_choice()
        name = ch["name"]
        n_ordered += int(bool(ch["ordered"]))
        sc.close()
        frames[variant] = (out, rays)
    ref = frames[32]
    for v, (out, rays) in frames.items():
        if rays != ref[1] or not np.array_equal(out, ref[0], equal_nan=True):
            bad += 1
            print("MISMATCH seed %d %dx%dx%d n %d bvh %s variant %d (%s): rays %s vs %s" % (seed, W, H, S, n, bvh, v, name, rays, ref[1]))
    osc.close()
print("order soak: %d worlds x 4 schedules (%d renders took the ordered path), %d mismatches" % (count, n_ordered, bad))
