#!/usr/bin/env python3
"""Soak of the per-wave pixel pools (csrc/pt_kernel.h POOL), ON THE GPU BOX: python tools/pool_soak.py [first_seed] [count]

Seeded sphere clouds (tests/test_gpu_parity.py _random_scene: 40..700 spheres, lambertian / metal / dielectric, a huge ground) at frames of
more than two pixels per lane -- the size from which the 1024-thread frame kernels keep pools -- with ragged edges, 2..20 samples (one launch
and two), depths that leave 32 / 16 / 8 pool entries, progressive frames. Per world: the default kernel (must carry ",pool"), the batched
refill (tuning bit 1048576), pools without hand-over (65536) and the exact VALU scan (4): every float and the ray count equal. The scan and
the batched refill are what the rest of the suite pins to the oracle at small sizes; here they pin the pools at full size.
"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import oracle_binding as ob  # noqa: E402
from conftest import load_ptgpu  # noqa: E402

spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
tgp = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tgp)


def run(first, count, verbose=True):
    ptgpu = load_ptgpu()
    pooled = bad = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        n = int(rng.choice([40, 120, 300, 488, 700]))
        half, rmax = float(rng.uniform(3, 20)), float(rng.uniform(0.1, 0.9))
        desc, cam = tgp._random_scene(ptgpu, ob, seed, n, half, rmax, float(rng.uniform(2.0, 8.0)) * half)
        W, H = int(rng.choice([1024, 1100, 1203, 1280])), int(rng.choice([576, 640, 720, 797]))
        S, depth, frame = int(rng.choice([2, 3, 12, 20])), int(rng.choice([10, 10, 20, 22])), int(rng.choice([0, 0, 2]))
        sc = ptgpu.Scene(desc, 0)
        p = ptgpu.PtParams(W, H, S, depth, 0, 0)
        prev = rng.uniform(0, 1, (H, W, 3)).astype(np.float32) if frame else np.zeros((H, W, 3), np.float32)
        frames = {}
        for variant in (0, 1048576, 65536, 4):
            sc.set_tuning(0, variant)
            out = prev.copy()
            rays = sc.update(p, cam, frame, out)
            frames[variant] = (rays, out, sc.last_kernel_choice())
        sc.close()
        rays0, out0, ch0 = frames[0]
        pooled += int(ch0["pool_slots"] != 0)
        for variant in (1048576, 65536, 4):
            rays, out, ch = frames[variant]
            if rays != rays0 or not np.array_equal(out, out0, equal_nan=True):
                bad += 1
                print("MISMATCH seed %d %dx%dx%d depth %d frame %d: %s vs variant %d (%s): rays %d vs %d, %s" % (seed, W, H, S, depth, frame, ch0["name"], variant, ch["name"], rays0, rays, tgp._report(out0, out)))
        if verbose and (seed - first) % 10 == 9:
            print("  ... seed %d: %d worlds, %d on pooled kernels, %d mismatches" % (seed, seed - first + 1, pooled, bad), flush=True)
    return pooled, bad


if __name__ == "__main__":
    first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 40
    pooled, bad = run(first, count)
    print("pool soak: seeds %d..%d, %d worlds on pooled kernels, %d mismatches" % (first, first + count - 1, pooled, bad))
    sys.exit(1 if bad else 0)
