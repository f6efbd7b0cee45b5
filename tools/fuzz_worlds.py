"""Ad-hoc fuzzing of the GPU kernels against the oracle with seeded random worlds (development aid; the seeds that
found bugs live on as cases in tests/test_gpu_parity.py). Usage: python tools/fuzz_worlds.py [first_seed] [count]"""
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
ptgpu = load_ptgpu()
L = ob.lib(ob.build_native(os.path.join(ROOT, "gpurun_out", "ora_native")))

first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 40
W, H, S = 96, 64, 3
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    kind = seed % 3
    if kind == 0:      # sphere worlds of assorted sizes and radius ranges, sometimes with extreme extras
        n = int(rng.choice([3, 20, 33, 64, 150, 400, 800]))
        extras = [(), ([0, -300, 0, 298],), ([0, 0, 0, 0.0], [2, 2, 2, -1e-3]), ([0, 0, 0, 25],)][int(rng.integers(0, 4))]
        w = tgp._random_sphere_world(ob, seed, n, W, H, float(rng.uniform(2, 15)), float(rng.uniform(0.1, 1.5)), extras)
    elif kind == 1:    # general worlds
        w = tgp._random_world(ob, seed, int(rng.integers(1, 40)), (0, 1, 2, 3, 4, 5), W, H, sky=(0.3, 0.3, 0.3) if seed & 1 else None)
        if seed % 4 == 1:
            w = tgp._with_image_textures(w, seed)    # Texture::Image on lambertians, the emitter and the media
    else:              # sphere + moving sphere worlds (fast MOVING kernels when >= 32 prefiltered)
        times = [((0.0, 1.0),), ((0.0, 1.0), (-1.0, 2.0)), ((0.25, 0.5),)][int(rng.integers(0, 3))]
        w = tgp._random_world(ob, seed, int(rng.choice([10, 60, 200])), (0, 1, 1), W, H, moving_times=times, media=False, instances=False)
    depth, frame = int(rng.choice([0, 1, 2, 5, 10, 10, 10, 25])), int(rng.choice([0, 0, 1, 7]))
    for bvh in (False, True):
        out, rays, ref, ref_rays = tgp._render_world_both(ptgpu, ob, w, W, H, S, bvh, depth=depth, frame=frame)
        ok = rays == ref_rays and np.array_equal(ref, out, equal_nan=True)
        if ok and bvh and kind != 1:     # BVH worlds also have a second device path: the internal-tree kernel
            out2, rays2, _, _ = tgp._render_world_both(ptgpu, ob, w, W, H, S, bvh, variant=256, depth=depth, frame=frame)
            ok = rays2 == ref_rays and np.array_equal(ref, out2, equal_nan=True)
        if not ok:
            bad += 1
            print("MISMATCH seed %d kind %d bvh %s: rays %d vs %d, %s" % (seed, kind, bvh, rays, ref_rays, tgp._report(ref, out)))
print("fuzz: %d worlds x list/BVH, %d mismatches" % (count, bad))
