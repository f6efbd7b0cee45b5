"""Development aid: far-origin sphere worlds (tests/test_gpu_parity.py _far_origin_world) through every list-kernel path
against the oracle, ONE SUBPROCESS PER WORLD so that a GPU fault in one does not hide the others.
Usage: python tools/far_worlds.py named | fuzz FIRST COUNT | one KIND SCALE SEED N SPREAD RMAX DEPTH W H S"""
import importlib.util
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(kind, scale, seed, n, spread, rmax, depth, W, H, S, moving=0):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import oracle_binding as ob
    from conftest import load_ptgpu
    spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
    tgp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tgp)
    ptgpu = load_ptgpu()
    w = tgp._far_origin_world(ob, seed, n, W, H, spread, rmax, kind, scale, moving=bool(moving))
    for bvh in (False, True):
        osc = ob.OracleScene.from_world(w["hitables"], w["transforms"], w["materials"], w["textures"], w["camera"], W, H, sky=w["sky"], use_bvh=bvh)
        ex = osc.export()
        ref, ref_rays = osc.update(S, max_depth=depth, frame_num=0)
        for variant in (4 | 64, 1024, 0) + ((256,) if bvh else (8,)):
            sc = ptgpu.Scene(ob.to_ptgpu_world_desc(ptgpu, ex), 0)
            sc.set_tuning(0, variant)
            if variant == 8:
                sc.debug_counters(reset=True)
            out = np.zeros((H, W, 3), np.float32)
            print("  bvh %d variant %d ..." % (bvh, variant), end="", flush=True)
            rays = sc.update(ptgpu.PtParams(W, H, S, depth, 0, 1 if bvh else 0), ptgpu.PtCamera.from_floats(ex["camera"]), 0, out)
            ok = rays == ref_rays and np.array_equal(ref, out, equal_nan=True)
            print(" %s rays %d vs %d, %s%s" % ("ok" if ok else "MISMATCH", rays, ref_rays, tgp._report(ref, out),
                                               (" " + repr(sc.debug_counters())) if variant == 8 else ""), flush=True)
            sc.close()


def spawn(args):
    print("world", *args, flush=True)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "one"] + [str(a) for a in args])
    if r.returncode != 0:
        print("  SUBPROCESS EXIT %d" % r.returncode, flush=True)


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "one":
        a = sys.argv[2:]
        one(a[0], float(a[1]), int(a[2]), int(a[3]), float(a[4]), float(a[5]), int(a[6]), int(a[7]), int(a[8]), int(a[9]), int(a[10]) if len(a) > 10 else 0)
    elif mode == "named":
        for kind, scale in [("enclosing", 2.0e2), ("enclosing", 2.0e3), ("enclosing", 2.0e4), ("enclosing", 2.0e5), ("enclosing", 3.0e6),
                            ("offcentre", 3.0e2), ("offcentre", 1.0e4), ("ground", 1.0e4), ("ground", 1.0e5), ("mirrors", 1.0e3), ("mirrors", 3.0e4)]:
            spawn([kind, scale, 31, 300, 6.0, 0.3, 25 if kind == "offcentre" else 10, 128, 96, 4])
        for scale in (3.0e2, 3.0e4, 3.0e6):
            spawn(["enclosing", scale, 41, 200, 4.0, 0.7, 10, 96, 64, 3, 1])
    else:
        first, count = int(sys.argv[2]), int(sys.argv[3])
        for seed in range(first, first + count):
            rng = np.random.default_rng(seed)
            n = int(rng.choice([40, 150, 400, 700, 900]))
            kind = ["enclosing", "enclosing", "offcentre", "ground", "mirrors"][int(rng.integers(0, 5))]
            scale = float(10.0 ** rng.uniform(2.5, 6.5))
            spread, rmax = float(rng.uniform(2, 15)), float(rng.uniform(0.1, 1.0))
            depth = int(rng.choice([2, 10, 10, 25]))
            spawn([kind, scale, seed, n, spread, rmax, depth, 64, 48, 2])
