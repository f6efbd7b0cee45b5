"""Ad-hoc fuzzing of the GPU kernels against the oracle with seeded random worlds (development aid; the seeds that
found bugs live on as cases in tests/test_gpu_parity.py). Usage: python tools/fuzz_worlds.py [first_seed] [count] [all|classic]
Kinds (seed % 5): 0 sphere worlds, 1 general worlds, 2 moving-sphere worlds, 3 far bounce origins, 4 scene graphs.
Mode "graphs": scene graphs with every nesting, flattened or interpreted (csrc/pt_graph.h) as the product decides.
Mode "noise": general worlds with Noise textures (on Lambertians, behind a Checker, in media, sometimes on the light) under a gradient /
black / coloured sky: ray counts against the oracle, colours within the sinf tolerance, and the general-world kernel's two ways of forming
a Noise colour -- when a lit path ends (default) / where the surface is hit (tuning bit 131072) -- against each other bit for bit."""
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
MODE = sys.argv[3] if len(sys.argv) > 3 else "all"    # "all": five kinds (seed % 5); "classic": the three kinds of rounds 1-2 (seed % 3)
W, H, S = 96, 64, 3
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    kind = seed % 5 if MODE == "all" else seed % 3
    if MODE == "graphs":   # scene graphs with every nesting (a medium around a List / a medium / a BVHNode, BVHNodes anywhere): flattened where possible, interpreted otherwise
        g = tgp._random_graph_world(ob, seed, W, H, n_top=int(rng.integers(1, 9)), max_depth=int(rng.integers(2, 7)), media=bool(seed % 3), wild=bool(seed % 4))
        depth = int(rng.choice([0, 1, 5, 10, 25]))
        out, rays, ref, ref_rays, choice = tgp._render_graph_both(ptgpu, ob, g, W, H, S, depth=depth, frame=int(rng.choice([0, 2])))
        n_interp = globals().get("n_interp", 0) + int(choice["world_graph"])
        globals()["n_interp"] = n_interp
        if rays != ref_rays or not np.array_equal(ref, out, equal_nan=True):
            bad += 1
            print("MISMATCH seed %d graph (%s): rays %d vs %d, %s" % (seed, choice["name"], rays, ref_rays, tgp._report(ref, out)))
        continue
    if MODE == "noise":
        sky = [None, (0.0, 0.0, 0.0), tuple(rng.uniform(0.0, 1.0, 3))][int(rng.integers(0, 3))]
        w = tgp._random_world(ob, seed, int(rng.integers(1, 30)), (0, 1, 2, 3, 4, 5), W, H, sky=sky)
        tex = w["textures"].copy()
        tex[1] = [2, 0, 0, 0, -1, -1, float(rng.uniform(0.2, 8.0))]
        tex[3] = [2, 0, 0, 0, -1, -1, float(rng.uniform(0.2, 8.0))]
        tex[5] = [1, 0, 0, 0, 0, 1, 0]
        if seed % 3 == 0:
            tex[6] = [2, 0, 0, 0, -1, -1, 2.0]
        w = dict(w, textures=tex)
        depth, frame = int(rng.choice([1, 2, 5, 10, 10, 25, 64, 65])), int(rng.choice([0, 0, 3]))
        for bvh in (False, True):
            out, rays, ref, ref_rays = tgp._render_world_both(ptgpu, ob, w, W, H, S, bvh, depth=depth, frame=frame)
            out2, rays2, _, _ = tgp._render_world_both(ptgpu, ob, w, W, H, S, bvh, variant=131072, depth=depth, frame=frame)
            if rays != ref_rays or rays2 != ref_rays or not np.array_equal(out, out2, equal_nan=True) or not np.allclose(out, ref, rtol=0, atol=2e-6, equal_nan=True):
                bad += 1
                print("MISMATCH seed %d noise bvh %s depth %d: rays %d / %d vs %d, lazy vs eager %s, vs oracle %s" % (seed, bvh, depth, rays, rays2, ref_rays, tgp._report(out2, out), tgp._report(ref, out)))
        continue
    if kind == 3:      # far bounce origins: concave mirrors / huge grounds / distant mirrors around a cloud (every list-kernel path)
        n = int(rng.choice([40, 150, 400, 700, 900]))
        fk = ["enclosing", "enclosing", "offcentre", "ground", "mirrors"][int(rng.integers(0, 5))]
        scale = float(10.0 ** rng.uniform(2.5, 6.5))
        w = tgp._far_origin_world(ob, seed, n, 64, 48, float(rng.uniform(2, 15)), float(rng.uniform(0.1, 1.0)), fk, scale, moving=bool(seed % 7 == 0 and n <= 400))
        for bvh in (False, True):
            msgs = tgp._check_all_list_paths_against_the_oracle(ptgpu, ob, w, 64, 48, 2, bvh, depth=int(rng.choice([2, 10, 10, 25])))
            if msgs:
                bad += 1
                print("MISMATCH seed %d far %s scale %.3g bvh %s: %s" % (seed, fk, scale, bvh, msgs))
        continue
    if kind == 4:      # scene graphs: nested Lists / Instances / media, flattened by the product, nested literally by the oracle
        g = tgp._random_graph_world(ob, seed, W, H, n_top=int(rng.integers(3, 9)), max_depth=int(rng.integers(2, 6)), media=bool(seed % 3))
        osc = ob.OracleScene.from_graph(g["hitables"], g["transforms"], g["materials"], g["textures"], g["camera"], W, H, g["nodes"], g["node_children"],
                                        g["root_node"], sky=g["sky"], library=L)
        depth = int(rng.choice([1, 5, 10, 25]))
        ref, ref_rays = osc.update(S, max_depth=depth, frame_num=0)
        materials = [(int(r[0]), r[1:4], r[4], int(r[5])) for r in g["materials"]]
        textures = [(int(r[0]), r[1:4], int(r[4]), int(r[5]), r[6]) for r in g["textures"]]
        sc = ptgpu.Scene(ptgpu.WorldDesc(g["hitables"], g["transforms"], materials, textures, sky=g["sky"], nodes=g["nodes"],
                                         node_children=g["node_children"], root_node=g["root_node"]), 0)
        out = np.zeros((H, W, 3), np.float32)
        rays = sc.update(ptgpu.PtParams(W, H, S, depth, 0, 0), ptgpu.PtCamera.from_floats(g["camera"]), 0, out)
        sc.close()
        if rays != ref_rays or not np.array_equal(ref, out, equal_nan=True):
            bad += 1
            print("MISMATCH seed %d graph: rays %d vs %d, %s" % (seed, rays, ref_rays, tgp._report(ref, out)))
        continue
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
print("fuzz: %d worlds x list/BVH, %d mismatches%s" % (count, bad, (" (%d graphs interpreted)" % globals().get("n_interp", 0)) if MODE == "graphs" else ""))
