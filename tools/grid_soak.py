"""Soak of the uniform cell grid (csrc/pt_grid.h) on the GPU box: seeded dense sphere fields -- jittered 2D lattices, layers, 3D packings, from seed
100 000 on also RANDOM fields (filled cubes and thick layers: cells with no sphere, cells with a dozen), radii within a band, a ground and a few big spheres beside them, cameras inside, near, far and very far -- rendered by the default kernel (which must be
the grid walk for most of them) and by the exact VALU scan (tuning 4 | 64: the reference's semantics; BVH worlds: by the binary-tree kernel without a grid), which must agree bit for bit in every pixel
and in the ray count. Usage: python tools/grid_soak.py [first_seed] [count]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
pthost = importlib.import_module("pathtrace-rs_amd.pthost")
ptgpu = pthost.ptgpu
from grid_ab import camera  # noqa: E402  (camera.rs:21-53 in numpy)


def field(seed):
    rng = np.random.default_rng(seed)
    kind = int(rng.integers(0, 4))
    pitch = float(rng.uniform(0.3, 1.5))
    r0 = pitch * float(rng.uniform(0.32, 0.48))
    cloudy = seed >= 100000 and np.random.default_rng([seed, 1]).random() < 0.4   # (seeds below 100 000 keep the fields they always had)
    if cloudy:         # round 6: RANDOM fields dense enough for the planner (83 % + of the cells occupied): a filled cube, or a layer a few spheres thick
        n_c = int(rng.integers(1500, 7000))
        if rng.random() < 0.6:
            c = rng.uniform(-1, 1, (n_c, 3)) * r0 * n_c ** (1.0 / 3.0) * float(rng.uniform(0.85, 1.9))
        else:
            c = rng.uniform(-1, 1, (n_c, 3)) * r0 * n_c ** 0.5 * float(rng.uniform(0.65, 1.0))
            c[:, 1] = rng.uniform(0.0, float(rng.uniform(1.0, 4.0)) * r0, n_c)
    elif kind == 0:      # one layer, like BASELINE config 5
        side = int(rng.integers(33, 60))
        ij = np.stack(np.meshgrid(np.arange(side), np.arange(side)), -1).reshape(-1, 2)
        c = np.stack([pitch * ij[:, 0], np.full(len(ij), r0), pitch * ij[:, 1]], 1)
    elif kind == 1:    # two or three layers
        side, layers = int(rng.integers(24, 40)), int(rng.integers(2, 4))
        ijk = np.stack(np.meshgrid(np.arange(side), np.arange(layers), np.arange(side)), -1).reshape(-1, 3)
        c = pitch * ijk.astype(np.float64) + [0, r0, 0]
    elif kind == 2:    # a packed cube
        side = int(rng.integers(11, 15))
        ijk = np.stack(np.meshgrid(np.arange(side), np.arange(side), np.arange(side)), -1).reshape(-1, 3)
        c = pitch * ijk.astype(np.float64) + [0, r0, 0]
    else:              # a long strip (very different extents per axis)
        a, b = int(rng.integers(120, 200)), int(rng.integers(9, 14))
        ij = np.stack(np.meshgrid(np.arange(a), np.arange(b)), -1).reshape(-1, 2)
        c = np.stack([pitch * ij[:, 0], np.full(len(ij), r0), pitch * ij[:, 1]], 1)
    n = len(c)
    c = c - c.mean(0) + [0, r0, 0] * np.array([0, 1, 0])
    c += rng.uniform(0, float(rng.uniform(0.0, 0.6)) * (pitch - 2 * r0) + 0.3 * pitch * (rng.random() < 0.3), c.shape) * [1, float(rng.random() < 0.3), 1]
    r = r0 * rng.uniform(float(rng.uniform(0.75, 1.0)), 1.0, n)
    sph = np.concatenate([c, r[:, None]], 1)
    extras = []
    if rng.random() < 0.7:
        extras.append([0.0, float(c[:, 1].min() - r0) - 1000.0 - float(rng.uniform(0, 3)) * (rng.random() < 0.3), 0.0, 1000.0])   # ground
    for _ in range(int(rng.integers(0, 4))):
        extras.append(list(rng.uniform(-1, 1, 3) * np.ptp(c, 0) * 0.6 + [0, 2, 0]) + [float(rng.uniform(3, 12)) * r0])           # big spheres (tested for every ray)
    if rng.random() < 0.2:
        extras.append(list(rng.uniform(-1, 1, 3) * np.ptp(c, 0) * 0.4) + [float(rng.uniform(1e-7, 1e-5))])                       # a degenerate one
    sph = np.concatenate([sph] + [np.asarray([e]) for e in extras]).astype(np.float32)
    tex = [(ptgpu.TEX_CONSTANT, (0.5, 0.5, 0.5), -1, -1, 0.0), (ptgpu.TEX_CONSTANT, (0.8, 0.3, 0.3), -1, -1, 0.0)]
    mats = [(ptgpu.MAT_LAMBERTIAN, (0, 0, 0), 0.0, 0), (ptgpu.MAT_LAMBERTIAN, (0, 0, 0), 0.0, 1), (ptgpu.MAT_METAL, (0.8, 0.8, 0.8), float(rng.uniform(0, 0.4)), -1),
            (ptgpu.MAT_METAL, (0.9, 0.9, 0.9), 0.0, -1), (ptgpu.MAT_DIELECTRIC, (0, 0, 0), 1.5, -1)]
    half = 0.5 * float(np.linalg.norm(np.ptp(c, 0)))
    where = rng.random()
    dist = half * (rng.uniform(0.0, 0.6) if where < 0.2 else rng.uniform(0.8, 2.5) if where < 0.7 else rng.uniform(3, 12) if where < 0.9 else rng.uniform(50, 5000))
    look = rng.normal(0, 1, 3)
    look[1] = abs(look[1]) * 0.6 + 0.05
    look = look / np.linalg.norm(look) * max(dist, 0.5 * r0)
    desc = ptgpu.SceneDesc(sph, rng.integers(0, len(mats), len(sph)).astype(np.uint32), mats, tex)
    desc._materials, desc._textures = mats, tex
    return desc, look, n


def bvh_over(sph, rng):
    """A caller's tree as BVHNode::new would build one (bvh.rs:64-94: random axis, median split; boxes = unions of the children's): what a
    BVH world's gates and DFS ranks are derived from."""
    lo, hi = sph[:, :3] - np.abs(sph[:, 3:4]), sph[:, :3] + np.abs(sph[:, 3:4])
    minmax, lr = [], []

    def build(idx):
        if len(idx) == 1:
            return ~int(idx[0])
        axis = int(rng.integers(0, 3))
        idx = idx[np.argsort(lo[idx, axis], kind="stable")]
        me = len(minmax)
        minmax.append(None), lr.append(None)
        left, right = build(idx[:len(idx) // 2]), build(idx[len(idx) // 2:])
        minmax[me] = np.concatenate([lo[idx].min(0), hi[idx].max(0)])
        lr[me] = (left, right)
        return me

    sys.setrecursionlimit(10000)
    root = build(np.arange(len(sph)))
    return (np.asarray(minmax, np.float32), np.asarray(lr, np.int32)), root


def run(first, count):
    W, H, S = 96, 64, 2
    grids = bad = 0
    for seed in range(first, first + count):
        desc, look, n = field(seed)
        cam = camera(look, float(np.random.default_rng(seed + 7).uniform(20, 70)), W / H)
        bvh = seed % 3 == 0   # a third of the worlds with BVHNode::ray_hit semantics (ancestor gates, DFS-rank ties)
        if bvh:
            nodes, root = bvh_over(desc.spheres, np.random.default_rng(seed + 11))
            desc = ptgpu.SceneDesc(desc.spheres, desc.sphere_material, desc._materials, desc._textures, bvh_nodes=nodes, bvh_root=root)
        sc = ptgpu.Scene(desc, 0)
        p = ptgpu.PtParams(W, H, S, 10, 0, 1 if bvh else 0)
        # the reference side: a list world on the exact VALU scan; a BVH world (whose flavour is always a tree kernel: 4 | 64 would select the grid walk
        # again) on the BINARY tree without a grid (524288 | 2048) -- another traversal, the same gates and DFS-rank ties
        sc.set_tuning(0, (524288 | 2048) if bvh else (4 | 64))
        exact = np.zeros((H, W, 3), np.float32)
        rays_exact = sc.update(p, cam, 0, exact)
        assert not sc.last_kernel_choice()["name"].startswith("grid<"), sc.last_kernel_choice()["name"]
        sc.set_tuning(0, 0)
        out = np.zeros((H, W, 3), np.float32)
        rays = sc.update(p, cam, 0, out)
        name = sc.last_kernel_choice()["name"]
        grids += name.startswith("grid<")
        if rays != rays_exact or not np.array_equal(out, exact, equal_nan=True):
            bad += 1
            print("MISMATCH seed %d%s: %d spheres, kernel %s, rays %d vs %d, %d pixels differ" % (seed, " (BVH world)" if bvh else "", n, name, rays, rays_exact, int((out != exact).any(2).sum())), flush=True)
        sc.close()
    print("grid soak: seeds %d..%d, %d on the grid kernel, %d mismatches" % (first, first + count - 1, grids, bad))
    return grids, bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 200)[1] else 0)
