"""Development aid (GPU box): the uniform cell grid (csrc/pt_grid.h) against the 4-wide tree (tuning bit 524288) on synthetic sphere
clouds -- which one should kernel selection take where? Usage: python tools/grid_ab.py"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pthost = importlib.import_module("pathtrace-rs_amd.pthost")
ptgpu = pthost.ptgpu


def cloud(seed, n, half, rlo, rhi, flat=False, cluster=False):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-half, half, (n, 3))
    if flat:
        c[:, 1] = rng.uniform(0.0, 2.0 * rhi, n)
    if cluster:   # a third of the spheres in a tight cluster: an uneven spread
        k = n // 3
        c[:k] = rng.normal(0.0, half * 0.03, (k, 3))
    r = rng.uniform(rlo, rhi, n)
    sph = np.concatenate([c, r[:, None]], axis=1).astype(np.float32)
    sph[0] = [0.0, -1000.0 - half, 0.0, 1000.0]
    tex = [(ptgpu.TEX_CONSTANT, (0.5, 0.5, 0.5), -1, -1, 0.0), (ptgpu.TEX_CONSTANT, (0.8, 0.3, 0.3), -1, -1, 0.0)]
    mats = [(ptgpu.MAT_LAMBERTIAN, (0, 0, 0), 0.0, 0), (ptgpu.MAT_LAMBERTIAN, (0, 0, 0), 0.0, 1), (ptgpu.MAT_METAL, (0.8, 0.8, 0.8), 0.1, -1), (ptgpu.MAT_DIELECTRIC, (0, 0, 0), 1.5, -1)]
    return ptgpu.SceneDesc(sph, rng.integers(0, 4, n).astype(np.uint32), mats, tex)


def lattice(seed, nx, ny, nz, spacing, rlo, rhi):
    """a jittered lattice (the fields tests/test_host_cpu.py and tools/grid_soak.py build): nx x ny x nz spheres `spacing` apart"""
    rng = np.random.default_rng(seed)
    ijk = np.stack(np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz)), -1).reshape(-1, 3).astype(np.float64)
    c = spacing * (ijk - 0.5 * np.array([nx - 1, 0, nz - 1])) + rng.uniform(0, 0.2, ijk.shape) * [1, 0.5, 1]
    sph = np.concatenate([c, rng.uniform(rlo, rhi, (len(c), 1))], 1).astype(np.float32)
    sph = np.concatenate([np.array([[0, -1000.5, 0, 1000.0]], np.float32), sph])
    tex = [(ptgpu.TEX_CONSTANT, (0.5, 0.5, 0.5), -1, -1, 0.0), (ptgpu.TEX_CONSTANT, (0.8, 0.3, 0.3), -1, -1, 0.0)]
    mats = [(ptgpu.MAT_LAMBERTIAN, (0, 0, 0), 0.0, 0), (ptgpu.MAT_LAMBERTIAN, (0, 0, 0), 0.0, 1), (ptgpu.MAT_METAL, (0.8, 0.8, 0.8), 0.1, -1), (ptgpu.MAT_DIELECTRIC, (0, 0, 0), 1.5, -1)]
    return ptgpu.SceneDesc(sph, rng.integers(0, 4, len(sph)).astype(np.uint32), mats, tex)


def camera(look, vfov, aspect):
    """camera.rs:21-53 in numpy (an A/B tool: nobody compares these frames with the reference's)"""
    look = np.asarray(look, np.float64)
    hh = np.tan(np.radians(vfov) / 2.0)
    hw = aspect * hh
    focus = np.linalg.norm(look)
    w = look / focus
    u = np.cross([0.0, 1.0, 0.0], w)
    u /= np.linalg.norm(u)
    v = np.cross(w, u)
    llc = look - hw * focus * u - hh * focus * v - focus * w
    f = np.concatenate([look, llc, 2 * hw * focus * u, 2 * hh * focus * v, u, v, w, [0.0, 0.0, 0.0]]).astype(np.float32)
    return ptgpu.PtCamera.from_floats(f)


def main():
    W, H, S = 1200, 800, 16
    cases = [("3D cloud 3k r .05-.4", dict(n=3000, half=12, rlo=0.05, rhi=0.4)), ("3D cloud 20k r .05-.3", dict(n=20000, half=20, rlo=0.05, rhi=0.3)),
             ("3D cloud 100k r .02-.15", dict(n=100000, half=20, rlo=0.02, rhi=0.15)), ("flat 10k r .1-.3", dict(n=10000, half=25, rlo=0.1, rhi=0.3, flat=True)),
             ("clustered 10k r .05-.3", dict(n=10000, half=20, rlo=0.05, rhi=0.3, cluster=True)), ("3D cloud 1.5k r .2-.8", dict(n=1500, half=15, rlo=0.2, rhi=0.8)),
             ("3D cloud 10k equal r .2", dict(n=10000, half=20, rlo=0.2, rhi=0.2)), ("dense 3D 2.5k r .05-.4", dict(n=2500, half=6, rlo=0.05, rhi=0.4)),
             ("dense 3D 10k equal r .2", dict(n=10000, half=5, rlo=0.2, rhi=0.2)), ("dense 3D 10k r .15-.25", dict(n=10000, half=7, rlo=0.15, rhi=0.25)),
             ("flat 10k equal r .2", dict(n=10000, half=25, rlo=0.2, rhi=0.2, flat=True)), ("flat 40k equal r .1", dict(n=40000, half=25, rlo=0.1, rhi=0.1, flat=True)),
             ("flat sparse 3k r .2", dict(n=3000, half=40, rlo=0.2, rhi=0.2, flat=True))]
    # a density series: 10 000 spheres of r = 0.2 (and of r 0.1-0.3) in cubes of growing size; layers of growing sparsity
    cases += [("cube 10k r .2 half %g" % h, dict(n=10000, half=h, rlo=0.2, rhi=0.2)) for h in (4, 6, 7, 8, 10, 14)]
    cases += [("cube 10k r .1-.3 half %g" % h, dict(n=10000, half=h, rlo=0.1, rhi=0.3)) for h in (5, 7, 10)]
    cases += [("flat 10k r .2 half %g" % h, dict(n=10000, half=h, rlo=0.2, rhi=0.2, flat=True)) for h in (15, 20, 35)]
    cases += [("lattice strip 150x1x10 .7", dict(lat=(150, 1, 10, 0.7, 0.22, 0.3), half=30)), ("lattice 40x3x40 .7", dict(lat=(40, 3, 40, 0.7, 0.22, 0.3), half=14)),
              ("lattice 100x1x100 .5 r.2", dict(lat=(100, 1, 100, 0.5, 0.2, 0.2), half=25)), ("lattice 22x22x22 .6", dict(lat=(22, 22, 22, 0.6, 0.15, 0.25), half=7)),
              ("lattice 60x1x60 1.0 r.2-.3", dict(lat=(60, 1, 60, 1.0, 0.2, 0.3), half=30)), ("lattice 300x1x12 .6", dict(lat=(300, 1, 12, 0.6, 0.2, 0.25), half=50)),
              ("lattice strip 150x1x10 .6", dict(lat=(150, 1, 10, 0.6, 0.22, 0.3), half=30)), ("lattice strip 200x1x8 .5", dict(lat=(200, 1, 8, 0.5, 0.18, 0.22), half=30)),
              ("lattice strip 120x2x12 .55", dict(lat=(120, 2, 12, 0.55, 0.2, 0.26), half=25))]
    if len(sys.argv) > 1:
        cases = [c for c in cases if sys.argv[1] in c[0]]
    for name, kw in cases:
        desc = lattice(7, *kw["lat"]) if "lat" in kw else cloud(7, **kw)
        half = kw["half"]
        cam = camera([1.6 * half, 0.8 * half + 1.0, 1.2 * half], 40.0, W / H)
        sc = ptgpu.Scene(desc, 0)
        p = ptgpu.PtParams(W, H, S, 10, 0, 0)
        out = np.zeros((H, W, 3), np.float32)
        line = "%-26s" % name
        try:   # the plan the scene gets (development builds: PTGPU_GRID_OCC / PTGPU_GRID_PER_CELL lift the planner's thresholds)
            g = ptgpu.debug_cell_grid(desc)
            cells = int(np.prod(g["n"]))
            line += " %3dx%3dx%3d h %.2f occ %3.0f%% sph/cell %.2f rec/cell %.2f " % (g["n"][0], g["n"][1], g["n"][2], g["h"][0], 100.0 * g["occupied"], g["items_per_cell"], g["records"].shape[0] / cells)
        except ptgpu.PtError:
            line += " (no grid)".ljust(58)
        for variant in (0, 524288):
            sc.set_tuning(0, variant | 8192)
            for _ in range(3):
                rays = sc.update(p, cam, 0, out)
            ms = sc.last_kernel_ms()
            line += "  %-16s %7.2f ms %6.2f Grays/s" % (sc.last_kernel_choice()["name"], ms, rays / ms * 1e-6)
        print(line, flush=True)
        sc.close()


if __name__ == "__main__":
    main()
