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
    for name, kw in cases:
        desc = cloud(7, **kw)
        half = kw["half"]
        cam = camera([1.6 * half, 0.8 * half + 1.0, 1.2 * half], 40.0, W / H)
        sc = ptgpu.Scene(desc, 0)
        p = ptgpu.PtParams(W, H, S, 10, 0, 0)
        out = np.zeros((H, W, 3), np.float32)
        line = "%-26s" % name
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
