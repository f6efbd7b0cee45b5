"""Parity tests proper (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs, against the committed golden fixtures, and through size-independent properties
at BASELINE.json's full sizes.

Bars: ray_count and every float bit-exact for scenes whose control flow uses no libm call
(small / aras / random_spheres, list and BVH); 2e-6 absolute per channel where colour passes through
sinf() (noise textures); see DESIGN.md "Parity".
"""
import ctypes as C
import glob
import json
import subprocess
import importlib.util
import os
import re
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
NOISE_ATOL = 2e-6
WORLD_PRESETS = ("random", "simple_light", "cornell", "cornell_smoke")   # non-sphere Hitable arms (general kernel)


@pytest.fixture(scope="module")
def pthost(ptgpu):
    name = "pathtrace_rs_amd_pthost"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "pathtrace-rs_amd", "pthost.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _gpu_render(ptgpu, oracle, preset, W, H, S, use_bvh, depth=10, frame=0, prev=None, variant=0):
    """Render on the GPU the scene the ORACLE built (same description fed to both sides)."""
    osc = oracle.OracleScene(preset, W, H, use_bvh=use_bvh)
    ex = osc.export()
    sc = ptgpu.Scene(oracle.to_ptgpu_desc(ptgpu, ex), 0)
    if variant:
        sc.set_tuning(0, variant)
    out = np.zeros((H, W, 3), np.float32) if prev is None else prev.copy()
    rays = sc.update(ptgpu.PtParams(W, H, S, depth, 0, 1 if use_bvh else 0), ptgpu.PtCamera.from_floats(ex["camera"]),
                     frame, out)
    sc.close()
    return osc, out, rays


def _report(ref, out):
    d = np.abs(ref - out)
    bad = (d > 0).any(axis=-1)
    return "mismatching pixels %d / %d, max |d| %.3e" % (bad.sum(), bad.size, np.nanmax(d))


# ---- small sizes: whole frame vs oracle ----------------------------------------------------------
@pytest.mark.parametrize("preset,W,H,S,bvh", [
    ("small", 200, 100, 4, False),           # BASELINE config 1
    ("small", 200, 100, 4, True),
    ("aras", 160, 90, 4, False),
    ("aras", 160, 90, 4, True),
    ("random_spheres", 120, 80, 4, False),
    ("random_spheres", 120, 80, 4, True),
    ("random_spheres", 33, 17, 3, False),    # ragged: not a multiple of the 8x8 work tiles
])
def test_exact_parity(ptgpu, oracle, preset, W, H, S, bvh):
    osc, out, rays = _gpu_render(ptgpu, oracle, preset, W, H, S, bvh)
    ref, ref_rays = osc.update(S)
    assert rays == ref_rays, "ray_count %d vs oracle %d; %s" % (rays, ref_rays, _report(ref, out))
    assert np.array_equal(ref, out), _report(ref, out)


def test_scan_variants_agree(ptgpu, oracle):
    """Kernel variants are the same function: MFMA-prefiltered scan in one 768-thread workgroup per CU with the
    attenuation stacks in LDS (default), the same in three 256-thread workgroups with the stacks in HBM (2), exact
    VALU scan from LDS (4), exact scan from HBM/L2 (4|1), exact scan with the stack in HBM (4|2), no tile reordering (32), no tile culling (1024)."""
    osc, a, ra = _gpu_render(ptgpu, oracle, "random_spheres", 96, 64, 4, False, variant=0)
    ref, ref_rays = osc.update(4)
    assert ra == ref_rays and np.array_equal(a, ref)
    for variant in (2, 4, 5, 6, 32, 1024, 1048576):
        _, b, rb = _gpu_render(ptgpu, oracle, "random_spheres", 96, 64, 4, False, variant=variant)
        assert rb == ref_rays and np.array_equal(b, ref), "variant %d: %s" % (variant, _report(ref, b))


@pytest.mark.parametrize("preset,W,H,S,bvh,depth", [   # (frames of more than two pixels per lane of the GPU: smaller ones run the kernel without pools)
    ("random_spheres", 1203, 797, 3, False, 10),   # ragged frame: claims straddle the frame's edge; 32 pool entries per wave; one launch
    ("random_spheres", 1200, 800, 12, False, 20),  # two launches (measuring launch + ordered frame kernel), 16 entries
    ("random_spheres", 1100, 900, 4, True, 14),    # BVH world (gates in LDS), 8 entries
    ("aras", 1280, 720, 16, False, 10),            # BASELINE config 2
    ("random", 1200, 800, 3, False, 10),           # MovingSphere world
])
def test_pixel_pools_change_when_a_pixel_starts_never_its_value(ptgpu, pthost, oracle, preset, W, H, S, bvh, depth):
    """The 1024-thread frame kernels keep a pool of ready-to-start pixels per wave in LDS (csrc/pt_kernel.h POOL): a freed lane takes the next
    entry at once instead of waiting for a batched refill. The same frame with the pools (default), with the batched refill of rounds 2-5
    (development bit 1048576), with and without the cooperative hand-over, measuring every frame, and on the exact scan / general kernel: one
    set of bits, the oracle's."""
    ref, ref_rays = oracle.OracleScene(preset, W, H, use_bvh=bvh).update(S, depth)
    hs = pthost.HostScene(preset, W, H, samples=S, use_bvh=bvh, device=0)
    sc = hs.device_scene()
    p = ptgpu.PtParams(W, H, S, depth, 0, 1 if bvh else 0)
    for variant in (0, 1048576, 65536, 1048576 | 65536, 8192, 128 if preset == "random" else 4):
        sc.set_tuning(0, variant)
        out = np.zeros((H, W, 3), np.float32)
        rays = sc.update(p, hs.camera, 0, out)
        assert rays == ref_rays and np.array_equal(out, ref), "variant %d: %s" % (variant, _report(ref, out))
        ch = sc.last_kernel_choice()
        if variant in (0, 65536, 8192):   # (the pools were really on: the kernel's name says so)
            assert ch["name"].endswith(",pool>") and ch["pool_slots"] == {10: 32, 20: 16, 14: 8}[depth] // (2 if preset == "random" else 1), ch
        else:
            assert ch["pool_slots"] == 0 and "pool" not in ch["name"], ch
    sc.set_tuning(0, 0)


def test_pixel_pool_soak_slice(ptgpu):
    """A slice of tools/pool_soak.py: seeded sphere clouds at frames of more than two pixels per lane (ragged edges, one launch and two, 32 / 16 / 8
    pool entries, progressive frames): the pooled kernel, the batched refill, pools without hand-over and the exact scan render the same floats and
    the same ray count. (Round 6's soak: seeds 1000..4999, 3 203 worlds on pooled kernels, 0 mismatches.)"""
    spec = importlib.util.spec_from_file_location("pool_soak", os.path.join(ROOT, "tools", "pool_soak.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    pooled, bad = mod.run(7000, 30, verbose=False)
    assert bad == 0 and pooled >= 15, (pooled, bad)


@pytest.mark.parametrize("preset,W,H,S", [("random_spheres", 1200, 800, 2), ("aras", 640, 360, 8)])
def test_mfma_prefilter_never_drops_a_positive_discriminant(ptgpu, pthost, preset, W, H, S):
    """Verify mode (variant 8): for EVERY ray of the frame, every sphere whose reference discriminant
    (sphere.rs:33-37) is > 0 must have been flagged by the MFMA prefilter (or be in the always-exact
    'large' set). The prefilter may over-report, never under-report. The same run audits the tile culling: the
    tile holding each ray's brute-force winner must be one the ray's lane asks for (a culled winner is a miss)."""
    hs = pthost.HostScene(preset, W, H, samples=S, device=0)
    sc = hs.device_scene()
    p = ptgpu.PtParams(W, H, S, 10, 0, 0)
    sc.set_tuning(0, 4)
    exact = np.zeros((H, W, 3), np.float32)
    rays_exact = sc.update(p, hs.camera, 0, exact)
    sc.set_tuning(0, 8)
    sc.debug_counters(reset=True)
    out = np.zeros((H, W, 3), np.float32)
    rays = sc.update(p, hs.camera, 0, out)
    c = sc.debug_counters()
    sc.set_tuning(0, 0)
    assert c["exact_positives"] > rays_exact // 2            # the check really ran over the frame
    assert c["misses"] == 0, c
    assert c["candidates"] < 2 * c["exact_positives"], c     # and the filter is still selective
    assert rays == rays_exact and np.array_equal(out, exact)


def _random_scene(ptgpu, oracle, seed, n, half, rmax, cam_dist, look_perm=(0, 1, 2)):
    """Synthetic sphere cloud (not a reference preset) to stress the prefilter's margins and the internal tree."""
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-half, half, size=(n, 3)).astype(np.float32)
    radii = rng.uniform(0.05, rmax, size=n).astype(np.float32)
    spheres = np.concatenate([centres, radii[:, None]], axis=1)
    spheres[0] = [0.0, -1000.0 - half, 0.0, 1000.0]          # a huge ground sphere like the presets
    kinds = rng.integers(0, 3, size=n)
    materials, textures = [], []
    for i in range(n):
        if kinds[i] == 0:
            textures.append((ptgpu.TEX_CONSTANT, rng.uniform(0.1, 0.9, 3), -1, -1, 0.0))
            materials.append((ptgpu.MAT_LAMBERTIAN, (0, 0, 0), 0.0, len(textures) - 1))
        elif kinds[i] == 1:
            materials.append((ptgpu.MAT_METAL, rng.uniform(0.3, 1.0, 3), float(rng.uniform(0, 0.5)), -1))
        else:
            materials.append((ptgpu.MAT_DIELECTRIC, (0, 0, 0), 1.5, -1))
    cam = np.zeros(24, np.float32)
    look = np.array([cam_dist, 0.3 * cam_dist + 1.0, 0.2 * cam_dist], np.float32)[list(look_perm)]
    at, up = np.zeros(3, np.float32), np.array([0, 1, 0], np.float32)
    vfov = float(np.degrees(2 * np.arctan(1.2 * half / np.linalg.norm(look))))
    oracle.lib().ora_camera_new(look.ctypes.data, at.ctypes.data, up.ctypes.data, vfov, 1.5, 0.05,
                                float(np.linalg.norm(look)), 0.0, 1.0, cam.ctypes.data)
    desc = ptgpu.SceneDesc(spheres, np.arange(n, dtype=np.uint32), materials, textures)
    return desc, ptgpu.PtCamera.from_floats(cam)


@pytest.mark.parametrize("seed,n,half,rmax,cam_dist", [
    (1, 64, 4.0, 0.8, 30.0), (2, 300, 15.0, 0.5, 60.0), (3, 700, 30.0, 1.0, 150.0),
    (4, 500, 10.0, 0.2, 800.0),      # far camera: margins grow with |o - c0|^2, the filter must stay conservative
    (5, 2500, 40.0, 0.6, 120.0),     # beyond the scan limit: list world walks the internal tree
])
def test_synthetic_scenes_all_scan_paths_agree(ptgpu, oracle, seed, n, half, rmax, cam_dist):
    desc, cam = _random_scene(ptgpu, oracle, seed, n, half, rmax, cam_dist)
    sc = ptgpu.Scene(desc, 0)
    W, H, S = 150, 100, 4
    p = ptgpu.PtParams(W, H, S, 10, 0, 0)
    sc.set_tuning(0, 4 | 64)                                  # exact VALU scan: the reference semantics
    exact = np.zeros((H, W, 3), np.float32)
    rays_exact = sc.update(p, cam, 0, exact)
    sc.set_tuning(0, 0)                                       # default: MFMA prefilter or internal tree
    out = np.zeros((H, W, 3), np.float32)
    rays = sc.update(p, cam, 0, out)
    assert rays == rays_exact and np.array_equal(out, exact), _report(exact, out)
    if n <= 768:
        sc.set_tuning(0, 8)
        sc.debug_counters(reset=True)
        ver = np.zeros((H, W, 3), np.float32)
        sc.update(p, cam, 0, ver)
        c = sc.debug_counters()
        assert c["misses"] == 0 and c["exact_positives"] > 0, c
        assert np.array_equal(ver, exact)
    sc.close()


@pytest.mark.parametrize("perm", [(0, 1, 2), (2, 0, 1), (1, 2, 0), (2, 1, 0)])
@pytest.mark.parametrize("cam_dist", [300.0, 3000.0])
def test_far_camera_on_every_axis_keeps_tile_culling_exact(ptgpu, oracle, perm, cam_dist):
    """Small spheres seen from 10-100x the scene extent: the reference's f32 discriminant then accepts rays passing well
    outside a sphere, so the tile culling must widen (or switch itself off) with the same error budget as the MFMA margin.
    Camera placed towards each axis in turn (the tiles are sorted along ONE of them); default kernel == exact scan."""
    desc, cam = _random_scene(ptgpu, oracle, 11, 500, 10.0, 0.2, cam_dist, look_perm=perm)
    sc = ptgpu.Scene(desc, 0)
    W, H, S = 150, 100, 4
    p = ptgpu.PtParams(W, H, S, 10, 0, 0)
    sc.set_tuning(0, 4 | 64)
    exact = np.zeros((H, W, 3), np.float32)
    rays_exact = sc.update(p, cam, 0, exact)
    for variant in (0, 1024):
        sc.set_tuning(0, variant)
        out = np.zeros((H, W, 3), np.float32)
        rays = sc.update(p, cam, 0, out)
        assert rays == rays_exact and np.array_equal(out, exact), "variant %d: %s" % (variant, _report(exact, out))
    sc.close()


@pytest.mark.parametrize("preset,W,H,S,bvh", [
    ("two_perlin_spheres", 160, 90, 4, False),
    ("two_perlin_spheres", 160, 90, 4, True),
    ("perlin_spheres", 96, 54, 2, True),
    ("perlin_spheres", 96, 54, 2, False),   # list world on 10k spheres: walks the internal tree, list tie rule
])
def test_noise_parity(ptgpu, oracle, preset, W, H, S, bvh):
    osc, out, rays = _gpu_render(ptgpu, oracle, preset, W, H, S, bvh)
    ref, ref_rays = osc.update(S)
    assert rays == ref_rays                      # noise colour never feeds control flow
    np.testing.assert_allclose(out, ref, rtol=0, atol=NOISE_ATOL)


@pytest.mark.parametrize("depth", [0, 1, 3, 50])
def test_max_depth_edges(ptgpu, oracle, depth):
    """depth 0: primary rays only; depth 50: the attenuation stack leaves LDS for the HBM fallback."""
    osc, out, rays = _gpu_render(ptgpu, oracle, "small", 64, 32, 3, False, depth=depth)
    ref, ref_rays = osc.update(3, max_depth=depth)
    assert rays == ref_rays and np.array_equal(ref, out), _report(ref, out)
    if depth == 0:
        assert rays == 64 * 32 * 3


def test_progressive_frames_blend_like_scene_update(ptgpu, oracle):
    """scene.rs:86-87,99-101,114-116: frame_num enters the seed and the running-mean blend."""
    W, H, S = 80, 40, 2
    osc = oracle.OracleScene("small", W, H)
    ex = osc.export()
    sc = ptgpu.Scene(oracle.to_ptgpu_desc(ptgpu, ex), 0)
    cam, p = ptgpu.PtCamera.from_floats(ex["camera"]), ptgpu.PtParams(W, H, S, 10, 0, 0)
    ref = np.zeros((H, W, 3), np.float32)
    out = np.zeros((H, W, 3), np.float32)
    for frame in range(4):
        _, rr = osc.update(S, 10, frame, buffer=ref)
        rg = sc.update(p, cam, frame, out)
        assert rg == rr and np.array_equal(ref, out), "frame %d: %s" % (frame, _report(ref, out))
    sc.close()


@pytest.mark.parametrize("preset,bvh", [("random_spheres", False), ("random_spheres", True), ("perlin_spheres", True),
                                        ("cornell_smoke", False), ("simple_light", True), ("random", False)])   # (general-world kernel too)
def test_work_order_from_the_previous_frame_never_changes_a_pixel(ptgpu, pthost, preset, bvh):
    """A frame of the view the scene rendered last is ordered by the rays each tile took in that frame (measured by the frame
    kernel) instead of by a 1-spp pilot pass. The order of the work must not change a pixel or the ray count: progressive
    frames 0..3 (a camera change in between) equal the run that pilots every frame (variant 8192) and the run with no
    ordering at all (variant 32)."""
    W, H, S = 512, 320, 32          # 2 560 work tiles, 32 spp: large enough for every ordering scheme to be used
    runs = {}
    # 8192: every frame measures its own order with its first sample (two launches)
    # 262144: ... over every tile (default on the MFMA list kernels: one colour of a checkerboard, the other tiles start in the second launch)
    for name, variant in (("reuse", 0), ("pilot", 8192), ("pilot_all", 8192 | 262144), ("unordered", 32)):
        hs = pthost.HostScene(preset, W, H, samples=S, use_bvh=bvh, device=0)
        sc = hs.device_scene()
        sc.set_tuning(0, variant)
        p = ptgpu.PtParams(W, H, S, 10, 0, 1 if bvh else 0)
        camf = np.ctypeslib.as_array(C.cast(C.pointer(hs.camera), C.POINTER(C.c_float)), shape=(24,)).copy()
        camf[0] += np.float32(0.25)                        # origin.x
        other = ptgpu.PtCamera.from_floats(camf)
        out, side, rays = np.zeros((H, W, 3), np.float32), np.zeros((H, W, 3), np.float32), []
        for frame in range(4):
            rays.append(sc.update(p, hs.camera, frame, out))
            if frame == 1:
                rays.append(sc.update(p, other, 0, side))    # another view in between: its key differs, the pilot runs
        runs[name] = (out, side, rays)
    for name in ("pilot", "pilot_all", "unordered"):
        assert runs["reuse"][2] == runs[name][2], (runs["reuse"][2], runs[name][2])
        assert np.array_equal(runs["reuse"][0], runs[name][0]) and np.array_equal(runs["reuse"][1], runs[name][1])


@pytest.mark.parametrize("W,H", [(520, 328), (519, 321), (1000, 264), (264, 1000)])   # 65 x 41, 65 x 41 (ragged edges), 125 x 33, 33 x 125 tiles: odd counts both ways
def test_checkerboard_measuring_with_odd_tile_counts(ptgpu, pthost, W, H):
    """The measuring launch of a new view traces one colour of a checkerboard of 8x8 tiles (include/ptgpu.h, tuning bit 262144 = every
    tile); the list of the measured colour and the neighbour means are indexed by tile row and column. Same frame, same ray count as
    measuring every tile and as no ordering at all, whatever the parity of the tile grid."""
    S = 16
    runs = {}
    for name, variant in (("checker", 8192), ("all", 8192 | 262144), ("unordered", 32)):
        hs = pthost.HostScene("random_spheres", W, H, samples=S, device=0)
        sc = hs.device_scene()
        sc.set_tuning(0, variant)
        out = np.zeros((H, W, 3), np.float32)
        rays = [sc.update(ptgpu.PtParams(W, H, S, 10, 0, 0), hs.camera, f, out) for f in range(2)]
        assert sc.last_kernel_choice()["name"].startswith("mfma<")
        runs[name] = (out, rays)
    for name in ("all", "unordered"):
        assert runs["checker"][1] == runs[name][1] and np.array_equal(runs["checker"][0], runs[name][0]), name


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_work_order_on_seeded_worlds_at_random_frame_sizes(ptgpu, oracle, seed):
    """A slice of tools/order_soak.py: seeded sphere worlds (list and BVH, 40-700 spheres) at random frame sizes and sample counts
    (>= 12: the two-launch path of scene.rs:73-121's frame) -- checkerboard measuring, every tile measured, natural order and the
    hand-over switched off produce the same two progressive frames and ray counts."""
    ob = oracle
    rng = np.random.default_rng(seed)
    W, H, S = int(rng.integers(200, 900)), int(rng.integers(150, 700)), int(rng.choice([12, 13, 16, 24, 40]))
    n = int(rng.choice([40, 150, 400, 700]))
    bvh = seed % 3 == 0
    w = _random_sphere_world(ob, seed, n, W, H, float(rng.uniform(3, 12)), float(rng.uniform(0.2, 1.0)))
    osc = ob.OracleScene.from_world(w["hitables"], w["transforms"], w["materials"], w["textures"], w["camera"], W, H, sky=w["sky"], use_bvh=bvh)
    ex = osc.export()
    frames, ordered = {}, 0
    for variant in (8192, 8192 | 262144, 32, 8192 | 65536):
        sc = ptgpu.Scene(ob.to_ptgpu_world_desc(ptgpu, ex), 0)
        sc.set_tuning(0, variant)
        out = np.zeros((H, W, 3), np.float32)
        rays = [sc.update(ptgpu.PtParams(W, H, S, 10, 0, 1 if bvh else 0), ptgpu.PtCamera.from_floats(ex["camera"]), f, out) for f in range(2)]
        ordered += int(bool(sc.last_kernel_choice()["ordered"]))
        sc.close()
        frames[variant] = (out, rays)
    osc.close()
    assert ordered == 3   # (tuning 32 is the natural order)
    for v, (out, rays) in frames.items():
        assert rays == frames[32][1] and np.array_equal(out, frames[32][0], equal_nan=True), v


@pytest.mark.parametrize("preset,bvh,W,H,S,frames,depth", [
    ("random_spheres", False, 96, 64, 48, 2, 10),     # 6 144 pixels for 262 144 lanes: nearly every pixel is finished by a worker
    ("random_spheres", True, 96, 64, 48, 1, 10),      # BVH world on the list kernel: the ancestor-AABB gate + DFS-rank ties inside the workers' scan
    ("random", False, 96, 64, 32, 1, 10),             # Sphere + MovingSphere world: rays keep their time in the workers as well
    ("aras", False, 160, 90, 32, 2, 10),              # diffuse lights, 46 spheres (one register set per lane)
    ("random_spheres", False, 600, 400, 24, 1, 10),   # more pixels than lanes for a while: the hand-over only starts once the list is dry
    ("random_spheres", False, 96, 64, 32, 1, 33),     # depth 33: the 768-thread kernel (12 waves per workgroup), 33 attenuation levels in the workers' lanes
    ("random_spheres", True, 96, 64, 32, 1, 0),       # depth 0: every path ends at its first hit
])
def test_cooperative_handover_never_changes_a_pixel(ptgpu, pthost, oracle, preset, bvh, W, H, S, frames, depth):
    """csrc/pt_coop.h: once the work list is dry, waves that have run out of pixels finish pixels handed over by busy waves, all 64
    lanes on each ray (scene.rs:96-111 makes a pixel ONE serial chain of samples; this shortens the chain). The pixel's RNG stream,
    colour sum and counters travel with it, so neither a pixel nor the ray count may depend on who traced it: the frame with the
    hand-over equals the frame without (tuning bit 65536) and the oracle's, bit for bit, over progressive frames -- and the
    counters prove that pixels really changed hands."""
    runs = {}
    for name, variant in (("coop", 0), ("plain", 65536)):
        hs = pthost.HostScene(preset, W, H, samples=S, use_bvh=bvh, device=0)
        sc = hs.device_scene()
        sc.set_tuning(0, variant)
        sc.coop_counters(reset=True)
        p = ptgpu.PtParams(W, H, S, depth, 0, 1 if bvh else 0)
        out, rays = np.zeros((H, W, 3), np.float32), []
        for frame in range(frames):
            rays.append(sc.update(p, hs.camera, frame, out))
        ch = sc.last_kernel_choice()
        runs[name] = (out, rays, sc.coop_counters(), ch)
    assert runs["coop"][3]["coop"] == 1 and runs["plain"][3]["coop"] == 0, (runs["coop"][3], runs["plain"][3])
    assert runs["plain"][2]["pixels"] == 0
    handed = runs["coop"][2]
    if depth > 0:   # (at depth 0 a sample is one ray: no pixel ever has the two dozen estimated rays left that a hand-over asks for)
        assert handed["pixels"] > 0 and handed["rays"] > 0, handed
    if depth == 33:
        assert runs["coop"][3]["block"] == 768, runs["coop"][3]
    if preset == "random_spheres" and W * H <= 16384 and depth == 10:
        assert handed["pixels"] > W * H // 4, "a frame this small should be finished mostly by workers: %r" % (handed,)
    assert runs["coop"][1] == runs["plain"][1], (runs["coop"][1], runs["plain"][1])
    assert np.array_equal(runs["coop"][0], runs["plain"][0]), _report(runs["plain"][0], runs["coop"][0])
    osc = oracle.OracleScene(preset, W, H, use_bvh=bvh)
    ref, ref_rays = np.zeros((H, W, 3), np.float32), []
    for frame in range(frames):
        ref, r = osc.update(S, depth, frame, buffer=ref)
        ref_rays.append(r)
    assert runs["coop"][1] == ref_rays, (runs["coop"][1], ref_rays)
    assert np.array_equal(runs["coop"][0], ref), _report(ref, runs["coop"][0])


# ---- golden fixtures: HIP vs committed oracle output at BASELINE sizes (no oracle call needed) ------
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))))
@pytest.mark.parametrize("mode", ["as_recorded", "other_world"])
def test_golden_fixture(ptgpu, pthost, path, mode):
    g = np.load(path)
    preset, W, H, S = str(g["preset"]), int(g["width"]), int(g["height"]), int(g["samples"])
    bvh = bool(g["use_bvh"])
    if mode == "other_world":
        if preset in WORLD_PRESETS:
            pytest.skip("instances / media: list and BVH worlds differ in the reference itself (aabb.rs:75-100)")
        bvh = not bvh                    # list and BVH worlds give the same image (closest hit either way)
    hs = pthost.HostScene(preset, W, H, samples=S, use_bvh=bvh, device=0)   # scene built by the C++ host
    out = np.zeros((H, W, 3), np.float32)
    sc = hs.device_scene()
    rays = sc.update(ptgpu.PtParams(W, H, S, int(g["depth"]), 0, 1 if bvh else 0), hs.camera, 0, out)
    got = out.reshape(-1, 3)[g["pixels"]]
    if "frame_ray_count" in g.files and mode == "other_world":
        # config 5's whole frame as a LIST world against the BVH world's fixture: the ancestor-AABB gates of bvh.rs:37-62 refuse a handful of
        # grazing hits a list accepts (8 rays of the frame's 733 152 639) -- with 8 262 pixels sampled, one of those paths is among them
        off = (np.abs(got - g["rgb"]) > NOISE_ATOL).any(axis=1)
        assert off.sum() <= 3 and 0 < abs(rays - int(g["frame_ray_count"])) < 100, (int(off.sum()), rays)
    elif "perlin" in preset or preset == "simple_light":
        np.testing.assert_allclose(got, g["rgb"], rtol=0, atol=NOISE_ATOL)
    else:
        assert np.array_equal(got, g["rgb"]), _report(g["rgb"], got)
    if len(g["pixels"]) == W * H:
        assert rays == int(g["ray_count"])
    if "frame_ray_count" in g.files and mode == "as_recorded":   # the oracle rendered the WHOLE frame for this count (tests/golden/make_c5_fullframe.py: ~19 core-hours for config 5)
        assert rays == int(g["frame_ray_count"]), "frame ray count %d vs the oracle's %d" % (rays, int(g["frame_ray_count"]))
    if "tile_rays" in g.files and mode == "as_recorded":   # ... and kept the rays of every 8x8 tile: the kernel that renders config 5 (the cell grid) against the oracle, tile by tile
        mine = sc.tile_rays()
        assert mine.shape == g["tile_rays"].shape and int(mine.sum(dtype=np.uint64)) == rays
        wrong = np.argwhere(mine != g["tile_rays"])
        assert len(wrong) == 0, "%d of %d tiles differ in their ray count, first (row %d, column %d): %d vs the oracle's %d" % (
            len(wrong), mine.size, wrong[0][0], wrong[0][1], mine[tuple(wrong[0])], g["tile_rays"][tuple(wrong[0])])


def test_tile_ray_counts_belong_to_the_last_frame_only(ptgpu, pthost):
    """pt_scene_debug_tile_rays: the counts of a two-launch frame add up to its ray count; a frame that runs as one launch (4 samples) counts nothing,
    and the call then refuses instead of handing back the previous frame's numbers."""
    hs = pthost.HostScene("random_spheres", 320, 200, samples=16, device=0)
    sc = hs.device_scene()
    out = np.zeros((200, 320, 3), np.float32)
    rays = sc.update(ptgpu.PtParams(320, 200, 16, 10, 0, 0), hs.camera, 0, out)
    tiles = sc.tile_rays()
    assert tiles.shape == (25, 40) and int(tiles.sum(dtype=np.uint64)) == rays and tiles.min() >= 64 * 16
    sc.update(ptgpu.PtParams(320, 200, 4, 10, 0, 0), hs.camera, 0, out)
    with pytest.raises(ptgpu.PtError) as e:
        sc.tile_rays()
    assert e.value.code == ptgpu.PT_ERR_UNSUPPORTED


def test_config5_full_frame_cell_grid_equals_the_tree_kernel_tile_by_tile(ptgpu, pthost):
    """BASELINE config 5 at full size (1920 x 1080 x 128, 10 002 spheres, BVH world) on the kernel that renders it -- the uniform cell grid of
    csrc/pt_grid.h, whose stragglers park their walks between calls (round 6) --, on the same kernel with every call walked to its end (2097152)
    and on the 4-wide tree kernel (development bit 524288): the same ray count in every one of the 32 400 8x8 tiles, the
    frame's 733 152 639 rays, and EVERY pixel equal bit for bit (both kernels evaluate Texture::Noise with the same wave-balanced sums; only
    the oracle comparison needs the sinf tolerance, test_golden_fixture). List world as well: a third and fourth render of the same frame."""
    W, H, S = 1920, 1080, 128
    frames = {}
    for bvh in (True, False):
        hs = pthost.HostScene("perlin_spheres", W, H, samples=S, use_bvh=bvh, device=0)
        sc = hs.device_scene()
        for variant in (0, 524288, 2097152):   # the grid walk with parked walks (default), the tree, the grid walk with every call walked to its end
            sc.set_tuning(0, variant)
            out = np.zeros((H, W, 3), np.float32)
            rays = sc.update(ptgpu.PtParams(W, H, S, 10, 0, 1 if bvh else 0), hs.camera, 0, out)
            name = sc.last_kernel_choice()["name"]
            assert name.startswith("tree4<" if variant == 524288 else "grid<"), name
            frames[(bvh, variant)] = (rays, out, sc.tile_rays())
    assert frames[(True, 0)][0] == 733152639
    for bvh in (True, False):   # (a BVH world's ancestor-AABB gates refuse a handful of grazing hits a list world accepts -- bvh.rs:37-62: 8 rays of this frame)
        ref_rays, ref, ref_tiles = frames[(bvh, 0)]
        assert ref_tiles.shape == (135, 240) and int(ref_tiles.sum(dtype=np.uint64)) == ref_rays
        for variant in (524288, 2097152):
            rays, out, tiles = frames[(bvh, variant)]
            assert rays == ref_rays and np.array_equal(tiles, ref_tiles), (bvh, variant, rays, ref_rays, int((tiles != ref_tiles).sum()))
            assert np.array_equal(out, ref), "bvh %r variant %d: %s" % (bvh, variant, _report(ref, out))
    assert 0 < abs(frames[(False, 0)][0] - frames[(True, 0)][0]) < 100


@pytest.mark.parametrize("half,rlo,rhi,bvh", [(5.0, 0.1, 0.3, False), (6.0, 0.2, 0.2, True)])
def test_dense_random_cube_cell_grid_equals_the_tree_and_the_exact_scan(ptgpu, half, rlo, rhi, bvh):
    """The class the planner of round 6 newly admits (csrc/pt_prep.hip plan_cell_grid; tools/grid_ab.py): 10 000 spheres THROWN into a cube -- 85-89 % of the
    cells occupied, cells without a sphere beside chains of three records, Lambertian / Metal / Dielectric, a huge ground -- at 1200 x 800 x 16 on the grid
    walk (default), the grid walk without parked walks (2097152) and the 4-wide tree (524288): the same frame bit for bit and the same ray count in every
    8x8 tile; and at 240 x 160 x 4 against the exact scan (4 | 64: the reference's semantics; the BVH world: against the binary-tree kernel) as well. List world and a caller's BVH over it."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("grid_ab", os.path.join(ROOT, "tools", "grid_ab.py"))
    ab = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ab)
    ptgpu = ab.ptgpu   # (the tool's own binding of the same library: its SceneDesc / PtCamera classes are the ones its builders return)
    desc = ab.cloud(7, n=10000, half=half, rlo=rlo, rhi=rhi)
    if bvh:
        spec2 = importlib.util.spec_from_file_location("grid_soak", os.path.join(ROOT, "tools", "grid_soak.py"))
        gs = importlib.util.module_from_spec(spec2)
        spec2.loader.exec_module(gs)
        nodes, root = gs.bvh_over(desc.spheres, np.random.default_rng(5))
        tex = [(ptgpu.TEX_CONSTANT, (0.5, 0.5, 0.5), -1, -1, 0.0), (ptgpu.TEX_CONSTANT, (0.8, 0.3, 0.3), -1, -1, 0.0)]
        mats = [(ptgpu.MAT_LAMBERTIAN, (0, 0, 0), 0.0, 0), (ptgpu.MAT_LAMBERTIAN, (0, 0, 0), 0.0, 1), (ptgpu.MAT_METAL, (0.8, 0.8, 0.8), 0.1, -1), (ptgpu.MAT_DIELECTRIC, (0, 0, 0), 1.5, -1)]
        desc = ptgpu.SceneDesc(desc.spheres, desc.sphere_material, mats, tex, bvh_nodes=nodes, bvh_root=root)
    sc = ptgpu.Scene(desc, 0)
    other = (524288 | 2048) if bvh else (4 | 64)   # (a BVH world's flavour is always a tree kernel: its second reference is the BINARY tree without a grid)
    for W, H, S, variants in ((1200, 800, 16, (0, 2097152, 524288)), (240, 160, 4, (0, 2097152, 524288, other))):
        cam = ab.camera([1.6 * half, 0.8 * half + 1.0, 1.2 * half], 40.0, W / H)
        p = ptgpu.PtParams(W, H, S, 10, 0, 1 if bvh else 0)
        got = {}
        for v in variants:
            sc.set_tuning(0, v)
            out = np.zeros((H, W, 3), np.float32)
            rays = sc.update(p, cam, 0, out)
            name = sc.last_kernel_choice()["name"]
            assert name.startswith("grid<") == (v in (0, 2097152)), (v, name)
            try:
                tiles = sc.tile_rays()
            except ptgpu.PtError:   # (a frame in one launch -- few samples, or the exact scan -- does not count rays per work tile)
                tiles = None
            got[v] = (rays, out, tiles)
        ref_rays, ref, ref_tiles = got[0]
        assert (ref_tiles is not None) == (S >= 12) and (ref_tiles is None or int(ref_tiles.sum(dtype=np.uint64)) == ref_rays)
        for v in variants[1:]:
            rays, out, tiles = got[v]
            assert rays == ref_rays and np.array_equal(out, ref, equal_nan=True), (W, v, rays, ref_rays, _report(ref, out))
            if tiles is not None and ref_tiles is not None:
                assert np.array_equal(tiles, ref_tiles), (W, v, int((tiles != ref_tiles).sum()))
    sc.close()


# ---- full BASELINE sizes: sampled pixels + size-independent properties ----------------------------
def test_config3_full_size_sampled_against_oracle(ptgpu, pthost, oracle):
    """random_spheres 1200x800 64 spp (the metric's configuration): every 811th pixel vs the oracle."""
    W, H, S = 1200, 800, 64
    hs = pthost.HostScene("random_spheres", W, H, samples=S, device=0)
    out = np.zeros((H, W, 3), np.float32)
    rays = hs.device_scene().update(ptgpu.PtParams(W, H, S, 10, 0, 0), hs.camera, 0, out)
    px = np.arange(0, W * H, 811, dtype=np.uint32)
    ref = np.zeros((H, W, 3), np.float32)
    oracle.OracleScene("random_spheres", W, H).update(S, pixels=px, buffer=ref)
    a, b = out.reshape(-1, 3)[px], ref.reshape(-1, 3)[px]
    assert np.array_equal(a, b), _report(b, a)
    assert W * H * S <= rays <= W * H * S * 11                      # 1..max_depth+1 rays per sample
    assert np.isfinite(out).all() and out.min() >= 0.0


@pytest.mark.parametrize("preset,W,H,S,bvh", [("random_spheres", 1200, 800, 64, False),      # BASELINE config 3, EVERY pixel
                                              ("aras", 1280, 720, 16, False),                # config 2
                                              ("random", 1200, 800, 16, False)])             # motion blur on the MOVING kernels
def test_full_frames_at_baseline_size_are_bit_exact(ptgpu, pthost, oracle, preset, W, H, S, bvh):
    """The whole frame at the size the metric is quoted on, against the oracle on all host threads (~20 s of CPU)."""
    lib = oracle.lib(oracle.build_native(os.path.join(ROOT, "gpurun_out", "ora_native")))
    hs = pthost.HostScene(preset, W, H, samples=S, use_bvh=bvh, device=0)
    out = np.zeros((H, W, 3), np.float32)
    rays = hs.device_scene().update(ptgpu.PtParams(W, H, S, 10, 0, 1 if bvh else 0), hs.camera, 0, out)
    ref, ref_rays = oracle.OracleScene(preset, W, H, use_bvh=bvh, library=lib).update(S)
    assert rays == ref_rays, "ray_count %d vs oracle %d; %s" % (rays, ref_rays, _report(ref, out))
    assert np.array_equal(ref, out), _report(ref, out)


def test_config3_shard_union_equals_full_frame(ptgpu, pthost):
    """Disjoint row shards (the multi-GPU decomposition) reproduce the single-launch frame bit for bit,
    and their ray counts add up -- checked at the full 1200x800 size with 8 shards on one GPU."""
    import torch
    W, H, S, N = 1200, 800, 8, 8
    hs = pthost.HostScene("random_spheres", W, H, samples=S, device=0)
    sc = hs.device_scene()
    p = ptgpu.PtParams(W, H, S, 10, 0, 0)
    full = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    rc = torch.zeros(1, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    sc.update_device(p, hs.camera, 0, full.data_ptr(), rc.data_ptr(), stream)
    torch.cuda.synchronize()
    total_full = int(rc.item())
    frame = torch.zeros_like(full)
    total = 0
    for r in range(N):
        rows = ptgpu.shard_rows(H, r, N)
        shard = torch.zeros((rows, W, 3), dtype=torch.float32, device="cuda")
        sc.update_shard_device(p, hs.camera, 0, r, N, shard.data_ptr(), rc.data_ptr(), stream)
        torch.cuda.synchronize()
        total += int(rc.item())
        frame[r::N] = shard
    assert total == total_full
    assert torch.equal(frame, full)


def test_config4_shards_at_256spp_match_the_oracle_and_the_full_frame(ptgpu, pthost, oracle):
    """BASELINE config 4 (random_spheres 1200x800, 256 spp, rows y % 8 over 8 GPUs) with the eight shards rendered one
    after the other on one GPU into the layout ncclAllGather produces, de-interleaved by the C ABI's own kernel
    (pt_shard_unpack_all): every 1777th pixel equals the oracle at 256 spp bit for bit, the assembled frame equals the
    unsharded render, and the shards' ray counts add up to its count (scene.rs:90-118: disjoint pixels, summed count)."""
    import torch
    W, H, S, N = 1200, 800, 256, 8
    hs = pthost.HostScene("random_spheres", W, H, samples=S, device=0)
    sc = hs.device_scene()
    p = ptgpu.PtParams(W, H, S, 10, 0, 0)
    stream = torch.cuda.current_stream().cuda_stream
    rc = torch.zeros(1, dtype=torch.int64, device="cuda")
    prow = (H + N - 1) // N
    gathered = torch.zeros((N, prow, W, 3), dtype=torch.float32, device="cuda")
    total = 0
    for r in range(N):
        sc.update_shard_device(p, hs.camera, 0, r, N, gathered[r].data_ptr(), rc.data_ptr(), stream)
        torch.cuda.synchronize()
        total += int(rc.item())
    frame = torch.full((H, W, 3), -1.0, dtype=torch.float32, device="cuda")
    ptgpu.shard_unpack_all(gathered.data_ptr(), frame.data_ptr(), W, H, N, stream)
    full = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    sc.update_device(p, hs.camera, 0, full.data_ptr(), rc.data_ptr(), stream)
    torch.cuda.synchronize()
    assert total == int(rc.item())
    assert torch.equal(frame, full)
    px = np.arange(0, W * H, 1777, dtype=np.uint32)
    ref = np.zeros((H, W, 3), np.float32)
    oracle.OracleScene("random_spheres", W, H).update(S, pixels=px, buffer=ref)
    a, b = frame.cpu().numpy().reshape(-1, 3)[px], ref.reshape(-1, 3)[px]
    assert np.array_equal(a, b), _report(b, a)
    # the inverse layout kernel: packing rank 3's rows out of the frame returns its shard
    shard = torch.zeros((ptgpu.shard_rows(H, 3, N), W, 3), dtype=torch.float32, device="cuda")
    ptgpu.shard_pack(frame.data_ptr(), shard.data_ptr(), W, H, 3, N, stream)
    torch.cuda.synchronize()
    assert torch.equal(shard, gathered[3, :shard.shape[0]])


def test_sharded_update_through_the_rccl_communicator(ptgpu, pthost):
    """pt_render_sharded / pt_comm_gather_frame on a one-rank RCCL communicator (all a 1-GPU box can form): the calls
    go through ncclAllGather / ncclGather / ncclAllReduce and must reproduce pt_render_device, including the
    progressive blend of a second frame (the pack step reads the previous frame, scene.rs:114-116)."""
    import torch
    W, H, S = 301, 203, 4          # odd sizes: rows are not a multiple of anything
    hs = pthost.HostScene("random_spheres", W, H, samples=S, device=0)
    sc = hs.device_scene()
    p = ptgpu.PtParams(W, H, S, 10, 0, 0)
    stream = torch.cuda.current_stream().cuda_stream
    comm = ptgpu.Comm.create(ptgpu.Comm.unique_id(), 0, 1, 0)
    ref = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    got = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    rc_ref = torch.zeros(1, dtype=torch.int64, device="cuda")
    rc = torch.zeros(1, dtype=torch.int64, device="cuda")
    for f, root in ((0, -1), (1, 0)):            # all-gather form, then gather-to-root form
        sc.update_device(p, hs.camera, f, ref.data_ptr(), rc_ref.data_ptr(), stream)
        sc.update_sharded(comm, p, hs.camera, f, got.data_ptr(), rc.data_ptr(), root, stream)
        torch.cuda.synchronize()
        assert int(rc.item()) == int(rc_ref.item()) and torch.equal(got, ref)
    # the exchange step alone, on a shard the caller rendered itself
    shard = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    sc.update_shard_device(p, hs.camera, 0, 0, 1, shard.data_ptr(), rc.data_ptr(), stream)
    out = torch.zeros_like(shard)
    comm.gather_frame(W, H, shard.data_ptr(), out.data_ptr(), rc.data_ptr(), -1, stream)
    torch.cuda.synchronize()
    sc.update_device(p, hs.camera, 0, ref.zero_().data_ptr(), rc_ref.data_ptr(), stream)
    torch.cuda.synchronize()
    assert torch.equal(out, ref) and int(rc.item()) == int(rc_ref.item())
    with pytest.raises(ptgpu.PtError):
        sc.update_sharded(comm, p, hs.camera, 0, got.data_ptr(), rc.data_ptr(), 5, stream)   # root out of range
    comm.close()


def test_frames_rendered_apart_blend_like_sequential_updates(ptgpu, pthost):
    """bench.py's weak-scaling mode: each GPU renders one progressive frame into a zeroed buffer, the blend is
    replayed in frame order. On one GPU: frames 0..3 rendered apart and folded == four Scene::update calls."""
    import torch
    spec = importlib.util.spec_from_file_location("pathtrace_rs_amd_sharding_g", os.path.join(ROOT, "pathtrace-rs_amd", "sharding.py"))
    sharding = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sharding)
    W, H, S, N = 300, 200, 4, 4
    hs = pthost.HostScene("random_spheres", W, H, samples=S, device=0)
    sc, p = hs.device_scene(), ptgpu.PtParams(W, H, S, 10, 0, 0)
    rc = torch.zeros(1, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    seq = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    apart = torch.zeros((N, H, W, 3), dtype=torch.float32, device="cuda")
    for f in range(N):
        sc.update_device(p, hs.camera, f, seq.data_ptr(), rc.data_ptr(), stream)
        sc.update_device(p, hs.camera, f, apart[f].data_ptr(), rc.data_ptr(), stream)
    torch.cuda.synchronize()
    assert torch.equal(sharding.blend_frames(apart), seq)


def test_device_buffer_entry_point_matches_host_entry_point(ptgpu, pthost):
    import torch
    W, H, S = 200, 100, 4
    hs = pthost.HostScene("small", W, H, samples=S, device=0)
    sc = hs.device_scene()
    p = ptgpu.PtParams(W, H, S, 10, 0, 0)
    host = np.zeros((H, W, 3), np.float32)
    rays = sc.update(p, hs.camera, 0, host)
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        dev = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        rc = torch.zeros(1, dtype=torch.int64, device="cuda")
        sc.update_device(p, hs.camera, 0, dev.data_ptr(), rc.data_ptr(), s2.cuda_stream)
    s2.synchronize()
    assert int(rc.item()) == rays and np.array_equal(dev.cpu().numpy(), host)
    assert sc.last_kernel_ms() > 0.0
    grid, block, lds = sc.last_launch_info()
    assert block % 64 == 0 and grid >= 1 and lds > 0


def test_registered_host_buffer_renders_in_place(ptgpu, pthost):
    """pt_buffer_register: pt_render on a pinned + mapped caller buffer works in place over PCIe (no staging) and gives the
    same frames as the staged path, including the progressive blend that reads the previous frame from that buffer."""
    W, H, S = 320, 200, 8
    hs = pthost.HostScene("random_spheres", W, H, samples=S, device=0)
    sc, p = hs.device_scene(), ptgpu.PtParams(W, H, S, 10, 0, 0)
    staged = np.zeros((H, W, 3), np.float32)
    big = np.zeros((H + 7, W, 3), np.float32)          # the frame is a sub-range of the registered allocation
    inplace = big[3:3 + H]
    ptgpu.buffer_register(big)
    try:
        for f in range(3):
            ra = sc.update(p, hs.camera, f, staged)
            rb = sc.update(p, hs.camera, f, inplace)
            assert ra == rb and np.array_equal(staged, inplace), "frame %d" % f
        assert not big[:3].any() and not big[3 + H:].any()
    finally:
        ptgpu.buffer_unregister(big)
    with pytest.raises(ptgpu.PtError):
        ptgpu.buffer_unregister(big)                   # not registered any more


@pytest.mark.parametrize("preset,W,H,S", [("random_spheres", 320, 200, 16), ("cornell_smoke", 160, 120, 16), ("small", 96, 64, 4)])
def test_pageable_host_buffer_pipeline_against_the_oracle(ptgpu, pthost, oracle, preset, W, H, S):
    """pt_render on a pageable buffer (the call offline.rs:27-34 times): the kernels render into a pinned + mapped copy, the
    copy-in -- or the scan that finds the buffer all +0.0f and skips it -- runs under the measuring launch, helper threads copy
    back. Frame 0 on zeros, progressive frames on what the last one left, and frame 0 on a buffer full of values whose product
    with mix_prev = 0 is NOT +0 (-0.0, negatives, inf, NaN: scene.rs:114-116 multiplies whatever is there): all equal the oracle
    given the same starting buffer. Sizes with and without the two-launch work order; sphere and general-world kernels."""
    hs = pthost.HostScene(preset, W, H, samples=S, device=0)
    sc, p = hs.device_scene(), ptgpu.PtParams(W, H, S, 10, 0, 0)
    osc = oracle.OracleScene(preset, W, H)
    got, ref = np.zeros((H, W, 3), np.float32), np.zeros((H, W, 3), np.float32)
    for f in range(3):
        rays = sc.update(p, hs.camera, f, got)
        _, ref_rays = osc.update(S, frame_num=f, buffer=ref)
        assert rays == ref_rays and np.array_equal(got, ref), "frame %d: %s" % (f, _report(ref, got))
    assert sc.last_host_ms()[3] > 0.0                      # the pageable pipeline ran (and timed itself)
    rng = np.random.default_rng(3)
    start = np.zeros((H, W, 3), np.float32)
    flat = start.reshape(-1)
    for v in (-0.0, -3.5, np.inf, -np.inf, np.nan, 1e-45):
        flat[rng.integers(0, flat.size, 40)] = v
    for first in (start, np.full((H, W, 3), -0.0, np.float32)):
        got, ref = first.copy(), first.copy()
        rays = sc.update(p, hs.camera, 0, got)
        _, ref_rays = osc.update(S, frame_num=0, buffer=ref)
        assert rays == ref_rays and np.array_equal(got, ref, equal_nan=True), _report(ref, got)
        zero = ~np.isnan(ref) & (ref == 0)                                                   # (+0 vs -0 where the blend yields a zero)
        assert np.array_equal(np.signbit(got[zero]), np.signbit(ref[zero]))


def test_two_threads_render_two_scenes_through_the_host_entry_point(ptgpu, pthost):
    """include/ptgpu.h "Threading": distinct handles may be used from distinct threads. Two host threads call pt_render on
    pageable buffers at the same time (ctypes releases the GIL): the helper threads of the host-buffer pipeline serve one copy
    at a time, the frames are the ones each scene renders alone."""
    import threading
    jobs = [("random_spheres", 240, 160, 16), ("cornell_smoke", 160, 120, 16)]
    scenes = [pthost.HostScene(pr, W, H, samples=S, device=0) for pr, W, H, S in jobs]
    alone = []
    for hs, (pr, W, H, S) in zip(scenes, jobs):
        buf = np.zeros((H, W, 3), np.float32)
        alone.append((hs.device_scene().update(ptgpu.PtParams(W, H, S, 10, 0, 0), hs.camera, 0, buf), buf))
    results = [None, None]

    def work(i):
        hs, (pr, W, H, S) = scenes[i], jobs[i]
        out = []
        for _ in range(6):
            buf = np.zeros((H, W, 3), np.float32)
            out.append((hs.device_scene().update(ptgpu.PtParams(W, H, S, 10, 0, 0), hs.camera, 0, buf), buf))
        results[i] = out

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for i in range(2):
        for rays, buf in results[i]:
            assert rays == alone[i][0] and np.array_equal(buf, alone[i][1])


def test_communicators_from_a_device_list_and_side_by_side(ptgpu, pthost):
    """pt_comm_create_all with one device (all a 1-GPU box can list), two communicators alive at once, the gather-to-root form
    across two progressive frames, and the RCCL the library resolved at run time: the copy this process had already loaded
    (torch's), never a second one."""
    import torch
    ver, path = ptgpu.comm_runtime()
    assert ver >= 21000 and "librccl" in path
    torch_rccl = [l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l]
    assert len(set(torch_rccl)) == 1 and os.path.samefile(torch_rccl[0], path), (set(torch_rccl), path)    # ONE RCCL in the process
    W, H, S = 200, 120, 4
    hs = pthost.HostScene("aras", W, H, samples=S, device=0)
    sc, p = hs.device_scene(), ptgpu.PtParams(W, H, S, 10, 0, 0)
    stream = torch.cuda.current_stream().cuda_stream
    (a,) = ptgpu.Comm.create_all([0])
    b = ptgpu.Comm.create(ptgpu.Comm.unique_id(), 0, 1, 0)
    ref = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    rc_ref = torch.zeros(1, dtype=torch.int64, device="cuda")
    outs = {id(c): torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") for c in (a, b)}
    rc = torch.zeros(1, dtype=torch.int64, device="cuda")
    for f in range(3):                                   # frames alternate between the two communicators' forms
        sc.update_device(p, hs.camera, f, ref.data_ptr(), rc_ref.data_ptr(), stream)
        for c, root in ((a, 0), (b, -1)):
            sc.update_sharded(c, p, hs.camera, f, outs[id(c)].data_ptr(), rc.data_ptr(), root, stream)
            torch.cuda.synchronize()
            assert int(rc.item()) == int(rc_ref.item()) and torch.equal(outs[id(c)], ref), "frame %d root %d" % (f, root)
    # the grouped forms (all ranks of a clique from one thread: ncclGroupStart, every rank's posts, ncclGroupEnd, unpacks on the ranks'
    # streams) through the REAL library with the one rank a 1-GPU box has -- the mock behind tests/mock_rccl never blocks the host
    full = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    ref.zero_()                                          # (it holds frames 0..2 blended: start over)
    for f, root in ((0, -1), (1, 0), (2, -1)):
        sc.update_device(p, hs.camera, f, ref.data_ptr(), rc_ref.data_ptr(), stream)
        ptgpu.render_sharded_all([sc], [a], p, hs.camera, f, [full.data_ptr()], [rc.data_ptr()], root=root, streams=[stream])
        torch.cuda.synchronize()
        assert int(rc.item()) == int(rc_ref.item()) and torch.equal(full, ref), "render_sharded_all frame %d root %d" % (f, root)
    shard = ref.clone()                                  # one rank owns every row: its compact shard IS the frame
    full.zero_()
    rc.fill_(12345)
    ptgpu.gather_frame_all([a], W, H, [shard.data_ptr()], [full.data_ptr()], [rc.data_ptr()], root=0, streams=[stream])
    torch.cuda.synchronize()
    assert int(rc.item()) == 12345 and torch.equal(full, ref)
    with pytest.raises(ptgpu.PtError):                   # a NULL frame on the receiving rank is refused BEFORE anything is enqueued
        ptgpu.render_sharded_all([sc], [a], p, hs.camera, 3, [0], [rc.data_ptr()], root=0, streams=[stream])
    a.close()
    b.close()
    with pytest.raises(ptgpu.PtError):
        ptgpu.Comm.create_all([7])                       # no such device


@pytest.mark.parametrize("preset", ["random_spheres", "random", "aras"])
def test_closest_hit_queries_match_the_oracle(ptgpu, pthost, oracle, preset):
    """pt_closest_hit (csrc/pt_query.hip): the reference's closest-hit functions on explicit rays, device vs oracle, BIT for bit in
    t, entry, point and normal -- HitableList::ray_hit (hitable_list.rs:40-56), BVHNode::ray_hit over the scene's own tree
    (bvh.rs:37-62) and the three SpheresSoA variants (spheres_soa.rs:105-155 / 161-268 / 274-391), whose arithmetic
    differs from Sphere::ray_hit's (no `a`, no division, normal * (1 / r)): hence their own oracle functions and no render mode.
    Rays: camera-like unit rays, bounce-like rays from inside the cloud, and non-unit directions (the SoA forms assume |d| = 1 and
    are wrong for those -- but wrong the SAME way on both sides)."""
    import torch
    W, H = 200, 100
    rng = np.random.default_rng(11)
    n = 4096
    rays = np.zeros((n, 7), np.float32)
    rays[:, 0:3] = np.array([13, 2, 3], np.float32) + rng.normal(0, 0.05, (n, 3))
    target = rng.uniform(-6, 6, (n, 3)).astype(np.float32)
    target[:, 1] = rng.uniform(-0.5, 1.5, n)
    rays[n // 2:, 0:3] = rng.uniform(-5, 5, (n - n // 2, 3)) * np.array([1, 0.1, 1]) + np.array([0, 0.6, 0])   # from inside the cloud
    d = target - rays[:, 0:3]
    d /= np.sqrt((d * d).sum(axis=1, keepdims=True))
    d[::7] *= rng.uniform(0.3, 3.0, (len(d[::7]), 1))       # some non-unit directions
    rays[:, 3:6] = d
    rays[:, 6] = rng.uniform(0, 1, n)
    rays = rays.astype(np.float32)
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.zeros((n, 8), dtype=torch.float32, device="cuda")
    moving = preset == "random"
    modes = [("list", ptgpu.QUERY_LIST, False), ("bvh", ptgpu.QUERY_BVH, True)]
    if not moving:
        modes += [("soa1", ptgpu.QUERY_SOA_SCALAR, False), ("soa4", ptgpu.QUERY_SOA_SSE4_1, False), ("soa8", ptgpu.QUERY_SOA_AVX2, False)]
    for name, mode, bvh in modes:
        hs = pthost.HostScene(preset, W, H, samples=1, use_bvh=bvh, device=0)
        osc = oracle.OracleScene(preset, W, H, use_bvh=bvh)
        hs.device_scene().closest_hit(mode, n, d_rays.data_ptr(), d_hits.data_ptr())
        torch.cuda.synchronize()
        got = d_hits.cpu().numpy()
        entry = got[:, 1].view(np.uint32)
        n_hit = 0
        for i in range(n):
            o, dd, tm = rays[i, 0:3], rays[i, 3:6], float(rays[i, 6])
            ref = osc.world_ray_hit(o, dd, tm) if name in ("list", "bvh") else osc.soa_ray_hit({"soa1": 1, "soa4": 4, "soa8": 8}[name], o, dd)
            if ref is None:
                assert entry[i] == 0xffffffff, (preset, name, i, entry[i], got[i])
                continue
            n_hit += 1
            assert entry[i] == ref[1] and got[i, 0] == ref[0], (preset, name, i, entry[i], got[i, 0], ref[:2])
            assert np.array_equal(got[i, 2:5], ref[2]) and np.array_equal(got[i, 5:8], ref[3]), (preset, name, i, got[i], ref)
        assert n_hit > n // 2, (preset, name, n_hit)
    if moving:
        with pytest.raises(ptgpu.PtError):     # SpheresSoA::new panics on a MovingSphere (spheres_soa.rs:52)
            pthost.HostScene(preset, W, H, samples=1, device=0).device_scene().closest_hit(ptgpu.QUERY_SOA_SCALAR, n, d_rays.data_ptr(), d_hits.data_ptr())


MOCK_RCCL = os.path.join(ROOT, "tests", "mock_rccl")


def _mock_rccl_dir():
    """tests/mock_rccl/_build/librccl.so.1, (re)built with hipcc when the source is newer."""
    src, out = os.path.join(MOCK_RCCL, "mock_rccl.hip"), os.path.join(MOCK_RCCL, "_build", "librccl.so.1")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-shared", "-fPIC", "-I/opt/rocm/include", src, "-o", out])
    return os.path.dirname(out)


@pytest.mark.parametrize("ranks,root,height,mode,extra", [
    (2, -1, 51, "sharded_all", []),                  # all-gather, H not divisible by N, all ranks from one thread in ONE group
    (2, 1, 50, "sharded_all", ["--null-stale"]),     # gather to the last rank; the other passes NULL from frame 1 on (its slot keeps its rows)
    (3, 0, 50, "sharded", ["--null-stale"]),         # one rank per call, issue order alternating between frames
    (3, -1, 50, "gather", []),                       # caller-owned shards + pt_comm_gather_frame per rank
    (8, -1, 50, "sharded_all", []),                  # the 8-GPU node's geometry: ranks 2..7 own one row less than ranks 0, 1
    (8, 7, 50, "gather_all", []),
    (8, 0, 20, "sharded", []),
])
def test_sharded_frames_on_an_rccl_test_double(ptgpu, ranks, root, height, mode, extra):
    """The N > 1 paths of the C ABI (include/ptgpu.h "multi-GPU frames") EXECUTED on a 1-GPU box: a child process that never loads
    torch resolves tests/mock_rccl (a test double of the eleven RCCL entry points: N ranks on one device, collectives performed as
    stream-ordered device copies once every rank has posted them) and renders progressive frames over N communicators of
    pt_comm_create_all([0] * N). Every receiving rank's frame equals the unsharded pt_render_device frame bit for bit, every rank's
    ray count equals the unsharded count (scene.rs:90-93, 118-120), and no RCCL group is left open. This exercises OUR offsets,
    slots, root handling and grouping -- not RCCL itself: the real library at N > 1 is still unexecuted."""
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = _mock_rccl_dir() + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    # every rank's stream gets a hardware queue of its own: all ranks share ONE device here, and a stream parked inside a collective
    # must not hold back another rank's render behind it in the same queue (on a real node every rank has its own device)
    env["GPU_MAX_HW_QUEUES"] = "16"
    cmd = [sys.executable, os.path.join(MOCK_RCCL, "run_sharded.py"), "--ranks", str(ranks), "--root", str(root), "--height", str(height), "--mode", mode,
           "--frames", "3"] + extra
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["ok"] and "mock_rccl" in res["rccl_path"] and res["rccl_version"] % 10000 == 9900, res   # the double, not the real RCCL
    receiving = ranks if root < 0 else 1
    assert res["checks"] == 3 * (ranks + receiving), res


def test_cli_offline_render_matches_reference_harness(pthost, oracle, tmp_path):
    """offline.rs:16-60 through the C++ host CLI: banner, `{:.2}secs {}rays {:.2}Mrays/s`, and an output.png whose
    pixels are the oracle's frame through linear_to_srgb + vertical flip (math.rs:36-48, offline.rs:43-51)."""
    import re
    import struct
    import subprocess
    import zlib
    exe = os.path.join(ROOT, "pathtrace-rs_amd", "_build", "pathtrace")
    out_png = str(tmp_path / "out.png")
    W, H, S = 200, 100, 4
    r = subprocess.run([exe, "-O", "-P", "small", "-W", str(W), "-H", str(H), "-S", str(S), "--output", out_png],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[0] == "generating 'small' preset at 200x100 with 4 samples per pixel"      # presets.rs:19-22
    m = re.fullmatch(r"(\d+\.\d\d)secs (\d+)rays (\d+\.\d\d)Mrays/s", lines[1])               # offline.rs:36-41
    ref, ref_rays = oracle.OracleScene("small", W, H).update(S)
    assert m and int(m.group(2)) == ref_rays
    data = open(out_png, "rb").read()
    pos, idat = 8, b""
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        if tag == b"IDAT":
            idat += data[pos + 8:pos + 8 + n]
        pos += 12 + n
    img = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(H, 1 + 3 * W)[:, 1:].reshape(H, W, 3)
    want = np.zeros((H, W, 3), np.uint8)
    oracle.lib().ora_frame_to_srgb8(ref.ctypes.data, W, H, want.ctypes.data)
    assert np.array_equal(img, want)


def test_rust_expectations_are_what_the_product_renders(tmp_path):
    """tests/golden/rust_expectations.json is what the pin kit (tools/pin_against_rust.py) holds a real `cargo run --release` of the
    reference to; it was generated from the ORACLE. Here the PRODUCT renders every case through its CLI (same flags, offline.rs:16-60)
    and must print the same ray count and write the same pixels -- decoded with the kit's own PNG reader, so that is exercised too.
    Cases whose colour passes through f32::sin / ln on the host side of the oracle (`libm_sensitive`) may differ in a last bit of a
    byte; everything else is hash for hash."""
    import hashlib
    import re
    spec = importlib.util.spec_from_file_location("pin_against_rust", os.path.join(ROOT, "tools", "pin_against_rust.py"))
    kit = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kit)
    exp = json.load(open(os.path.join(GOLDEN, "rust_expectations.json")))
    exe = os.path.join(ROOT, "pathtrace-rs_amd", "_build", "pathtrace")
    assert len(exp["cases"]) >= 10
    for c in exp["cases"]:
        out_png = str(tmp_path / "out.png")
        r = subprocess.run([exe] + c["args"] + ["--output", out_png], capture_output=True, text=True)
        assert r.returncode == 0, (c["args"], r.stderr)
        m = re.search(r"(\d+)rays", r.stdout)
        assert m and int(m.group(1)) == c["rays"], (c["args"], r.stdout, c["rays"])
        got = kit.png_rgb8(out_png)
        if not c["libm_sensitive"]:
            assert hashlib.sha256(got).hexdigest() == c["rgb8_sha256"], c["args"]
        else:
            assert len(got) == 3 * int(c["args"][c["args"].index("-W") + 1]) * int(c["args"][c["args"].index("-H") + 1])


def test_random_seed_mode_is_deterministic_per_base_and_differs_from_fixed(ptgpu, pthost):
    W, H, S = 64, 32, 2
    hs = pthost.HostScene("small", W, H, samples=S, device=0)
    sc = hs.device_scene()
    fixed = np.zeros((H, W, 3), np.float32)
    sc.update(ptgpu.PtParams(W, H, S, 10, 0, 0), hs.camera, 0, fixed)
    p = ptgpu.PtParams(W, H, S, 10, 1, 0)
    sc.set_seed_base(1234)
    a, b, c = (np.zeros((H, W, 3), np.float32) for _ in range(3))
    sc.update(p, hs.camera, 0, a)
    sc.update(p, hs.camera, 0, b)
    sc.set_seed_base(99)
    sc.update(p, hs.camera, 0, c)
    assert np.array_equal(a, b) and not np.array_equal(a, c) and not np.array_equal(a, fixed)


def test_error_reporting_on_device(ptgpu, pthost):
    hs = pthost.HostScene("small", 32, 16, device=0)          # no BVH built
    sc = hs.device_scene()
    buf = np.zeros((16, 32, 3), np.float32)
    with pytest.raises(ptgpu.PtError) as e:
        sc.update(ptgpu.PtParams(32, 16, 1, 10, 0, 1), hs.camera, 0, buf)
    assert e.value.code == ptgpu.PT_ERR_UNSUPPORTED and "BVH" in str(e.value)
    with pytest.raises(ptgpu.PtError) as e:
        sc.update(ptgpu.PtParams(32, 16, 0, 10, 0, 0), hs.camera, 0, buf)
    assert e.value.code == ptgpu.PT_ERR_INVALID_ARG
    # the sphere kernels pack (depth, sample) into one register: beyond these limits the call is refused, not wrong
    for bad in (ptgpu.PtParams(32, 16, 1, 4096, 0, 0), ptgpu.PtParams(32, 16, 1 << 20, 10, 0, 0)):
        with pytest.raises(ptgpu.PtError) as e:
            sc.update(bad, hs.camera, 0, buf)
        assert e.value.code == ptgpu.PT_ERR_UNSUPPORTED
    assert sc.update(ptgpu.PtParams(32, 16, 1, 4095, 0, 0), hs.camera, 0, buf) > 0
    # every kernel keeps a lane's pixel in one register: frames of 65536 pixels or more along an axis are refused
    wide = np.zeros((1, 65536, 3), np.float32)
    for scene in (sc, pthost.HostScene("cornell", 32, 16, device=0).device_scene()):
        with pytest.raises(ptgpu.PtError) as e:
            scene.update(ptgpu.PtParams(65536, 1, 1, 10, 0, 0), hs.camera, 0, wide)
        assert e.value.code == ptgpu.PT_ERR_UNSUPPORTED


# ---- general worlds: MovingSphere, Rect, Cuboid, Instance, ConstantMedium ----------------
def _world_render(ptgpu, pthost, preset, W, H, S, bvh, depth=10, frame=0, prev=None):
    hs = pthost.HostScene(preset, W, H, samples=S, use_bvh=bvh, device=0)
    out = np.zeros((H, W, 3), np.float32) if prev is None else prev.copy()
    rays = hs.device_scene().update(ptgpu.PtParams(W, H, S, depth, 0, 1 if bvh else 0), hs.camera, frame, out)
    return hs, out, rays


@pytest.mark.parametrize("preset", ["random", "simple_light", "cornell", "cornell_smoke", "smallpt"])
@pytest.mark.parametrize("bvh", [False, True])
def test_world_presets_match_the_oracle(ptgpu, pthost, oracle, preset, bvh):
    """Whole small frames: ray_count and every float identical to the oracle, in HitableList order and through the
    reference's BVH (which, as upstream, clips instances to a point box and moving spheres to t = 0)."""
    W, H, S = 200, 120, 8
    hs, out, rays = _world_render(ptgpu, pthost, preset, W, H, S, bvh)
    assert hs.is_world == (preset != "smallpt")
    ref, ref_rays = oracle.OracleScene(preset, W, H, use_bvh=bvh).update(S)
    assert rays == ref_rays, "ray_count %d vs oracle %d; %s" % (rays, ref_rays, _report(ref, out))
    if preset == "simple_light":      # colour passes through sinf (noise texture); control flow does not
        np.testing.assert_allclose(out, ref, rtol=0, atol=NOISE_ATOL)
    else:
        assert np.array_equal(ref, out), _report(ref, out)


def test_empty_world_renders_the_sky(ptgpu, pthost, oracle):
    """`final` (presets.rs:40-71) has no hitables: one ray per sample, the gradient of scene.rs:40-47, bit for bit."""
    W, H, S = 160, 100, 3
    hs, out, rays = _world_render(ptgpu, pthost, "final", W, H, S, False)
    ref, ref_rays = oracle.OracleScene("final", W, H).update(S)
    assert rays == ref_rays == W * H * S and np.array_equal(ref, out), _report(ref, out)
    with pytest.raises(ptgpu.PtError):      # -B on an empty list: the reference panics, the ABI reports UNSUPPORTED
        hs.device_scene().update(ptgpu.PtParams(W, H, S, 10, 0, 1), hs.camera, 0, out)


@pytest.mark.parametrize("preset,W,H,S,bvh", [("cornell_smoke", 1200, 800, 16, False), ("random", 1200, 800, 8, False),
                                              ("cornell", 1280, 720, 16, True)])
def test_world_full_size_sampled_against_oracle(ptgpu, pthost, oracle, preset, W, H, S, bvh):
    hs, out, rays = _world_render(ptgpu, pthost, preset, W, H, S, bvh)
    px = np.arange(0, W * H, 1013, dtype=np.uint32)
    ref = np.zeros((H, W, 3), np.float32)
    oracle.OracleScene(preset, W, H, use_bvh=bvh).update(S, pixels=px, buffer=ref)
    a, b = out.reshape(-1, 3)[px], ref.reshape(-1, 3)[px]
    assert np.array_equal(a, b), _report(b, a)
    assert W * H * S <= rays <= W * H * S * 11 and np.isfinite(out).all()


@pytest.mark.parametrize("depth", [0, 1, 50])
def test_world_max_depth_edges(ptgpu, pthost, oracle, depth):
    W, H, S = 96, 64, 4
    _, out, rays = _world_render(ptgpu, pthost, "cornell_smoke", W, H, S, False, depth=depth)
    ref, ref_rays = oracle.OracleScene("cornell_smoke", W, H).update(S, max_depth=depth)
    assert rays == ref_rays and np.array_equal(ref, out), _report(ref, out)


def test_world_progressive_frames_and_shards(ptgpu, pthost, oracle):
    import torch
    W, H, S, N = 120, 80, 4, 3
    osc = oracle.OracleScene("cornell_smoke", W, H)
    hs = pthost.HostScene("cornell_smoke", W, H, samples=S, device=0)
    sc, p = hs.device_scene(), ptgpu.PtParams(W, H, S, 10, 0, 0)
    out, ref = np.zeros((H, W, 3), np.float32), np.zeros((H, W, 3), np.float32)
    for frame in range(3):            # scene.rs:86-87,113-116 blend; seeds move with frame_num
        _, rr = osc.update(S, frame_num=frame, buffer=ref)
        assert sc.update(p, hs.camera, frame, out) == rr and np.array_equal(ref, out), frame
    frame0 = np.zeros((H, W, 3), np.float32)
    total_full = sc.update(p, hs.camera, 0, frame0)
    rc = torch.zeros(1, dtype=torch.int64, device="cuda")
    stitched, total = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"), 0
    for r in range(N):                # the multi-GPU decomposition: rows y % N == r
        shard = torch.zeros((ptgpu.shard_rows(H, r, N), W, 3), dtype=torch.float32, device="cuda")
        sc.update_shard_device(p, hs.camera, 0, r, N, shard.data_ptr(), rc.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        total += int(rc.item())
        stitched[r::N] = shard
    assert total == total_full and np.array_equal(stitched.cpu().numpy(), frame0)


def _random_world(oracle, seed, n, kinds, W, H, moving_times=((0.0, 1.0),), media=True, instances=True, sky=None):
    """A seeded world mixing the Hitable arms in `kinds` (0 Sphere, 1 MovingSphere, 2-4 Rect, 5 Cuboid), some inside an
    Instance (arbitrary invertible affine map, inverse supplied as Instance::new would store it) and/or a
    ConstantMedium, with every material kind. Returned as the flat description both sides consume."""
    rng = np.random.default_rng(seed)
    tex = [[0, *rng.uniform(0.1, 0.9, 3), -1, -1, 0] for _ in range(5)]
    tex.append([1, 0, 0, 0, 0, 1, 0])                                   # checker(constant, constant)
    tex.append([0, 6.0, 5.0, 4.0, -1, -1, 0])                           # emitter colour
    mats = [[0, 0, 0, 0, 0, t] for t in range(6)]                       # lambertians (incl. the checker)
    mats += [[1, *rng.uniform(0.5, 1.0, 3), f, -1] for f in (0.0, 0.3)]  # metals
    mats += [[2, 0, 0, 0, 1.5, -1], [3, 0, 0, 0, 0, 6]]                  # dielectric, diffuse light
    n_surface = len(mats)
    rec = np.zeros((n, 16), np.uint32)
    pf = rec[:, 6:].view(np.float32)
    xfs = []
    for i in range(n):
        k = int(rng.choice(kinds))
        c = rng.uniform(-4, 4, 3).astype(np.float32)
        rec[i, 0], rec[i, 1], rec[i, 3], rec[i, 4] = k, int(rng.integers(0, n_surface)), 0xffffffff, 0xffffffff
        if k == 0:
            pf[i, :4] = [*c, rng.uniform(0.2, 0.9) * (1 if rng.random() > 0.1 else -1)]
        elif k == 1:
            t0, t1 = moving_times[int(rng.integers(0, len(moving_times)))]
            pf[i, :9] = [*c, *rng.uniform(-0.6, 0.6, 3), rng.uniform(0.2, 0.7), t0, np.float32(1.0) / (np.float32(t1) - np.float32(t0))]
        elif k in (2, 3, 4):
            a0, b0 = rng.uniform(-4, 2, 2)
            pf[i, :5] = [a0, a0 + rng.uniform(0.5, 3), b0, b0 + rng.uniform(0.5, 3), rng.uniform(-4, 4)]
            rec[i, 2] = int(rng.integers(0, 2))
        else:
            pf[i, :6] = [*c, *(c + rng.uniform(0.4, 2.0, 3).astype(np.float32))]
        if instances and k >= 2 and rng.random() < 0.5:
            m = np.linalg.qr(rng.normal(size=(3, 3)))[0] * rng.uniform(0.7, 1.4, 3)   # rotation x non-uniform scale
            t = rng.uniform(-1, 1, 3)
            m32, t32 = m.astype(np.float32), t.astype(np.float32)
            inv = np.linalg.inv(m32.astype(np.float64))
            it = -(inv @ t32.astype(np.float64))
            xfs.append(np.concatenate([m32.T.reshape(-1), t32, inv.astype(np.float32).T.reshape(-1), it.astype(np.float32)]))  # columns
            rec[i, 3] = len(xfs) - 1
        if media and k in (0, 5) and rng.random() < 0.35:
            mats.append([4, 0, 0, 0, 0, int(rng.integers(0, 5))])        # Isotropic phase function (rows at the end)
            rec[i, 4] = len(mats) - 1
            rec[i, 5:6].view(np.float32)[0] = rng.uniform(0.1, 0.8)
    cam = np.zeros(24, np.float32)
    L = oracle.lib()
    lf, la, up = (np.asarray(v, np.float32) for v in ([9, 4, 11], [0, 0, 0], [0, 1, 0]))
    L.ora_camera_new(lf.ctypes.data, la.ctypes.data, up.ctypes.data, 40.0, W / H, 0.1, 14.0, 0.0, 1.0, cam.ctypes.data)
    return dict(hitables=rec, transforms=np.array(xfs, np.float32).reshape(-1, 24), materials=np.array(mats, np.float32),
                textures=np.array(tex, np.float32), camera=cam, sky=sky)


def _render_world_both(ptgpu, oracle, w, W, H, S, bvh, variant=0, depth=10, frame=0):
    images = w.get("images", ())
    osc = oracle.OracleScene.from_world(w["hitables"], w["transforms"], w["materials"], w["textures"], w["camera"], W, H,
                                        sky=w["sky"], use_bvh=bvh, images=images)
    ex = osc.export()
    assert ex["hitables"].tobytes() == w["hitables"].tobytes() and ex["transforms"].tobytes() == w["transforms"].tobytes()
    sc = ptgpu.Scene(oracle.to_ptgpu_world_desc(ptgpu, ex, images=images), 0)
    if variant:
        sc.set_tuning(0, variant)
    out = np.zeros((H, W, 3), np.float32)
    rays = sc.update(ptgpu.PtParams(W, H, S, depth, 0, 1 if bvh else 0), ptgpu.PtCamera.from_floats(ex["camera"]), frame, out)
    sc.close()
    ref, ref_rays = osc.update(S, max_depth=depth, frame_num=frame)
    return out, rays, ref, ref_rays


@pytest.mark.parametrize("seed,n,kinds,bvh", [
    (1, 24, (0, 1, 2, 3, 4, 5), False),     # every arm, list order with media drawing from the pixel's RNG
    (2, 24, (0, 1, 2, 3, 4, 5), True),      # the same through BVHNode::ray_hit (point boxes for instances and all)
    (3, 7, (2, 3, 4, 5), False),            # flat shapes only
    (4, 60, (5,), True),                    # many instanced / smoky cuboids
    (5, 1, (5,), True),                     # bvh.rs:73-79: one hitable, lhs == rhs (a medium then draws twice)
    (6, 2, (0, 5), True),                   # bvh.rs:80-88: two hitables, one node
])
def test_random_general_worlds_match_the_oracle(ptgpu, oracle, seed, n, kinds, bvh):
    W, H, S = 120, 80, 4
    w = _random_world(oracle, seed, n, kinds, W, H, sky=(0.5, 0.6, 0.8) if seed % 2 else None)
    out, rays, ref, ref_rays = _render_world_both(ptgpu, oracle, w, W, H, S, bvh)
    assert rays == ref_rays, "ray_count %d vs oracle %d; %s" % (rays, ref_rays, _report(ref, out))
    assert np.array_equal(ref, out), _report(ref, out)


@pytest.mark.parametrize("seed,n,kinds,bvh,sky,depth,noisy_light", [
    (11, 16, (0, 2, 3, 4, 5), False, None, 10, False),             # gradient sky: every path ends lit -> every parked point gets its colour
    (12, 16, (0, 2, 3, 4, 5), True, (0.0, 0.0, 0.0), 10, False),   # black sky: nearly every path ends dark -> no colours, no fold
    (13, 24, (0, 1, 2, 3, 4, 5), False, (0.5, 0.6, 0.8), 10, True),   # + a DiffuseLight with a Noise texture (emitted: evaluated where it is hit)
    (14, 12, (0, 5), True, (0.0, 0.0, 0.0), 64, True),             # the deepest stack the parked form takes (one bit per level)
    (15, 12, (0, 5), False, None, 65, False),                      # one level more: the kernel without it
])
def test_noise_colours_formed_at_the_end_of_a_path_equal_those_formed_at_the_hit(ptgpu, oracle, seed, n, kinds, bvh, sky, depth, noisy_light):
    """General worlds with Noise textures (texture.rs:86-88 over perlin.rs:54-111) on Lambertian surfaces, behind a Checker, in
    Isotropic media and on a light: by default a scatter parks the hit point and the colour is formed when the path ends lit
    (csrc/pt_world.h LAZY); tuning bit 131072 forms it at the hit. Both must give the same bits and the oracle's ray count; the
    colours agree with the oracle within the sinf tolerance."""
    W, H, S = 120, 80, 6
    w = _random_world(oracle, seed, n, kinds, W, H, sky=sky)
    tex = w["textures"].copy()
    tex[1] = [2, 0, 0, 0, -1, -1, 4.0]       # Noise, two scales
    tex[3] = [2, 0, 0, 0, -1, -1, 0.7]
    tex[5] = [1, 0, 0, 0, 0, 1, 0]           # Checker(Constant, Noise)
    if noisy_light:
        tex[6] = [2, 0, 0, 0, -1, -1, 2.0]
    rec = w["hitables"]
    if seed in (11, 12):     # a Noise-textured floor 20 000 below, 200 000 wide: hit points beyond the reach within which a colour may wait (pt_args.h)
        floor = np.zeros((1, 16), np.uint32)
        floor[0, 0], floor[0, 1], floor[0, 3], floor[0, 4] = 3, 1, 0xffffffff, 0xffffffff       # Rect XZ, the Lambertian of texture 1
        floor[0, 6:11] = np.array([-1e5, 1e5, -1e5, 1e5, -2e4], np.float32).view(np.uint32)
        rec = np.concatenate([rec, floor])
    w = dict(w, textures=tex, hitables=rec)
    osc = oracle.OracleScene.from_world(w["hitables"], w["transforms"], w["materials"], w["textures"], w["camera"], W, H, sky=w["sky"], use_bvh=bvh)
    ex = osc.export()
    ref, ref_rays = osc.update(S, max_depth=depth)
    frames = {}
    for variant in (0, 131072):
        sc = ptgpu.Scene(oracle.to_ptgpu_world_desc(ptgpu, ex), 0)
        sc.set_tuning(0, variant)
        out = np.zeros((H, W, 3), np.float32)
        rays = sc.update(ptgpu.PtParams(W, H, S, depth, 0, 1 if bvh else 0), ptgpu.PtCamera.from_floats(ex["camera"]), 0, out)
        choice = sc.last_kernel_choice()
        sc.close()
        assert choice["family"] == 0 and choice["world_lazy"] == (1 if variant == 0 and depth <= 64 else 0), choice
        assert ("lazy" in choice["name"]) == bool(choice["world_lazy"])
        assert rays == ref_rays, "variant %d: ray_count %d vs oracle %d" % (variant, rays, ref_rays)
        np.testing.assert_allclose(out, ref, rtol=0, atol=NOISE_ATOL)
        frames[variant] = out
    assert np.array_equal(frames[0], frames[131072], equal_nan=True), _report(frames[131072], frames[0])   # (the far floor's grazing hits make NaNs in the reference too)


@pytest.mark.parametrize("bvh", [False, True])
@pytest.mark.parametrize("where", ["albedo", "texture", "both"])
def test_a_non_finite_colour_keeps_the_fold_of_a_dark_path(ptgpu, oracle, bvh, where):
    """scene.rs:62-64 multiplies a path that ended in black by each of its attenuations: zero for finite colours (the general-world
    kernel then skips the loop), NaN as soon as one of them is infinite -- the kernel must notice (WArgs::atts_finite). The two places
    an infinite colour can sit are exercised apart: a Metal's albedo (pt_material.albedo), and a Constant texture under a Lambertian,
    which pt_scene_create folds into the material record where the kernel's colour() never looks at it again."""
    W, H, S = 96, 64, 4
    w = _random_world(oracle, 21, 14, (0, 2, 3, 4, 5), W, H, sky=(0.0, 0.0, 0.0))
    mats = w["materials"].copy()
    tex = w["textures"].copy()
    if where in ("albedo", "both"):
        mats[6, 1] = np.inf                   # the first Metal's red albedo
    if where in ("texture", "both"):
        tex[2, 2] = np.inf                    # a Constant texture's green, under a Lambertian
    w = dict(w, materials=mats, textures=tex)
    out, rays, ref, ref_rays = _render_world_both(ptgpu, oracle, w, W, H, S, bvh)
    assert rays == ref_rays
    assert np.isnan(ref).any(), "the fixture no longer sends a dark path over the infinite colour"
    assert np.array_equal(np.isnan(ref), np.isnan(out)) and np.array_equal(np.nan_to_num(ref, nan=-1.0), np.nan_to_num(out, nan=-1.0)), _report(ref, out)


def _random_graph_world(oracle, seed, W, H, n_top=7, max_depth=4, media=True, wild=False):
    """A world given as a SCENE GRAPH (include/ptgpu.h pt_node): leaf shapes of every arm under random nestings of HitableList,
    Instance (also Instance of Instance, Instance around a List) and ConstantMedium (around Instance levels around a shape, and
    itself inside Instances). Returned as the flat arrays both sides consume; the oracle nests them literally."""
    rng = np.random.default_rng(seed)
    w = _random_world(oracle, seed, 40, (0, 1, 2, 3, 4, 5), W, H, media=False, instances=False, sky=(0.4, 0.5, 0.7) if seed & 1 else None)
    mats = [list(r) for r in w["materials"]]
    xfs, nodes, children = [], [], []

    def transform():
        m = np.linalg.qr(rng.normal(size=(3, 3)))[0] * rng.uniform(0.8, 1.25, 3)
        t = rng.uniform(-0.7, 0.7, 3)
        m32, t32 = m.astype(np.float32), t.astype(np.float32)
        inv = np.linalg.inv(m32.astype(np.float64))
        it = -(inv @ t32.astype(np.float64))
        xfs.append(np.concatenate([m32.T.reshape(-1), t32, inv.astype(np.float32).T.reshape(-1), it.astype(np.float32)]))
        return len(xfs) - 1

    def add(kind, a, b, density=0.0):
        nodes.append([kind, a, b, int(np.float32(density).view(np.uint32))])
        return len(nodes) - 1

    leaves = list(rng.permutation(40))
    bvh_minmax, bvh_lr = [], []

    def leaf():   # (a shape may sit in the graph more than once: hitable.rs:12-21 holds references)
        return add(0, int(leaves.pop()) if leaves else int(rng.integers(0, 40)), 0)

    def subtree(depth):
        r = rng.random()
        if depth >= max_depth or len(leaves) < 6 or r < 0.3:
            return leaf()
        q = rng.random() if wild else 1.0
        if q < 0.2:                                        # ConstantMedium around ANYTHING: a List, an Instance of a List, another medium, a BVHNode
            mats.append([4, 0, 0, 0, 0, int(rng.integers(0, 5))])
            return add(3, len(mats) - 1, subtree(depth + 1), float(rng.uniform(0.05, 0.6)))
        if q < 0.35:                                       # BVHNode below the root: any box (both sides take it as given), children = two subtrees
            c, h = rng.uniform(-3, 3, 3), rng.uniform(1.0, 7.0, 3) * (1 if rng.random() < 0.8 else 50)
            bvh_minmax.append([*(c - h), *(c + h)])
            bvh_lr.append([subtree(depth + 1), subtree(depth + 1)])
            return add(4, len(bvh_lr) - 1, 0)
        if r < 0.55:                                       # Instance around anything (a List, another Instance, a medium)
            return add(2, transform(), subtree(depth + 1))
        if r < 0.8:                                        # HitableList inside whatever we are in
            kids = [subtree(depth + 1) for _ in range(int(rng.integers(2, 4)))]
            first = len(children)
            children.extend(kids)
            return add(1, first, len(kids))
        if not media:
            return leaf()
        b = leaf()                                          # ConstantMedium around Instance^k(shape), k = 0..2
        for _ in range(int(rng.integers(0, 3))):
            b = add(2, transform(), b)
        mats.append([4, 0, 0, 0, 0, int(rng.integers(0, 5))])
        return add(3, len(mats) - 1, b, float(rng.uniform(0.1, 0.8)))

    kids = [subtree(1) for _ in range(n_top)]
    first = len(children)
    children.extend(kids)
    root = add(1, first, len(kids))
    return dict(w, materials=np.array(mats, np.float32), transforms=np.array(xfs, np.float32).reshape(-1, 24), nodes=np.array(nodes, np.uint32),
                node_children=np.array(children, np.uint32), root_node=root, bvh_minmax=np.array(bvh_minmax, np.float32).reshape(-1, 6),
                bvh_children=np.array(bvh_lr, np.int32).reshape(-1, 2))


def _render_graph_both(ptgpu, oracle, g, W, H, S, depth=10, frame=0):
    osc = oracle.OracleScene.from_graph(g["hitables"], g["transforms"], g["materials"], g["textures"], g["camera"], W, H, g["nodes"], g["node_children"],
                                        g["root_node"], sky=g["sky"], bvh_minmax=g["bvh_minmax"], bvh_children=g["bvh_children"])
    ref, ref_rays = osc.update(S, max_depth=depth, frame_num=frame)
    materials = [(int(r[0]), r[1:4], r[4], int(r[5])) for r in g["materials"]]
    textures = [(int(r[0]), r[1:4], int(r[4]), int(r[5]), r[6]) for r in g["textures"]]
    desc = ptgpu.WorldDesc(g["hitables"], g["transforms"], materials, textures, sky=g["sky"], nodes=g["nodes"], node_children=g["node_children"],
                           root_node=g["root_node"], bvh_nodes=(g["bvh_minmax"], g["bvh_children"]) if len(g["bvh_minmax"]) else None)
    sc = ptgpu.Scene(desc, 0)
    out = np.zeros((H, W, 3), np.float32)
    rays = sc.update(ptgpu.PtParams(W, H, S, depth, 0, 0), ptgpu.PtCamera.from_floats(g["camera"]), frame, out)
    choice = sc.last_kernel_choice()
    sc.close()
    return out, rays, ref, ref_rays, choice


@pytest.mark.parametrize("bvh", [False, True])
def test_a_medium_around_a_list_of_shapes_flattens_to_a_group(ptgpu, oracle, bvh):
    """constant_medium.rs:32-43 with a HitableList as the boundary: the medium asks the list twice ((-MAX, MAX), then (t + 0.0001, MAX)), and the
    list answers as hitable_list.rs:40-56 does, children in order with the closest parameter so far as t_max. Since round 5 such a medium is a
    list entry of the general-world kernel (PT_HIT_MEDIUM_GROUP: the medium + its children behind it) instead of a reason to interpret the whole
    graph: shapes of every kind under Instances of their own, Instances between the medium and the list, Instances around the medium, two
    groups side by side; as a list world and -- the groups being single leaves of the caller's tree -- as a BVH world. Bit for bit."""
    W, H, S = 96, 64, 4
    g = _random_graph_world(oracle, 977, W, H, n_top=4, max_depth=2, wild=False)     # (for its 40 leaf shapes, transforms and tables)
    mats = [list(r) for r in g["materials"]]
    mats.append([4, 0, 0, 0, 0, 0])
    iso = len(mats) - 1
    rng = np.random.default_rng(5)
    xfs = []
    for _ in range(2):    # glam Affine3A (x_axis, y_axis, z_axis, translation) and its inverse, as Instance::new stores them (instance.rs:16-22)
        m = np.linalg.qr(rng.normal(size=(3, 3)))[0] * rng.uniform(0.8, 1.25, 3)
        t = rng.uniform(-0.7, 0.7, 3)
        m32, t32 = m.astype(np.float32), t.astype(np.float32)
        inv = np.linalg.inv(m32.astype(np.float64))
        xfs.append(np.concatenate([m32.T.reshape(-1), t32, inv.astype(np.float32).T.reshape(-1), (-(inv @ t32.astype(np.float64))).astype(np.float32)]))
    g = dict(g, transforms=np.array(xfs, np.float32).reshape(-1, 24))
    n_xf = 2
    nodes, children = [], []

    def add(kind, a, b, density=0.0):
        nodes.append([kind, a, b, int(np.float32(density).view(np.uint32))])
        return len(nodes) - 1

    def lst(kids):
        first = len(children)
        children.extend(kids)
        return add(1, first, len(kids))

    leaf = [add(0, i, 0) for i in range(12)]
    inner_a = lst([leaf[0], add(2, 0, leaf[1]), add(2, 1, add(2, 0, leaf[2])), leaf[3]])       # shapes, Instance(shape), Instance(Instance(shape))
    group_a = add(2, 1 % n_xf, add(3, iso, add(2, 0, inner_a), 0.45))                        # Instance(Medium(Instance(List)))
    group_b = add(3, iso, lst([leaf[4], leaf[5]]), 0.2)                                      # Medium(List(a, b)), no Instances
    tops = [leaf[6], group_a, leaf[7], group_b, add(2, 0, leaf[8])]
    if not bvh:
        root = lst(tops)
        gg = dict(g, materials=np.array(mats, np.float32), nodes=np.array(nodes, np.uint32), node_children=np.array(children, np.uint32), root_node=root,
                  bvh_minmax=np.zeros((0, 6), np.float32), bvh_children=np.zeros((0, 2), np.int32))
        out, rays, ref, ref_rays, choice = _render_graph_both(ptgpu, oracle, gg, W, H, S, depth=10)
    else:
        # the caller's tree over the five children of the root list: ((0 1) (2 (3 4))), boxes generous (both sides take them as given)
        box = [-60, -60, -60, 60, 60, 60]
        root_list = lst(tops)
        # oracle: the same tree as BVHNode NODES of the graph (rows hold node indices), rooted at the top node
        o_nodes, o_children = [list(r) for r in nodes], list(children)
        rows = []

        def bnode(l, r):
            rows.append([l, r])
            o_nodes.append([4, len(rows) - 1, 0, 0])
            return len(o_nodes) - 1

        top = bnode(bnode(tops[0], tops[1]), bnode(tops[2], bnode(tops[3], tops[4])))
        osc = oracle.OracleScene.from_graph(g["hitables"], g["transforms"], np.array(mats, np.float32), g["textures"], g["camera"], W, H, np.array(o_nodes, np.uint32),
                                            np.array(o_children, np.uint32), top, sky=g["sky"], bvh_minmax=np.array([box] * len(rows), np.float32),
                                            bvh_children=np.array(rows, np.int32))
        ref, ref_rays = osc.update(S, max_depth=10, frame_num=0)
        # product: the root LIST + bvh_nodes whose leaves index its children, rendered with use_bvh
        p_rows = [[~0, ~1], [~3, ~4], [~2, 1], [0, 2]]     # node 3 is the root
        materials = [(int(r[0]), r[1:4], r[4], int(r[5])) for r in np.array(mats, np.float32)]
        textures = [(int(r[0]), r[1:4], int(r[4]), int(r[5]), r[6]) for r in g["textures"]]
        desc = ptgpu.WorldDesc(g["hitables"], g["transforms"], materials, textures, sky=g["sky"], nodes=np.array(nodes, np.uint32), node_children=np.array(children, np.uint32),
                               root_node=root_list, bvh_nodes=(np.array([box] * 4, np.float32), np.array(p_rows, np.int32)), bvh_root=3)
        sc = ptgpu.Scene(desc, 0)
        out = np.zeros((H, W, 3), np.float32)
        rays = sc.update(ptgpu.PtParams(W, H, S, 10, 0, 1), ptgpu.PtCamera.from_floats(g["camera"]), 0, out)
        choice = sc.last_kernel_choice()
        sc.close()
    assert choice["world_graph"] == 0 and "chains" in choice["name"] and choice["ref_bvh"] == (1 if bvh else 0), choice
    assert rays == ref_rays, "ray_count %d vs oracle %d; %s" % (rays, ref_rays, _report(ref, out))
    assert np.array_equal(ref, out, equal_nan=True), _report(ref, out)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
def test_scene_graphs_that_do_not_flatten_are_interpreted(ptgpu, oracle, seed):
    """A ConstantMedium whose boundary is a HitableList or another medium (constant_medium.rs:32-43 asks the boundary twice; a medium
    in there draws from the pixel's RNG both times) and BVHNodes below the root (bvh.rs:37-62, hitable.rs:12-21) have no list form:
    the device walks such a graph as the reference recurses (csrc/pt_graph.h). Same ray count, same frame, bit for bit."""
    W, H, S = 96, 64, 3
    g = _random_graph_world(oracle, 900 + seed, W, H, n_top=int(3 + seed % 5), max_depth=int(3 + seed % 3), wild=True)
    kinds = set(int(k) for k in g["nodes"][:, 0])
    out, rays, ref, ref_rays, choice = _render_graph_both(ptgpu, oracle, g, W, H, S, depth=[10, 3, 25][seed % 3], frame=seed % 2)
    assert choice["family"] == 0 and ("graph" in choice["name"]) == bool(choice["world_graph"])
    assert rays == ref_rays, "ray_count %d vs oracle %d (node kinds %s, %s); %s" % (rays, ref_rays, kinds, choice["name"], _report(ref, out))
    assert np.array_equal(ref, out, equal_nan=True), _report(ref, out)


def test_interpreted_graph_with_noise_textures_progressive_frames_and_shards(ptgpu, oracle):
    """The interpreted walk next to the rest of the path: Noise textures on the graph's leaves (coloured where they are hit: the GRAPH
    instantiation has no lazy form), progressive frames 0..2 blended like scene.rs:113-116, and the multi-GPU decomposition (rows
    y % 3) of frame 0 stitched back together -- all equal to the oracle's literal recursion."""
    W, H, S = 96, 66, 3
    g = _random_graph_world(oracle, 931, W, H, n_top=5, max_depth=4, wild=True)
    tex = g["textures"].copy()
    tex[1] = [2, 0, 0, 0, -1, -1, 3.0]
    tex[5] = [1, 0, 0, 0, 0, 1, 0]
    g = dict(g, textures=tex)
    osc = oracle.OracleScene.from_graph(g["hitables"], g["transforms"], g["materials"], g["textures"], g["camera"], W, H, g["nodes"], g["node_children"],
                                        g["root_node"], sky=g["sky"], bvh_minmax=g["bvh_minmax"], bvh_children=g["bvh_children"])
    ex = osc.export()
    materials = [(int(r[0]), r[1:4], r[4], int(r[5])) for r in g["materials"]]
    textures = [(int(r[0]), r[1:4], int(r[4]), int(r[5]), r[6]) for r in g["textures"]]
    desc = ptgpu.WorldDesc(g["hitables"], g["transforms"], materials, textures, perlin=ex["perlin"], sky=g["sky"], nodes=g["nodes"], node_children=g["node_children"],
                           root_node=g["root_node"], bvh_nodes=(g["bvh_minmax"], g["bvh_children"]) if len(g["bvh_minmax"]) else None)
    sc = ptgpu.Scene(desc, 0)
    p, cam = ptgpu.PtParams(W, H, S, 10, 0, 0), ptgpu.PtCamera.from_floats(g["camera"])
    out, ref, total, ref_total = np.zeros((H, W, 3), np.float32), np.zeros((H, W, 3), np.float32), 0, 0
    for frame in range(3):
        total += sc.update(p, cam, frame, out)
        ref, n = osc.update(S, max_depth=10, frame_num=frame, buffer=ref)   # (blends into the buffer as Scene::update does)
        ref_total += n
        if frame == 0:
            frame0 = out.copy()
    assert sc.last_kernel_choice()["world_graph"] == 1 and total == ref_total
    np.testing.assert_allclose(out, ref, rtol=0, atol=NOISE_ATOL)
    # `-B` over an interpreted graph is refused by the render call itself, not only by pt_debug_select (the graph's BVHNode rows are
    # nodes of the graph, not a tree over the world) -- and the refusal leaves the scene usable
    with pytest.raises(ptgpu.PtError, match="use_bvh requested"):
        sc.update(ptgpu.PtParams(W, H, S, 10, 0, 1), cam, 0, np.zeros((H, W, 3), np.float32))
    import torch
    rc = torch.zeros(1, dtype=torch.int64, device="cuda")
    stitched = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    for r in range(3):
        shard = torch.zeros((ptgpu.shard_rows(H, r, 3), W, 3), dtype=torch.float32, device="cuda")
        sc.update_shard_device(p, cam, 0, r, 3, shard.data_ptr(), rc.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        stitched[r::3] = shard
    sc.close()
    assert np.array_equal(stitched.cpu().numpy(), frame0, equal_nan=True)


def test_the_interpreted_graphs_of_the_test_above_cover_every_nesting(ptgpu, oracle):
    """(what the seeds above contain: a medium around a List, around another medium, around a BVHNode; a BVHNode under an Instance)"""
    seen = set()
    for seed in range(1, 11):
        g = _random_graph_world(oracle, 900 + seed, 96, 64, n_top=int(3 + seed % 5), max_depth=int(3 + seed % 3), wild=True)
        nd, lr = g["nodes"], g["bvh_children"]
        for k, a, b, _ in nd:
            if k == 3:
                seen.add(("medium around", int(nd[b][0])))
            if k == 2:
                seen.add(("instance around", int(nd[b][0])))
            if k == 4:
                seen.update(("bvh over", int(nd[c][0])) for c in lr[a])
    assert {("medium around", 1), ("medium around", 3), ("medium around", 4), ("instance around", 4), ("instance around", 3), ("bvh over", 3)} <= seen, seen


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8])
def test_scene_graph_worlds_match_the_oracle(ptgpu, oracle, seed):
    """collision/hitable.rs:12-21 nests Hitables freely. The product flattens a scene graph into its list form (Lists
    concatenated, an Instance distributed over the List inside it, Instance levels around and inside a ConstantMedium as
    transform chains); the oracle builds the nesting literally and recurses as the reference does. Same frame, bit for bit."""
    W, H, S = 112, 80, 4
    g = _random_graph_world(oracle, 500 + seed, W, H, media=seed % 4 != 0)
    osc = oracle.OracleScene.from_graph(g["hitables"], g["transforms"], g["materials"], g["textures"], g["camera"], W, H, g["nodes"], g["node_children"],
                                        g["root_node"], sky=g["sky"])
    ref, ref_rays = osc.update(S, max_depth=10, frame_num=0)
    materials = [(int(r[0]), r[1:4], r[4], int(r[5])) for r in g["materials"]]
    textures = [(int(r[0]), r[1:4], int(r[4]), int(r[5]), r[6]) for r in g["textures"]]
    desc = ptgpu.WorldDesc(g["hitables"], g["transforms"], materials, textures, sky=g["sky"], nodes=g["nodes"], node_children=g["node_children"],
                           root_node=g["root_node"])
    sc = ptgpu.Scene(desc, 0)
    out = np.zeros((H, W, 3), np.float32)
    rays = sc.update(ptgpu.PtParams(W, H, S, 10, 0, 0), ptgpu.PtCamera.from_floats(g["camera"]), 0, out)
    sc.close()
    assert rays == ref_rays, "ray_count %d vs oracle %d; %s" % (rays, ref_rays, _report(ref, out))
    assert np.array_equal(ref, out, equal_nan=True), _report(ref, out)


def _random_sphere_world(oracle, seed, n, W, H, spread, rmax, extras=()):
    """Sphere-only world with every material kind, duplicates (exact t ties), negative radii (hollow glass) and the
    listed edge-case spheres appended: the specialised sphere kernels (MFMA prefilter / exact scan / internal tree)."""
    rng = np.random.default_rng(seed)
    tex = [[0, *rng.uniform(0.1, 0.9, 3), -1, -1, 0] for _ in range(4)] + [[1, 0, 0, 0, 0, 1, 0], [0, 5.0, 4.0, 3.0, -1, -1, 0]]
    mats = [[0, 0, 0, 0, 0, t] for t in range(5)] + [[1, *rng.uniform(0.5, 1, 3), f, -1] for f in (0.0, 0.4)]
    mats += [[2, 0, 0, 0, 1.5, -1], [2, 0, 0, 0, 2.4, -1], [3, 0, 0, 0, 0, 5]]
    sph = np.concatenate([rng.uniform(-spread, spread, (n, 3)), rng.uniform(0.05, rmax, (n, 1))], axis=1).astype(np.float32)
    sph[rng.random(n) < 0.08, 3] *= -1                      # presets.rs:265 style negative radius
    sph[n // 2] = sph[n // 3]                                # an exact duplicate: equal t, the lower list index must win
    if len(extras):
        sph = np.concatenate([sph, np.asarray(extras, np.float32)])
    rec = np.zeros((len(sph), 16), np.uint32)
    rec[:, 1] = rng.integers(0, len(mats), len(sph))
    rec[:, 3] = rec[:, 4] = 0xffffffff
    rec[:, 6:10] = sph.view(np.uint32)
    cam = np.zeros(24, np.float32)
    lf, la, up = (np.asarray(v, np.float32) for v in ([0.3 * spread, 0.4 * spread, 1.6 * spread], [0, 0, 0], [0, 1, 0]))
    oracle.lib().ora_camera_new(lf.ctypes.data, la.ctypes.data, up.ctypes.data, 45.0, W / H, 0.05, 1.5 * spread, 0.0, 1.0, cam.ctypes.data)
    return dict(hitables=rec, transforms=np.zeros((0, 24), np.float32), materials=np.array(mats, np.float32),
                textures=np.array(tex, np.float32), camera=cam, sky=None)


@pytest.mark.parametrize("seed,n,spread,rmax,extras", [
    (21, 300, 6.0, 0.5, ()),                                             # MFMA prefilter territory
    (22, 40, 3.0, 0.8, ()),                                              # two tiles
    (23, 12, 2.0, 0.9, ()),                                              # too few spheres for the prefilter: exact scan
    (24, 200, 6.0, 0.4, ([0, -500, 0, 498], [0, 0, 0, 30], [1e4, 0, 0, 0.5], [0.3, 0.4, 1.6, 0.01])),  # ground, a sphere
    #                      around the whole scene (camera inside), one far outside the f16 feature range, one tiny
    (25, 100, 4.0, 0.5, ([0, 0, 0, 0.0], [1, 1, 1, 1e-20])),             # r = 0 and r^2 underflowing to 0
    (26, 900, 12.0, 0.4, ()),                                            # > 768 spheres: list mode walks the internal tree
    (27, 900, 12.0, 0.4, ([0.5, 0.5, 2.0, 1e-7], [0.2, 0.1, 3.0, 2e-5], [-1, 0.3, 2.5, -3e-6])),   # the same with spheres too small
    #                      for the packed tree nodes' pad constant (kept out of the tree, tested for every ray)
])
@pytest.mark.parametrize("bvh", [False, True])
def test_random_sphere_worlds_match_the_oracle(ptgpu, oracle, seed, n, spread, rmax, extras, bvh):
    W, H, S = 128, 96, 4
    w = _random_sphere_world(oracle, seed, n, W, H, spread, rmax, extras)
    out, rays, ref, ref_rays = _render_world_both(ptgpu, oracle, w, W, H, S, bvh)
    assert rays == ref_rays, "ray_count %d vs oracle %d; %s" % (rays, ref_rays, _report(ref, out))
    assert np.array_equal(ref, out, equal_nan=True), _report(ref, out)


def _far_origin_world(oracle, seed, n, W, H, spread, rmax, kind, scale, moving=False):
    """A cloud of small spheres plus huge ones that send BOUNCE rays back at it from far away: the reference's f32
    discriminant (sphere.rs:33-38) has an error of ~1.3e-6 |o - c|^2, so from |o - c| = 2000 it accepts rays passing
    2 units outside a sphere of radius 0.2 -- whatever it does there the kernels must do too.
      kind 'enclosing': the camera and the cloud inside one sphere of radius -`scale`. NEGATIVE, so that its normals point
                        inwards (sphere.rs:42 divides by the signed radius) and it works as a concave mirror: with outward normals
                        a Metal absorbs every ray that reaches it from inside (material.rs:76) and a Lambertian scatters outwards.
                        Metal without fuzz (most seeds: rays from near the centre come back through the cloud, from `scale` away),
                        fuzzy metal or lambertian.
      kind 'offcentre': the same mirror with its centre 0.3 `scale` away from the cloud (chords return after several bounces)
      kind 'ground':    a ground sphere of radius `scale` under the cloud, camera 50 units up
      kind 'mirrors':   two metal spheres of radius `scale` facing each other across the cloud, 3 radii apart"""
    if moving:   # Sphere + MovingSphere entries within +-4 (the MOVING instantiations of the sphere kernels)
        w = _random_world(oracle, seed, n, (0, 1, 1), W, H, media=False, instances=False)
        metal, fuzzy, lambert, spread = 6, 7, seed % 5, 4.0   # rows of _random_world's material table
    else:
        w = _random_sphere_world(oracle, seed, n, W, H, spread, rmax)
        metal, fuzzy, lambert = 5, 6, seed % 4            # rows of _random_sphere_world's material table
    rec = w["hitables"]
    look_from, look_at = [0.3 * spread, 0.4 * spread, 1.6 * spread], [0, 0, 0]
    if kind == "enclosing":
        extras, mats = [[0.1 * spread, 0.0, -0.1 * spread, -scale]], [[metal, metal, metal, fuzzy, lambert][seed % 5]]
    elif kind == "offcentre":
        extras, mats = [[0.3 * scale, 0.05 * scale, -0.1 * scale, -scale]], [metal]
    elif kind == "ground":
        extras, mats = [[0.0, -scale - spread - 1.0, 0.0, scale]], [lambert if seed & 1 else metal]
        look_from = [0.8 * spread, 50.0, 2.0 * spread]
    else:
        extras = [[3.0 * scale, 0.0, 0.0, scale], [-3.0 * scale, 0.5 * spread, 0.0, scale]]
        mats = [metal, metal]
    more = np.zeros((len(extras), 16), np.uint32)
    more[:, 1] = mats
    more[:, 3] = more[:, 4] = 0xffffffff
    more[:, 6:10] = np.asarray(extras, np.float32).view(np.uint32)
    cam = np.zeros(24, np.float32)
    lf, la, up = (np.asarray(v, np.float32) for v in (look_from, look_at, [0, 1, 0]))
    oracle.lib().ora_camera_new(lf.ctypes.data, la.ctypes.data, up.ctypes.data, 50.0, W / H, 0.05, float(np.linalg.norm(lf - la)), 0.0, 1.0,
                                cam.ctypes.data)
    return dict(w, hitables=np.concatenate([rec, more]), camera=cam)


def _check_all_list_paths_against_the_oracle(ptgpu, oracle, w, W, H, S, bvh, depth=10, more_variants=(), default_kernel=None):
    """Default kernel, no tile culling (1024), exact VALU scan (4 | 64) and -- list worlds the prefilter takes -- verify mode
    (8: dropped positives and culled winners must be 0), every one against the ORACLE."""
    osc = oracle.OracleScene.from_world(w["hitables"], w["transforms"], w["materials"], w["textures"], w["camera"], W, H, sky=w["sky"], use_bvh=bvh)
    ex = osc.export()
    ref, ref_rays = osc.update(S, max_depth=depth, frame_num=0)
    sc = ptgpu.Scene(oracle.to_ptgpu_world_desc(ptgpu, ex), 0)
    p, cam = ptgpu.PtParams(W, H, S, depth, 0, 1 if bvh else 0), ptgpu.PtCamera.from_floats(ex["camera"])
    bad = []
    for variant in (0, 1024, 4 | 64) + ((256,) if bvh else (8,)) + tuple(more_variants):
        sc.set_tuning(0, variant)
        if variant == 8:
            sc.debug_counters(reset=True)
        out = np.zeros((H, W, 3), np.float32)
        rays = sc.update(p, cam, 0, out)
        if variant == 0 and default_kernel is not None and not sc.last_kernel_choice()["name"].startswith(default_kernel):
            bad.append("default kernel is %s, expected %s..." % (sc.last_kernel_choice()["name"], default_kernel))
        if rays != ref_rays or not np.array_equal(ref, out, equal_nan=True):
            bad.append("variant %d: rays %d vs %d, %s" % (variant, rays, ref_rays, _report(ref, out)))
        if variant == 8:
            c = sc.debug_counters()
            if c["misses"] != 0 or (c["exact_positives"] == 0 and len(w["hitables"]) <= 700):   # (beyond 768 spheres: the tree kernel)
                bad.append("verify mode: %r" % (c,))
    sc.close()
    return bad


@pytest.mark.parametrize("kind,scale", [("enclosing", 2.0e2), ("enclosing", 2.0e3), ("enclosing", 2.0e4), ("enclosing", 2.0e5), ("enclosing", 3.0e6),
                                        ("offcentre", 3.0e2), ("offcentre", 1.0e4), ("ground", 1.0e4), ("ground", 1.0e5), ("mirrors", 1.0e3), ("mirrors", 3.0e4)])
@pytest.mark.parametrize("bvh", [False, True])
def test_far_ray_origins_keep_prefilter_and_tile_culling_exact(ptgpu, oracle, kind, scale, bvh):
    """Bounce origins far outside the cloud (off an enclosing sphere, a huge ground, two distant mirrors): the default MFMA
    kernel with tile culling, the same without culling, the exact scan and verify mode all equal the ORACLE bit for bit.
    The per-ray reach of lane_tile_mask and the far-ray path of make_ray_features are what this pins."""
    W, H, S = 128, 96, 4
    w = _far_origin_world(oracle, 31, 300, W, H, 6.0, 0.3, kind, scale)     # (seed 31: the enclosing sphere is a plain mirror)
    bad = _check_all_list_paths_against_the_oracle(ptgpu, oracle, w, W, H, S, bvh, depth=25 if kind == "offcentre" else 10)
    assert not bad, bad


@pytest.mark.parametrize("scale", [3.0e2, 3.0e4, 3.0e6])
@pytest.mark.parametrize("bvh", [False, True])
def test_far_ray_origins_with_moving_spheres(ptgpu, oracle, scale, bvh):
    """The same concave mirror around a world of Sphere + MovingSphere entries: the MOVING instantiations' swept prefilter
    bounds and swept tree boxes under far bounce origins."""
    W, H, S = 96, 64, 3
    w = _far_origin_world(oracle, 41, 200, W, H, 4.0, 0.7, "enclosing", scale, moving=True)
    bad = _check_all_list_paths_against_the_oracle(ptgpu, oracle, w, W, H, S, bvh)
    assert not bad, bad


def _as_dense_field(w, n, seed, pitch=0.5, radius=0.2):
    """Turns the first n entries (Sphere / MovingSphere rows) of a world from _far_origin_world into an even, dense field: a jittered
    side x side lattice of equal spheres in the plane y = radius, centred on the origin -- BASELINE config 5's layout, the kind of scene
    that gets a uniform cell grid (csrc/pt_grid.h; looser clouds keep the tree)."""
    rng = np.random.default_rng(seed)
    side = int(np.sqrt(n))
    assert side * side == n
    rec = w["hitables"]
    p = rec[:n, 6:16].view(np.float32)
    i = np.arange(n)
    p[:, 0] = pitch * ((i % side) - side / 2) + rng.uniform(0, 0.3, n).astype(np.float32)
    p[:, 1] = radius
    p[:, 2] = pitch * ((i // side) - side / 2) + rng.uniform(0, 0.3, n).astype(np.float32)
    moving = rec[:n, 0] == 1
    p[~moving, 3] = radius
    p[moving, 3:6] *= 0.1                      # (centre_delta: the sweeps stay within a cell or two)
    p[moving, 6] = radius
    return w


@pytest.mark.parametrize("kind,scale", [("enclosing", 1.0e2), ("enclosing", 2.0e3), ("enclosing", 3.0e5), ("offcentre", 3.0e2), ("ground", 1.0e4), ("mirrors", 1.0e3)])
@pytest.mark.parametrize("bvh", [False, True])
def test_cell_grid_walk_equals_the_oracle_from_near_far_and_beyond(ptgpu, oracle, kind, scale, bvh):
    """A dense field of 1 600 equal spheres -- the kind of scene that gets a uniform cell grid (csrc/pt_grid.h) -- with bounce origins inside
    the field, and so far away (off a concave mirror of radius 100 ... 300 000 around it, a huge ground, two distant mirrors) that the
    reference's f32 discriminant is coarser than the grid's registration: those rays walk the tree inside the grid kernel. The default
    kernel must BE the grid walk, and it, the 4-wide tree (524288), the exact scan and verify mode's counting twin (8 | 256) all equal the
    ORACLE bit for bit, list and BVH semantics."""
    W, H, S = 112, 80, 3
    w = _as_dense_field(_far_origin_world(oracle, 57, 1600, W, H, 9.0, 0.3, kind, scale), 1600, 57)
    bad = _check_all_list_paths_against_the_oracle(ptgpu, oracle, w, W, H, S, bvh, depth=25 if kind == "offcentre" else 10, more_variants=(524288, 8 | 256, 2097152), default_kernel="grid<")   # (2097152: no parked walks)
    assert not bad, bad


@pytest.mark.parametrize("offset,kernel", [((150.0, -40.0, 90.0), "grid<"), ((-260.0, 3.0, 310.0), "grid<"), ((2.0e3, 0.0, -1.0e3), "tree4<"), ((1.0e4, 5.0e3, 3.0e4), "tree4<"), ((1.0e5, 0.0, 0.0), "tree4<")])
@pytest.mark.parametrize("bvh", [False, True])
def test_dense_fields_away_from_the_origin(ptgpu, oracle, offset, kernel, bvh):
    """The whole scene -- field, mirror, camera -- translated away from the origin (round 5's advisor finding: every grid test sat on it). The walk
    forms its cell boundaries in f32 at the grid's coordinates, and the registrations are padded by h / 1000 for that rounding: within a few
    hundred cell sizes of the origin the field keeps its grid and the walk equals the oracle; farther out, where an ulp outgrows the pad, the
    planner hands the field to the tree (whose boxes are padded per ray) -- which equals the oracle as well."""
    W, H, S = 96, 64, 3
    w = _as_dense_field(_far_origin_world(oracle, 61, 1600, W, H, 9.0, 0.3, "enclosing", 1.0e2), 1600, 61)
    off = np.array(offset, np.float32)
    w["hitables"][:, 6:9] = (w["hitables"][:, 6:9].view(np.float32) + off).view(np.uint32)
    cam = np.array(w["camera"], np.float32)
    cam[0:3] += off          # origin
    cam[3:6] += off          # lower_left_corner (camera.rs:8-19: both are points)
    w["camera"] = cam
    bad = _check_all_list_paths_against_the_oracle(ptgpu, oracle, w, W, H, S, bvh, more_variants=(524288,), default_kernel=kernel)
    assert not bad, bad


@pytest.mark.parametrize("seed,scale", [(43, 3.0e2), (43, 3.0e5), (44, 1.0e2), (45, 3.0e3), (46, 3.0e2), (47, 1.0e4)])
@pytest.mark.parametrize("bvh", [False, True])
def test_cell_grid_walk_with_moving_spheres(ptgpu, oracle, seed, scale, bvh):
    """The MOVING flavour: a dense field of 1 296 Sphere + MovingSphere entries (cells hold a moving sphere wherever its sweep reaches; the
    discriminants use the centre at the ray's time), near and far bounce origins, against the oracle and the tree."""
    W, H, S = 96, 64, 3
    w = _as_dense_field(_far_origin_world(oracle, seed, 1296, W, H, 4.0, 0.7, "enclosing", scale, moving=True), 1296, seed)
    bad = _check_all_list_paths_against_the_oracle(ptgpu, oracle, w, W, H, S, bvh, more_variants=(524288,), default_kernel="grid<")
    assert not bad, bad


@pytest.mark.parametrize("first,count", [(40, 10), (285, 6), (5000, 60), (100000, 90)])
def test_cell_grid_soak_slice(ptgpu, first, count):
    """A slice of tools/grid_soak.py: seeded dense sphere fields (one layer, several, a packed cube, a long strip; radii within a band; a ground,
    big and degenerate spheres beside them; cameras inside, near, far, very far; every third one a BVH world) -- the default kernel, the grid
    walk for most of them, against the exact scan, every pixel and the ray count. Seeds 45 and 288 are the two that caught a real defect (a
    single cell along the thin axis of a layer thicker than the cell: the grid's box cut the spheres' tops off). From seed 100 000 on, two in five
    are RANDOM fields (filled cubes, thick layers): the planner of round 6 gives those a grid when 83 % of the cells hold a sphere, so the walk
    meets empty cells and long chains of records."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("grid_soak", os.path.join(ROOT, "tools", "grid_soak.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    grids, bad = mod.run(first, count)
    assert bad == 0 and grids >= count // 3, (grids, bad)


@pytest.mark.parametrize("first", [90000, 90040])
def test_fuzz_slice_of_far_origin_worlds(ptgpu, oracle, first):
    """40 seeded worlds per slice of the far-origin kind (tools/fuzz_worlds.py kind 3): cloud size, spread, radius range,
    the huge spheres' kind and scale (10^2.5 .. 10^6.5) all drawn from the seed; list and BVH, every kernel path vs the oracle."""
    W, H, S = 64, 48, 2
    bad = []
    for seed in range(first, first + 40):
        rng = np.random.default_rng(seed)
        n = int(rng.choice([40, 150, 400, 700, 900]))
        kind = ["enclosing", "enclosing", "offcentre", "ground", "mirrors"][int(rng.integers(0, 5))]
        scale = float(10.0 ** rng.uniform(2.5, 6.5))
        spread, rmax = float(rng.uniform(2, 15)), float(rng.uniform(0.1, 1.0))
        depth = int(rng.choice([2, 10, 10, 25]))
        w = _far_origin_world(oracle, seed, n, W, H, spread, rmax, kind, scale)
        for bvh in (False, True):
            for msg in _check_all_list_paths_against_the_oracle(ptgpu, oracle, w, W, H, S, bvh, depth=depth):
                bad.append((seed, kind, scale, bvh, msg))
    assert not bad, bad


@pytest.mark.parametrize("bvh", [False, True])
def test_unpackable_tree_falls_back_to_the_binary_tree(ptgpu, oracle, bvh):
    """A node whose pad constants do not fit the packed format (radii of 2e-5 inside a scene 2000 units wide: 6e-6 / r_min x
    |extent|^2 is beyond 2^17) makes pt_scene_create report the packed tree unusable; the kernels then walk the binary tree
    and the result is still the oracle's."""
    W, H, S = 96, 64, 2
    w = _random_sphere_world(oracle, 28, 900, W, H, 1000.0, 40.0, extras=([3.0, 2.0, 5.0, 2e-5], [-400.0, 100.0, 30.0, 3e-5]))
    osc = oracle.OracleScene.from_world(w["hitables"], w["transforms"], w["materials"], w["textures"], w["camera"], W, H, sky=w["sky"], use_bvh=bvh)
    ex = osc.export()
    sc = ptgpu.Scene(oracle.to_ptgpu_world_desc(ptgpu, ex), 0)
    nodes, usable = sc.debug_tree_packed()
    assert len(nodes) > 0 and not usable
    out = np.zeros((H, W, 3), np.float32)
    rays = sc.update(ptgpu.PtParams(W, H, S, 10, 0, 1 if bvh else 0), ptgpu.PtCamera.from_floats(ex["camera"]), 0, out)
    sc.close()
    ref, ref_rays = osc.update(S, max_depth=10, frame_num=0)
    assert rays == ref_rays and np.array_equal(ref, out, equal_nan=True), _report(ref, out)


@pytest.mark.parametrize("n,bvh", [(120, False), (120, True), (500, False)])
def test_sphere_world_with_a_nested_checker_leaves_the_palette_kernels(ptgpu, oracle, n, bvh):
    """The wide MFMA kernels keep 16-bit palette codes on the attenuation stack, which needs every texture to be a Constant
    or a Checker of two Constants. A Checker whose child is itself a Checker (texture.rs:78-85 recurses) is not: such a
    world must fall back to the float stack (256-thread MFMA kernel) and still match the oracle bit for bit."""
    W, H, S = 96, 64, 3
    w = _random_sphere_world(oracle, 4242 + n, n, W, H, 6.0, 0.7)
    tex = np.concatenate([w["textures"], np.array([[1, 0, 0, 0, 4, 2, 0]], np.float32)])       # checker(odd = checker #4, even = constant #2)
    mats = np.concatenate([w["materials"], np.array([[0, 0, 0, 0, 0, len(tex) - 1]], np.float32)])
    rec = w["hitables"].copy()
    rec[::3, 1] = len(mats) - 1
    w = dict(w, textures=tex, materials=mats, hitables=rec)
    out, rays, ref, ref_rays = _render_world_both(ptgpu, oracle, w, W, H, S, bvh)
    assert rays == ref_rays and np.array_equal(ref, out, equal_nan=True), _report(ref, out)


@pytest.mark.parametrize("first", [70000, 70060, 70120, 70180])
def test_fuzz_slice_of_seeded_random_worlds(ptgpu, oracle, first):
    """A bounded slice (4 x 60 worlds x list/BVH) of tools/fuzz_worlds.py inside the suite: sphere worlds of assorted sizes
    with extreme extras, general worlds (rects, cuboids, instances, media, image textures) and moving-sphere worlds, random
    depth / frame number, every one bit-exact against the oracle; BVH sphere worlds additionally on the tree kernel."""
    W, H, S = 64, 48, 2
    bad = []
    for seed in range(first, first + 60):
        rng = np.random.default_rng(seed)
        kind = seed % 3
        if kind == 0:
            n = int(rng.choice([3, 20, 33, 64, 150, 400, 800]))
            extras = [(), ([0, -300, 0, 298],), ([0, 0, 0, 0.0], [2, 2, 2, -1e-3]), ([0, 0, 0, 25],)][int(rng.integers(0, 4))]
            w = _random_sphere_world(oracle, seed, n, W, H, float(rng.uniform(2, 15)), float(rng.uniform(0.1, 1.5)), extras)
        elif kind == 1:
            w = _random_world(oracle, seed, int(rng.integers(1, 40)), (0, 1, 2, 3, 4, 5), W, H, sky=(0.3, 0.3, 0.3) if seed & 1 else None)
            if seed % 4 == 1:
                w = _with_image_textures(w, seed)
        else:
            times = [((0.0, 1.0),), ((0.0, 1.0), (-1.0, 2.0)), ((0.25, 0.5),)][int(rng.integers(0, 3))]
            w = _random_world(oracle, seed, int(rng.choice([10, 60, 200])), (0, 1, 1), W, H, moving_times=times, media=False, instances=False)
        depth, frame = int(rng.choice([0, 1, 2, 5, 10, 10, 10, 25])), int(rng.choice([0, 0, 1, 7]))
        for bvh in (False, True):
            out, rays, ref, ref_rays = _render_world_both(ptgpu, oracle, w, W, H, S, bvh, depth=depth, frame=frame)
            ok = rays == ref_rays and np.array_equal(ref, out, equal_nan=True)
            if ok and bvh and kind != 1:
                out2, rays2, _, _ = _render_world_both(ptgpu, oracle, w, W, H, S, bvh, variant=256, depth=depth, frame=frame)
                ok = rays2 == ref_rays and np.array_equal(ref, out2, equal_nan=True)
            if not ok:
                bad.append((seed, kind, bvh))
    assert not bad, bad


def _with_image_textures(w, seed):
    """Repoint some lambertians / the emitter of a random world at Texture::Image sources (texture.rs:5-37)."""
    rng = np.random.default_rng(seed)
    images = [rng.integers(0, 256, (3, 5, 3), dtype=np.uint8), rng.integers(0, 256, (32, 64, 3), dtype=np.uint8)]
    tex = w["textures"].copy()
    base = len(tex)
    tex = np.concatenate([tex, np.array([[3, 0, 0, 0, 0, -1, 0], [3, 0, 0, 0, 1, -1, 0], [1, 0, 0, 0, base, 1, 0]], np.float32)])
    mats = w["materials"].copy()
    mats[0, 5], mats[2, 5] = base, base + 2                   # a lambertian image, a checker alternating image 0 / constant
    mats[mats[:, 0] == 3, 5] = base + 1                       # DiffuseLight emits the big image
    mats[mats[:, 0] == 4, 5] = base + 1                       # ... and so do the media's Isotropic phase functions (u = v = 0)
    out = dict(w)
    out.update(textures=tex, materials=mats, images=images)
    return out


@pytest.mark.parametrize("seed,n,kinds,bvh", [
    (31, 20, (2, 3, 4, 5), False),          # rects and cuboid faces carry real (u, v) (rect.rs:97-98), also under instances
    (32, 24, (0, 1, 2, 3, 4, 5), True),
    (33, 40, (0,), False),                  # sphere worlds: u = v = 0, one texel per image, folded at scene creation
    (34, 60, (0, 1), True),                 # ... also on the MOVING kernels
])
def test_image_textures_match_the_oracle(ptgpu, oracle, seed, n, kinds, bvh):
    W, H, S = 120, 80, 4
    w = _with_image_textures(_random_world(oracle, seed, n, kinds, W, H, sky=(0.4, 0.5, 0.6)), seed)
    out, rays, ref, ref_rays = _render_world_both(ptgpu, oracle, w, W, H, S, bvh)
    assert rays == ref_rays, "ray_count %d vs oracle %d; %s" % (rays, ref_rays, _report(ref, out))
    assert np.array_equal(ref, out), _report(ref, out)


@pytest.mark.parametrize("seed,times,bvh", [
    (11, ((0.0, 1.0),), False),                               # one shutter interval for every moving sphere
    (12, ((0.0, 1.0), (-0.5, 1.5), (0.0, 2.0)), False),       # per-sphere time_start / inv_time_delta
    (13, ((0.0, 1.0), (-0.5, 1.5)), True),
])
def test_random_moving_sphere_worlds_on_the_fast_kernels(ptgpu, oracle, seed, times, bvh):
    """Sphere + MovingSphere worlds big enough for the MFMA prefilter (>= 32 prefiltered spheres): the MOVING kernels,
    the general kernel (variant 128) and the oracle must agree bit for bit."""
    W, H, S = 160, 100, 4
    w = _random_world(oracle, seed, 150, (0, 0, 1, 1, 1), W, H, moving_times=times, media=False, instances=False)
    out, rays, ref, ref_rays = _render_world_both(ptgpu, oracle, w, W, H, S, bvh)
    assert rays == ref_rays and np.array_equal(ref, out), _report(ref, out)
    out2, rays2, _, _ = _render_world_both(ptgpu, oracle, w, W, H, S, bvh, variant=128)
    assert rays2 == ref_rays and np.array_equal(ref, out2), _report(ref, out2)


def test_moving_sphere_kernels_agree_with_the_general_kernel(ptgpu, pthost, oracle):
    """`random` (Sphere + MovingSphere) runs on the MOVING instantiations of the sphere kernels: MFMA prefilter over
    swept bounding spheres in list mode, internal tree over swept boxes in BVH mode. They must reproduce the general
    kernel (variant 128, itself checked against the oracle above) for the preset's shutter and for a narrower one,
    and a shutter outside the motion's interval must fall back to the general kernel rather than miss spheres."""
    W, H, S = 240, 160, 8
    for bvh in (False, True):
        hs = pthost.HostScene("random", W, H, samples=S, use_bvh=bvh, device=0)
        sc, p = hs.device_scene(), ptgpu.PtParams(W, H, S, 10, 0, 1 if bvh else 0)
        ref, ref_rays = oracle.OracleScene("random", W, H, use_bvh=bvh).update(S)
        for t0, t1 in ((0.0, 1.0), (0.25, 0.75), (0.5, 1.5)):
            cam = ptgpu.PtCamera.from_floats(np.ctypeslib.as_array(C.cast(C.pointer(hs.camera), C.POINTER(C.c_float)), shape=(24,)).copy())
            cam.time0, cam.time1 = t0, t1
            outs = []
            for variant in (0, 128):
                sc.set_tuning(0, variant)
                out = np.zeros((H, W, 3), np.float32)
                rays = sc.update(p, cam, 0, out)
                outs.append((rays, out))
            assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1]), (bvh, t0, t1, _report(outs[1][1], outs[0][1]))
            if (t0, t1) == (0.0, 1.0):
                assert outs[0][0] == ref_rays and np.array_equal(outs[0][1], ref), _report(ref, outs[0][1])
        sc.set_tuning(0, 0)


def test_mfma_prefilter_covers_moving_spheres(ptgpu, pthost):
    """Verify mode on `random`: every (ray, sphere-at-ray-time) pair with a positive reference discriminant must be
    among the candidates of the swept-bound prefilter."""
    W, H, S = 600, 400, 2
    hs = pthost.HostScene("random", W, H, samples=S, device=0)
    sc = hs.device_scene()
    sc.set_tuning(0, 8)
    sc.debug_counters(reset=True)
    out = np.zeros((H, W, 3), np.float32)
    sc.update(ptgpu.PtParams(W, H, S, 10, 0, 0), hs.camera, 0, out)
    c = sc.debug_counters(reset=True)
    missed, queued, overflow, positives = c["misses"], c["candidates"], c["overflows"], c["exact_positives"]
    sc.set_tuning(0, 0)
    print("random: positives %d queued %d overflow %d missed %d" % (positives, queued, overflow, missed))
    assert positives > 100000 and missed == 0
    assert queued < 6 * positives          # the swept bounds stay selective


@pytest.mark.parametrize("preset", ["random_spheres", "aras", "small", "random"])
def test_bvh_world_on_the_prefilter_kernel_equals_the_tree_kernel(ptgpu, pthost, oracle, preset):
    """A BVH world is a list world + the ancestor-AABB gate + DFS-rank ties at hit acceptance, so it runs on the MFMA
    list kernel when the scene fits it. Both kernels and the oracle's BVHNode::ray_hit recursion must agree."""
    W, H, S = 240, 160, 4
    hs = pthost.HostScene(preset, W, H, samples=S, use_bvh=True, device=0)
    sc, p = hs.device_scene(), ptgpu.PtParams(W, H, S, 10, 0, 1)
    ref, ref_rays = oracle.OracleScene(preset, W, H, use_bvh=True).update(S)
    for variant in (0, 256, 8):        # default, tree kernel, verify mode of the default
        sc.set_tuning(0, variant)
        out = np.zeros((H, W, 3), np.float32)
        rays = sc.update(p, hs.camera, 0, out)
        assert rays == ref_rays and np.array_equal(ref, out), (variant, _report(ref, out))
    sc.set_tuning(0, 0)


@pytest.mark.parametrize("preset,bvh", [("perlin_spheres", True), ("random_spheres", True), ("random", True), ("two_perlin_spheres", False),
                                        ("small", True), ("aras", False)])
def test_device_tree_build_equals_the_host_restatement(ptgpu, pthost, preset, bvh):
    """the traversal tree is built on the device (level-synchronous radix sorts, csrc/pt_build.hip).
    Byte for byte it must be the tree the host restatement of the same rules builds (PTGPU_HOST_BUILD=1)."""
    hs = pthost.HostScene(preset, 64, 48, samples=1, use_bvh=bvh, device=0)
    info = hs.device_scene().build_info()
    dev_nodes = hs.device_scene().debug_tree()
    os.environ["PTGPU_HOST_BUILD"] = "1"
    try:
        hs2 = pthost.HostScene(preset, 64, 48, samples=1, use_bvh=bvh, device=0)
    finally:
        del os.environ["PTGPU_HOST_BUILD"]
    info2 = hs2.device_scene().build_info()
    host_nodes = hs2.device_scene().debug_tree()
    assert info["on_device"] and not info2["on_device"]
    assert info["n_nodes"] == info2["n_nodes"] > 0 and info["depth"] == info2["depth"]
    assert info["build_ms"] > 0.0
    assert np.array_equal(dev_nodes, host_nodes), "first differing node %d" % int(np.argmax((dev_nodes != host_nodes).any(axis=1)))
    # every sphere of the tree appears exactly once as a leaf, every inner child exactly once
    child = dev_nodes[:, 24:28].view(np.int32)
    inner = child[(child >= 0) & (child != 0x7fffffff)]
    assert sorted(inner.tolist()) == list(range(1, info["n_nodes"]))
    leaves = ~child[child < 0]
    assert len(set(leaves.tolist())) == len(leaves)


@pytest.mark.parametrize("preset,bvh", [("perlin_spheres", True), ("random_spheres", True), ("random", True), ("small", False)])
def test_packed_tree_nodes_contain_the_built_boxes(ptgpu, pthost, preset, bvh):
    """The kernels read 64-byte nodes (pt_tree4.h DNode4Q): f16 plane offsets from the node's min corner. Packing may only
    ENLARGE a child box (lower planes rounded down, upper ones up) and by no more than one f16 step of the offset; child
    counts and the first inner child must survive it."""
    hs = pthost.HostScene(preset, 64, 48, samples=1, use_bvh=bvh, device=0)
    sc = hs.device_scene()
    full = sc.debug_tree()
    packed, usable = sc.debug_tree_packed()
    assert usable and len(packed) == len(full) > 0
    lo = full[:, 0:12].view(np.float32).reshape(-1, 3, 4).astype(np.float64)
    hi = full[:, 12:24].view(np.float32).reshape(-1, 3, 4).astype(np.float64)
    child = full[:, 24:28].view(np.int32)
    used = child != 0x7fffffff
    planes = np.ascontiguousarray(packed[:, 0:12]).view(np.float16).reshape(-1, 3, 2, 4).astype(np.float64)
    origin = packed[:, 12:15].view(np.float32).astype(np.float64)
    meta = packed[:, 15]
    assert np.array_equal((meta >> 19) & 7, used.sum(axis=1))
    assert np.array_equal((meta >> 16) & 7, ((child >= 0) & used).sum(axis=1))
    has_inner = ((meta >> 16) & 7) > 0
    assert np.array_equal((meta & 0xffff)[has_inner], child[has_inner, 0])
    for k in range(3):
        plo = origin[:, k, None] + planes[:, k, 0, :]
        phi = origin[:, k, None] + planes[:, k, 1, :]
        assert (plo[used] <= lo[:, k, :][used]).all() and (phi[used] >= hi[:, k, :][used]).all()
        ext = (hi[:, k, :] - origin[:, k, None])[used]
        step = np.maximum(np.abs(ext), 2.0 ** -14) * 2.0 ** -10   # one f16 step of the largest offset of the box
        assert ((lo[:, k, :] - plo)[used] <= step).all() and ((phi - hi[:, k, :])[used] <= step).all()
        assert np.allclose(origin[:, k], np.where(used, lo[:, k, :], np.inf).min(axis=1))


def test_traversal_counters_report_internal_tree_work(ptgpu, pthost, oracle):
    """BVH-mode work is reported as node visits / sphere tests per ray. Verify mode on a tree kernel counts
    them for the device's internal tree; the oracle counts the reference's both-children traversal. The image must
    not depend on the counting kernel, and the internal tree must visit far fewer nodes than the reference."""
    W, H, S = 240, 160, 2
    hs = pthost.HostScene("random_spheres", W, H, samples=S, use_bvh=True, device=0)
    sc, p = hs.device_scene(), ptgpu.PtParams(W, H, S, 10, 0, 1)
    plain = np.zeros((H, W, 3), np.float32)
    rays = sc.update(p, hs.camera, 0, plain)
    sc.set_tuning(0, 8 | 256)       # count on the tree kernel (256: do not use the MFMA list kernel for this BVH world)
    sc.traversal_counters(reset=True)
    counted = np.zeros((H, W, 3), np.float32)
    assert sc.update(p, hs.camera, 0, counted) == rays and np.array_equal(plain, counted)
    t = sc.traversal_counters(reset=True)
    sc.set_tuning(0, 0)
    c = (C.c_uint64 * 2)()
    oracle.lib().ora_bvh_counters(c, 1)
    _, ref_rays = oracle.OracleScene("random_spheres", W, H, use_bvh=True).update(S)
    oracle.lib().ora_bvh_counters(c, 1)
    assert ref_rays == rays
    print("per ray: reference %.1f nodes / %.1f leaf tests, internal tree %.1f nodes / %.1f sphere tests" % (
        c[0] / rays, c[1] / rays, t["nodes"] / rays, t["sphere_tests"] / rays))
    assert 1 < t["nodes"] / rays < c[0] / rays and t["sphere_tests"] > 0


def test_device_ln_is_glibc_logf(ptgpu, oracle):
    """constant_medium.rs:60 -(1/density) * ln(u): the value decides whether a ray scatters, so the device logf must
    equal the host's bit for bit. u is a multiple of 2^-24 in [0, 1); also sweep ordinary floats."""
    L = oracle.lib()
    u = (np.arange(0, 1 << 24, 5, dtype=np.uint32).astype(np.float32) * np.float32(2.0 ** -24))
    more = np.random.default_rng(3).uniform(-60, 60, 400000)
    x = np.concatenate([u, np.exp2(more).astype(np.float32), np.float32([1.0, 0.99999994, 1.0000001, 1e-45, 3e38])])
    want = np.zeros_like(x)
    L.ora_ln_array(x.ctypes.data, want.ctypes.data, len(x))
    got = ptgpu.selftest_probe(ptgpu.PROBE_LN, x)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), int((got.view(np.uint32) != want.view(np.uint32)).sum())


def test_shortened_device_math_is_exhaustively_equal_to_what_it_replaces(ptgpu):
    """The kernels take f32::sqrt through a shortened form of the compiler's correctly rounded lowering and build `2 * draw - 1`,
    `draw * 2 * PI`, `n + draw` with one rounding from the draw's integer, and Vec3::normalize's `1.0 / length` without the scaling
    steps of a general division (pt_device.h). All are unary in 32 bits, so the device simply tries every input: all 2^32 bit
    patterns for the square root and the reciprocal length, every draw beside 256 pixel coordinates for the rest. The normal's
    division by the radius (two operands) goes through 2^32 seeded pairs that include every special value, and so do the sphere
    test's quotients by d.d (every divisor in [0.5, 2] for the once-per-ray reciprocal); the general worlds' reciprocal ray direction
    is checked on all 2^32 inputs."""
    for probe in (ptgpu.PROBE_SWEEP_SQRT, ptgpu.PROBE_SWEEP_DRAWS, ptgpu.PROBE_SWEEP_INVLEN, ptgpu.PROBE_SWEEP_DIV, ptgpu.PROBE_SWEEP_DIVA, ptgpu.PROBE_SWEEP_RECIP):
        out = ptgpu.selftest_probe(probe, np.zeros(2, dtype=np.float32))
        assert out[0] == 0.0, (probe, float(out[0]), hex(int(out.view(np.uint32)[1])))


# ---- device primitives vs oracle primitives ------------------------------------------------------
def test_device_rng_matches_oracle(ptgpu, oracle):
    L = oracle.lib()
    seeds = np.arange(0, 4096, dtype=np.uint32) * np.uint32(2654435761) | np.uint32(1)
    got = ptgpu.selftest_probe(ptgpu.PROBE_RNG, seeds.view(np.float32))
    st = (C.c_uint64 * 4)()
    for i in range(0, 4096, 37):
        L.ora_xoshiro_seed_from_u64(int(seeds[i]), st)
        v = 0.0
        for _ in range((i & 15) + 1):
            v = L.ora_xoshiro_gen_f32(st)
        assert got[i] == np.float32(v)


def test_device_sinf_cosf_bit_exact(ptgpu, oracle):
    L = oracle.lib()
    xs = np.concatenate([np.linspace(0, 2 * np.pi, 20001), np.random.default_rng(0).uniform(0, 6.2832, 20000)])
    xs = xs.astype(np.float32)
    gs, gc = ptgpu.selftest_probe(ptgpu.PROBE_SIN, xs), ptgpu.selftest_probe(ptgpu.PROBE_COS, xs)
    s, c = C.c_float(), C.c_float()
    for i in range(0, len(xs), 7):
        L.ora_sinf_cosf(float(xs[i]), C.byref(s), C.byref(c))
        assert gs[i] == np.float32(s.value) and gc[i] == np.float32(c.value), float(xs[i])


def test_device_pow5_vs_host_powf_mismatch_rate(ptgpu, oracle):
    """schlick's powf(1 - cos, 5.0) (math.rs:79) is the one control-affecting libm call. glibc's powf is
    not correctly rounded, so bit-identity cannot be promised; measure how often the device's
    correctly-rounded x^5 differs from this host's powf, and that it never differs by more than 1 ulp."""
    L = oracle.lib()
    cosine = np.random.default_rng(5).uniform(0.0, 1.0, 200000).astype(np.float32)
    x = np.float32(1.0) - cosine                                  # the f32 subtraction of math.rs:79
    got = ptgpu.selftest_probe(ptgpu.PROBE_POW5, x)
    # schlick(cos, ri) = r0 + (1 - r0) * powf(1 - cos, 5): at ri = 1, r0 = 0 and the result IS the host powf
    host = np.array([L.ora_schlick(float(ci), 1.0) for ci in cosine[:20000]], np.float32)
    diff = got[:20000] != host
    ulp = np.abs(got[:20000].view(np.int32).astype(np.int64) - host.view(np.int32).astype(np.int64))
    assert ulp.max() <= 1
    rate = diff.mean()
    print("pow5 vs host powf: %.4f %% of inputs differ by 1 ulp" % (100 * rate))
    assert rate < 0.02
    exact = (x.astype(np.float64) ** 5).astype(np.float32)       # correctly rounded reference
    assert (got != exact).mean() < 1e-4


def test_bench_sharded_path_self_check():
    """bench.py's N > 1 path (BASELINE config 4 through pt_comm_gather_frame, three buffer sets, exchange on a second stream)
    on the one rank a 1-GPU box can form, at a reduced sample count: every multi-rank run first checks, untimed, that the gathered
    frame equals the rank's own full render bit for bit (and says so in its line; a mismatch exits non-zero); the JSON line must say
    strong scaling and carry the roofline block."""
    import json
    import subprocess
    env = dict(os.environ, PT_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    env.pop("PT_BENCH_CHECK", None)   # (the check is on by default)
    for extra, spp in (([], "16"), (["--no-overlap"], "8")):   # (16 spp: the shard renders as two launches; 8: as one)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--samples", spp,
                              "--no-extras", "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        assert "[bench check] tiles frame over 1 rank(s) == single-GPU frame" in out.stderr
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        assert d["scaling"] == "strong" and d["unit"] == "Mrays/s" and d["value"] > 0 and "roofline" in d
        assert d["self_check"]["sharded_equals_single"] is True and d["self_check"]["ranks_seen"] == 1 and d["self_check"]["rccl"] >= 21000
        assert d["self_check"]["rays"] == d["config"]["rays_per_step"] and d["config"]["sharded_equals_single"] is True
        assert "split by rows" in d["config"]["workload"]


@pytest.mark.parametrize("ranks,extra", [(2, []), (3, ["--no-overlap"]), (8, [])])   # (8: the node the scaling run will use -- `bench.py --gpus 8` as the driver starts it)
def test_bench_multi_rank_path_on_one_gpu_behind_the_cross_process_test_double(ranks, extra):
    """bench.py's OWN N > 1 code path with N processes, started exactly as the driver starts them (python -m torch.distributed.run
    --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N): rank 0's unique id travels over the launcher's process
    group, every rank creates its pt_comm, renders its rows of BASELINE config 4's frame (at a reduced sample count), the shards are
    exchanged through pt_comm_gather_frame (on a second stream) or pt_render_sharded -- and the script's self-check, which runs before the
    warm-up on every multi-rank run, finds the gathered frame on EVERY rank equal to that rank's own unsharded render and the reduced ray
    count equal to the frame's. One GPU only, so: all ranks on device 0, the launcher's group over gloo, and the C ABI's RCCL pinned
    (PTGPU_RCCL_LIBRARY) to tests/mock_rccl/mock_rccl_xproc.hip, which moves the data through shared memory. Real RCCL stays out of it."""
    import json
    import subprocess
    src, lib = os.path.join(MOCK_RCCL, "mock_rccl_xproc.hip"), os.path.join(MOCK_RCCL, "_build", "librccl_xproc.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(lib), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-shared", "-fPIC", "-I/opt/rocm/include", src, "-o", lib, "-lrt"])
    env = dict(os.environ, PT_BENCH_ONE_DEVICE="1", PT_BENCH_BACKEND="gloo", PTGPU_RCCL_LIBRARY=lib)
    env.pop("PT_BENCH_CHECK", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1", "--master-port", str(29540 + ranks),
           os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--samples", "16", "--no-extras", "--no-cpu-baseline"] + extra
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    finally:
        for f in glob.glob("/dev/shm/mock_rccl_*"):    # (the last rank to leave unlinks the segment; a run that died half-way cannot)
            try:
                os.unlink(f)
            except OSError:
                pass
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    assert "[bench check] tiles frame over %d rank(s) == single-GPU frame on every rank" % ranks in out.stderr
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == ranks and d["scaling"] == "strong" and d["value"] > 0
    assert d["self_check"]["sharded_equals_single"] is True and d["self_check"]["ranks_seen"] == ranks and d["self_check"]["rccl"] == 29901
    assert d["config"]["ranks_seen"] == ranks and "mock_rccl" in d["config"]["rccl"]["library"]
    assert d["self_check"]["rays"] == d["config"]["rays_per_step"]


def test_bench_default_line_keeps_the_contract():
    """The one JSON line of `python bench.py` (N = 1): BASELINE's metric on BASELINE's workload, every fraction of the roofline
    block <= 1, the extras present and self-consistent (bench.py asserts that each extra reproduces the headline frame)."""
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["metric"] == "Mrays/sec, random_spheres 1200x800 64spp" and d["unit"] == "Mrays/s" and d["n_gpus"] == 1
    assert d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["config"]["rays_per_step"] == 162554454
    assert abs(d["value"] - d["config"]["rays_per_step"] / 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    # the line names the build it timed; counters committed for ANOTHER build are marked stale and their fractions withheld (never silently reused)
    assert re.fullmatch(r"ptgpu \d+\.\d+ gfx950 src [0-9a-f]{12}( defs .+)?", d["build"]), d["build"]
    frac_of = lambda blk: blk["frac_stale_counters"] if blk["stale_counters"] else blk["frac"]
    assert r["stale_counters"] == (r["counters_build"] != d["build"]) and (r["frac"] is None) == r["stale_counters"]
    assert r["bound"] == "valu_issue" and 0.0 < frac_of(r) <= 1.0 and 0.0 < r["hbm_frac"] <= 1.0 and 0.0 <= r["mfma_busy_frac"] <= 1.0
    assert r["traffic"] > 0 and r["kernel_ms"] <= r["pass_ms"] <= d["ms_per_step"] * 1.05
    for key in ("host_contract", "pipelined_frames", "progressive_view"):
        assert d[key]["value"] > 0 and d[key]["unit"] == "Mrays/s"
    assert d["host_contract"]["reused_buffer"]["value"] > 0 and d["host_contract"]["registered"]["value"] > 0 and "value_note" in d
    # every BASELINE.json configuration that fits one GPU is in the driver-timed line, at its own frame size
    bc = d["baseline_configs"]
    assert len(bc) == 3 and all(v["value"] > 0 and v["frames"] >= 3 and v["unit"] == "Mrays/s" for v in bc.values())
    assert bc["config 4 on one GPU: random_spheres 1200x800 256spp"]["rays_per_frame"] > 4 * 162000000
    c5 = bc["config 5: perlin_spheres 1920x1080 128spp BVH"]
    assert c5["hitables"] == 10002 and 0.0 < frac_of(c5["roofline"]) <= 1.0

