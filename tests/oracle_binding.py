"""ctypes binding of the CPU oracle (oracle/ptref.h). TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "_build", "libptref.so")

_lib = None


def build(force=False):
    if force or not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "_build/libptref.so"], stdout=subprocess.DEVNULL)
    return LIB


def build_native(out_dir=None):
    """-O3 -march=native build for the cpu_baseline leg; compiled on the box that runs it."""
    out_dir = out_dir or os.path.join(ORACLE_DIR, "_build")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libptref_native_%s.so" % os.uname().nodename.replace("/", "_"))
    src = os.path.join(ORACLE_DIR, "ptref.c")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-march=native", "-std=c11", "-D_GNU_SOURCE", "-fPIC", "-ffp-contract=off",
                               "-fno-fast-math", "-fexcess-precision=standard", "-shared", "-o", out, src, "-lm",
                               "-lpthread"])
    return out


def _bind(L):
    vp, u32, u64, i32, f32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_float
    L.ora_scene_from_preset.restype = vp
    L.ora_scene_from_preset.argtypes = [C.c_char_p, u32, u32, C.c_int]
    L.ora_scene_free.argtypes = [vp]
    L.ora_scene_from_world.restype = vp
    L.ora_scene_from_world.argtypes = [vp, u32, vp, u32, vp, u32, vp, u32, vp, C.c_int, vp, C.c_int, vp, vp, u32]
    L.ora_scene_free.restype = None
    L.ora_scene_from_graph.restype = vp
    L.ora_scene_from_graph.argtypes = [vp, u32, vp, u32, vp, u32, vp, u32, vp, C.c_int, vp, vp, u32, vp, u32, u32]
    L.ora_scene_from_graph_bvh.restype = vp
    L.ora_scene_from_graph_bvh.argtypes = [vp, u32, vp, u32, vp, u32, vp, u32, vp, C.c_int, vp, vp, u32, vp, u32, u32, vp, vp, u32]
    L.ora_scene_update.restype = u64
    L.ora_scene_update.argtypes = [vp, u32, u32, u32, u32, u32, vp, C.c_int]
    L.ora_scene_update_range.restype = u64
    L.ora_scene_update_range.argtypes = [vp, u32, u32, u32, u32, u32, vp, u64, u64, C.c_int]
    L.ora_scene_update_pixels.restype = u64
    L.ora_scene_update_pixels.argtypes = [vp, u32, u32, u32, u32, u32, vp, vp, u64, C.c_int]
    L.ora_scene_update_pixels_counted.restype = u64
    L.ora_scene_update_pixels_counted.argtypes = [vp, u32, u32, u32, u32, u32, vp, vp, u64, vp, C.c_int]
    for n in ("num_spheres", "num_materials", "num_textures", "num_bvh_nodes", "num_hitables", "num_transforms"):
        f = getattr(L, "ora_scene_" + n)
        f.restype, f.argtypes = u32, [vp]
    L.ora_scene_bvh_root.restype, L.ora_scene_bvh_root.argtypes = i32, [vp]
    L.ora_scene_has_perlin_texture.restype, L.ora_scene_has_perlin_texture.argtypes = C.c_int, [vp]
    L.ora_scene_build_draws.restype, L.ora_scene_build_draws.argtypes = u64, [vp]
    L.ora_scene_is_sphere_world.restype, L.ora_scene_is_sphere_world.argtypes = C.c_int, [vp]
    L.ora_scene_export_world.argtypes = [vp, vp, vp]
    L.ora_scene_export_spheres.argtypes = [vp, vp, vp]
    L.ora_scene_export_materials.argtypes = [vp, vp]
    L.ora_scene_export_textures.argtypes = [vp, vp]
    L.ora_scene_export_perlin.argtypes = [vp, vp, vp, vp, vp]
    L.ora_scene_export_bvh.argtypes = [vp, vp, vp]
    L.ora_scene_export_camera.argtypes = [vp, vp]
    L.ora_scene_export_sky.restype, L.ora_scene_export_sky.argtypes = C.c_int, [vp, vp]
    L.ora_splitmix64.argtypes = [u64, vp, C.c_int]
    L.ora_xoshiro_seed_from_u64.argtypes = [u64, vp]
    L.ora_xoshiro_next_u64.restype, L.ora_xoshiro_next_u64.argtypes = u64, [vp]
    L.ora_xoshiro_gen_f32.restype, L.ora_xoshiro_gen_f32.argtypes = f32, [vp]
    L.ora_xoshiro_gen_range_i32.restype, L.ora_xoshiro_gen_range_i32.argtypes = i32, [vp, i32, i32]
    L.ora_pixel_seed.restype, L.ora_pixel_seed.argtypes = u64, [u32, u32, u32]
    L.ora_sinf_cosf.argtypes = [f32, vp, vp]
    L.ora_ln_array.argtypes = [vp, vp, u64]
    L.ora_bvh_counters.argtypes = [vp, C.c_int]
    L.ora_hitable_ray_hit.restype = C.c_int
    L.ora_hitable_ray_hit.argtypes = [vp, u32, vp, vp, f32, f32, f32, vp, vp, vp]
    L.ora_sphere_ray_hit.restype, L.ora_sphere_ray_hit.argtypes = C.c_int, [vp, vp, vp, f32, f32, vp]
    L.ora_aabb_ray_hit.restype, L.ora_aabb_ray_hit.argtypes = C.c_int, [vp, vp, vp, vp, f32, f32]
    L.ora_schlick.restype, L.ora_schlick.argtypes = f32, [f32, f32]
    for n in ("random_unit_vector", "random_in_unit_sphere", "random_in_unit_disk"):
        getattr(L, "ora_" + n).argtypes = [vp, vp]
    L.ora_camera_get_ray.argtypes = [vp, f32, f32, vp, vp]
    L.ora_camera_new.argtypes = [vp, vp, vp, f32, f32, f32, f32, f32, f32, vp]
    L.ora_perlin_noise.restype, L.ora_perlin_noise.argtypes = f32, [vp, vp]
    L.ora_perlin_turb.restype, L.ora_perlin_turb.argtypes = f32, [vp, vp]
    L.ora_texture_value.argtypes = [vp, u32, vp, vp]
    L.ora_ray_trace.argtypes = [vp, vp, vp, f32, u32, vp, vp, vp]
    L.ora_soa_ray_hit.restype, L.ora_soa_ray_hit.argtypes = C.c_int, [vp, C.c_int, vp, vp, f32, f32, vp, vp]
    L.ora_world_ray_hit.restype, L.ora_world_ray_hit.argtypes = C.c_int, [vp, vp, vp, f32, f32, f32, vp, vp, vp]
    L.ora_bench_ray_hit.restype, L.ora_bench_ray_hit.argtypes = C.c_double, [vp, C.c_int, vp, vp, f32, u64]
    L.ora_linear_to_srgb.argtypes = [vp, vp]
    L.ora_frame_to_srgb8.argtypes = [vp, u32, u32, vp]
    return L


def lib(path=None):
    global _lib
    if path is not None:
        return _bind(C.CDLL(path))
    if _lib is None:
        _lib = _bind(C.CDLL(build()))
    return _lib


class OracleScene:
    """offline.rs:16-24: seed-0 rng -> Storage::new -> preset -> new_scene."""

    def __init__(self, preset, width, height, use_bvh=False, library=None):
        self.L = library or lib()
        self.h = self.L.ora_scene_from_preset(preset.encode(), width, height, 1 if use_bvh else 0)
        if not self.h:
            raise KeyError("unrecognised preset %r" % preset)
        self.preset, self.width, self.height, self.use_bvh = preset, width, height, bool(use_bvh)

    @classmethod
    def from_world(cls, hitables, transforms, materials, textures, camera, width, height, sky=None, use_bvh=False,
                   library=None, images=()):
        """Scene from the flat description (the arrays export() returns): arbitrary worlds for parity tests."""
        self = cls.__new__(cls)
        self.L = library or lib()
        rec = np.ascontiguousarray(hitables, dtype=np.uint32).reshape(-1, 16)
        xf = np.ascontiguousarray(transforms, dtype=np.float32).reshape(-1, 24)
        mats = np.ascontiguousarray(materials, dtype=np.float32).reshape(-1, 6)
        texs = np.ascontiguousarray(textures, dtype=np.float32).reshape(-1, 7)
        cam = np.ascontiguousarray(camera, dtype=np.float32).reshape(24)
        sk = np.ascontiguousarray(sky if sky is not None else [0, 0, 0], dtype=np.float32)
        # images: list of [H, W, 3] uint8 arrays (RgbImage, texture.rs:5-10)
        wh = np.array([[im.shape[1], im.shape[0]] for im in images], np.uint32).reshape(-1, 2)
        blob = np.concatenate([np.ascontiguousarray(im, np.uint8).reshape(-1) for im in images]) if len(images) else np.zeros(1, np.uint8)
        self.h = self.L.ora_scene_from_world(rec.ctypes.data, len(rec), xf.ctypes.data, len(xf), mats.ctypes.data, len(mats),
                                             texs.ctypes.data, len(texs), cam.ctypes.data, 1 if sky is not None else 0,
                                             sk.ctypes.data, 1 if use_bvh else 0, wh.ctypes.data, blob.ctypes.data, len(images))
        if not self.h:
            raise ValueError("malformed world description")
        self.preset, self.width, self.height, self.use_bvh = "<world>", width, height, bool(use_bvh)
        return self

    @classmethod
    def from_graph(cls, hitables, transforms, materials, textures, camera, width, height, nodes, node_children, root_node, sky=None, library=None,
                   bvh_minmax=None, bvh_children=None):
        """Scene whose world is a scene graph built literally (List in List, Instance of Instance, Instance around a medium, a medium
        around a List or another medium, BVHNodes anywhere: node kind 4 = row of bvh_minmax [n, 6] / bvh_children [n, 2] NODE indices)
        over the leaf shapes `hitables`; `nodes` = [n, 4] uint32 rows as include/ptgpu.h pt_node. List worlds only."""
        self = cls.__new__(cls)
        self.L = library or lib()
        rec = np.ascontiguousarray(hitables, dtype=np.uint32).reshape(-1, 16)
        xf = np.ascontiguousarray(transforms, dtype=np.float32).reshape(-1, 24)
        mats = np.ascontiguousarray(materials, dtype=np.float32).reshape(-1, 6)
        texs = np.ascontiguousarray(textures, dtype=np.float32).reshape(-1, 7)
        cam = np.ascontiguousarray(camera, dtype=np.float32).reshape(24)
        sk = np.ascontiguousarray(sky if sky is not None else [0, 0, 0], dtype=np.float32)
        nd = np.ascontiguousarray(nodes, dtype=np.uint32).reshape(-1, 4)
        ch = np.ascontiguousarray(node_children, dtype=np.uint32).reshape(-1)
        if len(ch) == 0:
            ch = np.zeros(1, np.uint32)
        mm = np.ascontiguousarray(bvh_minmax if bvh_minmax is not None else np.zeros((0, 6)), dtype=np.float32).reshape(-1, 6)
        lr = np.ascontiguousarray(bvh_children if bvh_children is not None else np.zeros((0, 2)), dtype=np.int32).reshape(-1, 2)
        self.h = self.L.ora_scene_from_graph_bvh(rec.ctypes.data, len(rec), xf.ctypes.data, len(xf), mats.ctypes.data, len(mats), texs.ctypes.data, len(texs),
                                                 cam.ctypes.data, 1 if sky is not None else 0, sk.ctypes.data, nd.ctypes.data, len(nd), ch.ctypes.data,
                                                 len(node_children), int(root_node), mm.ctypes.data if len(mm) else None, lr.ctypes.data if len(lr) else None, len(mm))
        if not self.h:
            raise ValueError("malformed scene graph")
        self.preset, self.width, self.height, self.use_bvh = "<graph>", width, height, False
        return self

    def close(self):
        if self.h:
            self.L.ora_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # Scene::update
    def update(self, samples, max_depth=10, frame_num=0, buffer=None, nthreads=0, pix_range=None, pixels=None, pixel_rays=None):
        """pixel_rays (with pixels): a uint32 array that receives the rays traced for each listed pixel."""
        W, H = self.width, self.height
        if buffer is None:
            buffer = np.zeros((H, W, 3), dtype=np.float32)
        assert buffer.dtype == np.float32 and buffer.flags["C_CONTIGUOUS"] and buffer.size == W * H * 3
        if pixel_rays is not None:
            px = np.ascontiguousarray(pixels, dtype=np.uint32)
            assert pixel_rays.dtype == np.uint32 and pixel_rays.flags["C_CONTIGUOUS"] and pixel_rays.size == len(px)
            rc = self.L.ora_scene_update_pixels_counted(self.h, W, H, samples, max_depth, frame_num, buffer.ctypes.data,
                                                        px.ctypes.data, len(px), pixel_rays.ctypes.data, nthreads)
        elif pixels is not None:
            px = np.ascontiguousarray(pixels, dtype=np.uint32)
            rc = self.L.ora_scene_update_pixels(self.h, W, H, samples, max_depth, frame_num, buffer.ctypes.data,
                                                px.ctypes.data, len(px), nthreads)
        elif pix_range is not None:
            rc = self.L.ora_scene_update_range(self.h, W, H, samples, max_depth, frame_num, buffer.ctypes.data,
                                               pix_range[0], pix_range[1], nthreads)
        else:
            rc = self.L.ora_scene_update(self.h, W, H, samples, max_depth, frame_num, buffer.ctypes.data, nthreads)
        return buffer, rc

    # closest-hit queries on explicit rays --------------------------------
    def soa_ray_hit(self, lanes, origin, direction, t_min=0.001, t_max=3.4028234663852886e38):
        """SpheresSoA::hit_scalar / hit_sse4_1 / hit_avx2 (lanes 1 / 4 / 8; spheres_soa.rs). None for a miss, else
        (t, index, point3, normal3, u, v); raises when the list holds anything but spheres."""
        o, d = np.asarray(origin, np.float32), np.asarray(direction, np.float32)
        out, idx = np.zeros(9, np.float32), C.c_uint32(0)
        rc = self.L.ora_soa_ray_hit(self.h, lanes, o.ctypes.data, d.ctypes.data, t_min, t_max, out.ctypes.data, C.byref(idx))
        if rc < 0:
            raise ValueError("SpheresSoA takes Hitable::Sphere entries only")
        return None if rc == 0 else (out[6], idx.value, out[0:3].copy(), out[3:6].copy(), out[7], out[8])

    def world_ray_hit(self, origin, direction, time=0.0, t_min=0.001, t_max=3.4028234663852886e38):
        """Hitable::ray_hit on the scene's world (list, or BVH when built with one). None or (t, index, point3, normal3)."""
        o, d = np.asarray(origin, np.float32), np.asarray(direction, np.float32)
        out, idx, st = np.zeros(9, np.float32), C.c_uint32(0), np.zeros(4, np.uint64)
        rc = self.L.ora_world_ray_hit(self.h, o.ctypes.data, d.ctypes.data, time, t_min, t_max, st.ctypes.data, out.ctypes.data, C.byref(idx))
        return None if rc == 0 else (out[6], idx.value, out[0:3].copy(), out[3:6].copy())

    def bench_ray_hit(self, which, origin, direction, time=0.0, reps=100000):
        """ns per closest-hit query (bench.rs:8-26's unit): which 0 = the world, 1 / 4 / 8 = SpheresSoA variants."""
        o, d = np.asarray(origin, np.float32), np.asarray(direction, np.float32)
        return float(self.L.ora_bench_ray_hit(self.h, which, o.ctypes.data, d.ctypes.data, time, reps))

    # flat export ---------------------------------------------------------
    def export(self):
        L, h = self.L, self.h
        n = L.ora_scene_num_hitables(h)
        xyzr = mid = None
        if L.ora_scene_is_sphere_world(h):
            xyzr = np.zeros((n, 4), np.float32)
            mid = np.zeros(n, np.uint32)
            L.ora_scene_export_spheres(h, xyzr.ctypes.data, mid.ctypes.data)
        records = np.zeros((n, 16), np.uint32)
        ntr = L.ora_scene_num_transforms(h)
        transforms = np.zeros((max(ntr, 1), 24), np.float32)
        L.ora_scene_export_world(h, records.ctypes.data, transforms.ctypes.data)
        transforms = transforms[:ntr]
        nm = L.ora_scene_num_materials(h)
        mats = np.zeros((nm, 6), np.float32)
        L.ora_scene_export_materials(h, mats.ctypes.data)
        nt = L.ora_scene_num_textures(h)
        texs = np.zeros((max(nt, 1), 7), np.float32)
        L.ora_scene_export_textures(h, texs.ctypes.data)
        texs = texs[:nt]
        rv = np.zeros((256, 3), np.float32)
        px, py, pz = (np.zeros(256, np.uint32) for _ in range(3))
        L.ora_scene_export_perlin(h, rv.ctypes.data, px.ctypes.data, py.ctypes.data, pz.ctypes.data)
        nn = L.ora_scene_num_bvh_nodes(h)
        minmax = np.zeros((max(nn, 1), 6), np.float32)
        lr = np.zeros((max(nn, 1), 2), np.int32)
        if nn:
            L.ora_scene_export_bvh(h, minmax.ctypes.data, lr.ctypes.data)
        cam = np.zeros(24, np.float32)
        L.ora_scene_export_camera(h, cam.ctypes.data)
        sky = np.zeros(3, np.float32)
        has_sky = L.ora_scene_export_sky(h, sky.ctypes.data)
        return dict(hitables=records, transforms=transforms, spheres=xyzr, sphere_material=mid, materials=mats, textures=texs, perlin=(rv, px, py, pz),
                    has_perlin=bool(L.ora_scene_has_perlin_texture(h)), bvh_minmax=minmax[:nn], bvh_children=lr[:nn],
                    bvh_root=L.ora_scene_bvh_root(h), camera=cam, sky=(sky if has_sky else None),
                    build_draws=L.ora_scene_build_draws(h))


def to_ptgpu_world_desc(ptgpu, ex, images=()):
    """Oracle export -> the product's pt_world_desc (general worlds; the same description fed to both sides)."""
    materials = [(int(r[0]), r[1:4], r[4], int(r[5])) for r in ex["materials"]]
    textures = [(int(r[0]), r[1:4], int(r[4]), int(r[5]), r[6]) for r in ex["textures"]]
    bvh = (ex["bvh_minmax"], ex["bvh_children"]) if len(ex["bvh_minmax"]) else None
    return ptgpu.WorldDesc(ex["hitables"], ex["transforms"], materials, textures,
                           perlin=ex["perlin"] if ex["has_perlin"] else None, bvh_nodes=bvh,
                           bvh_root=ex["bvh_root"], sky=ex["sky"], images=images)


def to_ptgpu_desc(ptgpu, ex):
    """Turn an oracle export into the product's pt_scene_desc (tests feed the SAME scene to both)."""
    materials = [(int(r[0]), r[1:4], r[4], int(r[5])) for r in ex["materials"]]
    textures = [(int(r[0]), r[1:4], int(r[4]), int(r[5]), r[6]) for r in ex["textures"]]
    bvh = (ex["bvh_minmax"], ex["bvh_children"]) if len(ex["bvh_minmax"]) else None
    return ptgpu.SceneDesc(ex["spheres"], ex["sphere_material"], materials, textures,
                           perlin=ex["perlin"] if ex["has_perlin"] else None, bvh_nodes=bvh,
                           bvh_root=ex["bvh_root"], sky=ex["sky"])
