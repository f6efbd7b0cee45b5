"""ctypes binding of libptgpu.so (include/ptgpu.h) -- plumbing only.

The product is the HIP library; this module only marshals PODs. It fails
loudly when the shared library is missing: there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PTGPU_BUILD_DIR (development): a build directory other than the shipped _build -- `make -C pathtrace-rs_amd B=_build_dev DEFS=...` puts a
# complete set (libptgpu.so + libpthost.so + CLI, linked to each other by $ORIGIN) there, so A/B runs never overwrite the product.
_BUILD = os.environ.get("PTGPU_BUILD_DIR") or "_build"
LIB_PATH = os.path.join(_BUILD if os.path.isabs(_BUILD) else os.path.join(_HERE, _BUILD), "libptgpu.so")

PT_OK = 0
PT_ERR_INVALID_ARG = 1
PT_ERR_HIP = 2
PT_ERR_NO_DEVICE = 3
PT_ERR_UNSUPPORTED = 4

MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_DIFFUSE_LIGHT, MAT_ISOTROPIC = 0, 1, 2, 3, 4
HIT_SPHERE, HIT_MOVING_SPHERE, HIT_RECT_XY, HIT_RECT_XZ, HIT_RECT_YZ, HIT_CUBOID = 0, 1, 2, 3, 4, 5
TEX_CONSTANT, TEX_CHECKER, TEX_NOISE, TEX_IMAGE = 0, 1, 2, 3


class PtParams(C.Structure):  # params.rs:11-18
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("samples", C.c_uint32),
                ("max_depth", C.c_uint32), ("random_seed", C.c_uint32), ("use_bvh", C.c_uint32)]


class PtCamera(C.Structure):  # camera.rs:8-19
    _fields_ = [("origin", C.c_float * 3), ("lower_left_corner", C.c_float * 3),
                ("horizontal", C.c_float * 3), ("vertical", C.c_float * 3),
                ("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3),
                ("time0", C.c_float), ("time1", C.c_float), ("lens_radius", C.c_float)]

    @classmethod
    def from_floats(cls, f24):
        f = np.ascontiguousarray(f24, dtype=np.float32)
        assert f.size == 24
        cam = cls()
        C.memmove(C.addressof(cam), f.ctypes.data, 96)
        return cam


class PtSphere(C.Structure):
    _fields_ = [("cx", C.c_float), ("cy", C.c_float), ("cz", C.c_float), ("radius", C.c_float)]


class PtMaterial(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("albedo", C.c_float * 3), ("param", C.c_float), ("texture", C.c_int32)]


class PtTexture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("color", C.c_float * 3), ("odd", C.c_int32), ("even", C.c_int32),
                ("scale", C.c_float)]


class PtPerlin(C.Structure):
    _fields_ = [("randvec", (C.c_float * 3) * 256), ("perm_x", C.c_uint32 * 256),
                ("perm_y", C.c_uint32 * 256), ("perm_z", C.c_uint32 * 256)]


class PtBvhNode(C.Structure):
    _fields_ = [("min", C.c_float * 3), ("max", C.c_float * 3), ("lhs", C.c_int32), ("rhs", C.c_int32)]


class PtSceneDesc(C.Structure):
    _fields_ = [("n_spheres", C.c_uint32), ("spheres", C.POINTER(PtSphere)),
                ("sphere_material", C.POINTER(C.c_uint32)),
                ("n_materials", C.c_uint32), ("materials", C.POINTER(PtMaterial)),
                ("n_textures", C.c_uint32), ("textures", C.POINTER(PtTexture)),
                ("perlin", C.POINTER(PtPerlin)),
                ("n_bvh_nodes", C.c_uint32), ("bvh_nodes", C.POINTER(PtBvhNode)), ("bvh_root", C.c_int32),
                ("has_sky", C.c_uint32), ("sky", C.c_float * 3)]


class PtHitable(C.Structure):  # one HitableList entry of a general world (64 bytes)
    _fields_ = [("kind", C.c_uint32), ("material", C.c_uint32), ("flip_normals", C.c_uint32),
                ("transform", C.c_int32), ("medium_material", C.c_int32), ("density", C.c_float),
                ("p", C.c_float * 10)]


class PtAffine(C.Structure):  # Affine3A columns + translation, then the inverse
    _fields_ = [("m", C.c_float * 12), ("inv", C.c_float * 12)]


class PtImage(C.Structure):  # texture.rs:5-10 RgbImage
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgb", C.POINTER(C.c_uint8))]


class PtWorldDesc(C.Structure):
    _fields_ = [("n_hitables", C.c_uint32), ("hitables", C.POINTER(PtHitable)),
                ("n_transforms", C.c_uint32), ("transforms", C.POINTER(PtAffine)),
                ("n_materials", C.c_uint32), ("materials", C.POINTER(PtMaterial)),
                ("n_textures", C.c_uint32), ("textures", C.POINTER(PtTexture)),
                ("perlin", C.POINTER(PtPerlin)),
                ("n_bvh_nodes", C.c_uint32), ("bvh_nodes", C.POINTER(PtBvhNode)), ("bvh_root", C.c_int32),
                ("has_sky", C.c_uint32), ("sky", C.c_float * 3),
                ("n_images", C.c_uint32), ("images", C.POINTER(PtImage)),
                ("n_nodes", C.c_uint32), ("nodes", C.c_void_p), ("n_node_children", C.c_uint32), ("node_children", C.POINTER(C.c_uint32)),
                ("root_node", C.c_uint32)]


class PtKernelChoice(C.Structure):
    """pt_kernel_choice: which kernel a frame runs on (include/ptgpu.h)."""
    _fields_ = [(n, C.c_uint32) for n in ("family", "block", "lds_bytes", "blocks_per_cu", "moving", "gate", "verify", "ref_bvh", "ordered",
                                          "stack_in_lds", "global_stack", "n_tiles", "world_hit_lds", "world_occ", "world_media", "refill_min", "coop", "world_graph", "world_lazy", "pool_slots")] + \
               [("name", C.c_char * 96)]

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n, _ in self._fields_[:-1]}
        d["name"] = self.name.decode()
        return d


QUERY_LIST, QUERY_BVH, QUERY_SOA_SCALAR, QUERY_SOA_SSE4_1, QUERY_SOA_AVX2 = range(5)
FAMILY_WORLD, FAMILY_TREE_BINARY, FAMILY_TREE4, FAMILY_MFMA, FAMILY_SCAN_LDS, FAMILY_SCAN_HBM = range(6)

EXPORTS = [
    "pt_device_count", "pt_scene_create", "pt_scene_create_world", "pt_scene_prepare", "pt_scene_destroy", "pt_render", "pt_render_device",
    "pt_render_shard_device", "pt_shard_rows", "pt_scene_set_seed_base", "pt_last_kernel_ms",
    "pt_last_launch_info", "pt_scene_set_tuning", "pt_last_error", "pt_version", "pt_selftest_probe", "pt_scene_debug_counters", "pt_scene_traversal_counters",
    "pt_last_pass_ms", "pt_comm_unique_id", "pt_comm_create", "pt_comm_create_all", "pt_comm_destroy", "pt_comm_rank", "pt_comm_gather_frame",
    "pt_render_sharded", "pt_shard_pack", "pt_shard_unpack_all", "pt_scene_build_info", "pt_scene_debug_tree", "pt_scene_debug_tree_packed",
    "pt_buffer_register", "pt_buffer_unregister", "pt_last_kernel_choice", "pt_debug_select", "pt_debug_last_kernel_symbols", "pt_debug_cell_grid", "pt_comm_runtime", "pt_last_host_ms", "pt_scene_coop_counters", "pt_scene_debug_tile_rays", "pt_render_sharded_all", "pt_comm_gather_frame_all", "pt_closest_hit",
]
COMM_ID_BYTES = 128

_lib = None


class PtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ptgpu error %d: %s" % (code, msg))
        self.code = code


def hip_runtimes_mapped():
    """Paths of the libamdhip64 copies mapped into this process (Linux: /proc/self/maps), in the order they appear."""
    seen = []
    try:
        for line in open("/proc/self/maps"):
            path = line.rsplit(None, 1)[-1] if "/" in line else ""
            if os.path.basename(path).startswith("libamdhip64") and os.path.realpath(path) not in [os.path.realpath(q) for q in seen]:
                seen.append(path)
    except OSError:
        pass
    return seen


def _hip_runtime_conflict(paths, torch_initialised):
    """None when the HIP runtimes mapped into the process can work together, else the message to raise. PyTorch's wheel carries its own
    libamdhip64 and libptgpu.so links the system's: both work in one process only if torch's runtime initialises the GPU FIRST -- the other
    way round torch later reports "No HIP GPUs are available", far from the cause."""
    if len(paths) < 2 or torch_initialised:
        return None
    return ("two HIP runtimes are mapped into this process (%s) and torch.cuda has not been initialised yet: libptgpu.so would initialise "
            "the GPU through the system's libamdhip64 first, after which torch's own copy reports \"No HIP GPUs are available\". "
            "`import torch; torch.cuda.init()` BEFORE the first ptgpu call (bench.py and tests/conftest.py do), or set "
            "PTGPU_ALLOW_SECOND_HIP=1 if this process never touches torch.cuda." % ", ".join(paths))


def check_hip_runtimes():
    """Raises RuntimeError for the one load order of two HIP runtimes that cannot work (see _hip_runtime_conflict); called by lib()."""
    import sys
    if os.environ.get("PTGPU_ALLOW_SECOND_HIP") == "1":
        return
    torch = sys.modules.get("torch")
    if torch is None:
        return   # (only a process that imported torch has its copy mapped; a torch imported LATER cannot be seen from here)
    try:
        initialised = bool(torch.cuda.is_initialized())
        gpus = int(torch.cuda.device_count())   # (counting devices does not initialise the GPU)
    except Exception:
        return
    if gpus == 0:
        return   # no GPU in this machine: neither runtime will initialise anything (the CPU test runs)
    msg = _hip_runtime_conflict(hip_runtimes_mapped(), initialised)
    if msg:
        raise RuntimeError(msg)


def lib():
    """Load libptgpu.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        check_hip_runtimes()   # (after the load: the system's libamdhip64 is mapped now, nothing has touched the GPU through it yet)
        vp = C.c_void_p
        L.pt_device_count.argtypes = [C.POINTER(C.c_int)]
        L.pt_scene_create.argtypes = [C.POINTER(PtSceneDesc), C.c_int, C.POINTER(vp)]
        L.pt_scene_create_world.argtypes = [C.POINTER(PtWorldDesc), C.c_int, C.POINTER(vp)]
        L.pt_scene_prepare.argtypes = [vp, C.POINTER(PtParams)]
        L.pt_scene_destroy.argtypes = [vp]
        L.pt_scene_destroy.restype = None
        L.pt_render.argtypes = [vp, C.POINTER(PtParams), C.POINTER(PtCamera), C.c_uint32, vp, C.POINTER(C.c_uint64)]
        L.pt_render_device.argtypes = [vp, C.POINTER(PtParams), C.POINTER(PtCamera), C.c_uint32, vp, vp, vp]
        L.pt_render_shard_device.argtypes = [vp, C.POINTER(PtParams), C.POINTER(PtCamera), C.c_uint32, C.c_uint32,
                                             C.c_uint32, vp, vp, vp]
        L.pt_shard_rows.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.pt_shard_rows.restype = C.c_uint32
        L.pt_scene_set_seed_base.argtypes = [vp, C.c_uint64]
        L.pt_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.pt_last_launch_info.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.pt_scene_set_tuning.argtypes = [vp, C.c_uint32, C.c_uint32]
        L.pt_selftest_probe.argtypes = [C.c_int, C.c_uint32, vp, vp, C.c_size_t]
        L.pt_scene_debug_counters.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
        L.pt_scene_traversal_counters.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
        L.pt_scene_coop_counters.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
        L.pt_scene_debug_tile_rays.argtypes = [vp, vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.pt_closest_hit.argtypes = [vp, C.c_uint32, C.c_uint32, vp, C.c_float, C.c_float, vp, vp]
        L.pt_last_pass_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.pt_comm_unique_id.argtypes = [vp]
        L.pt_comm_create.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(vp)]
        L.pt_comm_create_all.argtypes = [C.POINTER(C.c_int), C.c_uint32, C.POINTER(vp)]
        L.pt_comm_destroy.argtypes = [vp]
        L.pt_comm_destroy.restype = None
        L.pt_comm_rank.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.pt_comm_gather_frame.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp, vp, C.c_int, vp]
        L.pt_render_sharded.argtypes = [vp, vp, C.POINTER(PtParams), C.POINTER(PtCamera), C.c_uint32, vp, vp, C.c_int, vp]
        L.pt_shard_pack.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]
        L.pt_render_sharded_all.argtypes = [C.POINTER(vp), C.POINTER(vp), C.c_uint32, C.POINTER(PtParams), C.POINTER(PtCamera), C.c_uint32, C.POINTER(vp), C.POINTER(vp), C.c_int, C.POINTER(vp)]
        L.pt_comm_gather_frame_all.argtypes = [C.POINTER(vp), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_int, C.POINTER(vp)]
        L.pt_shard_unpack_all.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp]
        L.pt_scene_build_info.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.pt_scene_debug_tree.argtypes = [vp, vp, C.c_size_t]
        L.pt_scene_debug_tree_packed.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_uint32)]
        L.pt_buffer_register.argtypes = [vp, C.c_size_t]
        L.pt_buffer_unregister.argtypes = [vp]
        L.pt_last_kernel_choice.argtypes = [vp, C.POINTER(PtKernelChoice)]
        L.pt_debug_select.argtypes = [C.POINTER(PtSceneDesc), C.POINTER(PtWorldDesc), C.POINTER(PtParams), C.POINTER(PtCamera), C.c_uint32, C.c_uint32,
                                      C.c_uint32, C.POINTER(PtKernelChoice)]
        L.pt_debug_cell_grid.argtypes = [C.POINTER(PtSceneDesc), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_uint32)]
        L.pt_comm_runtime.argtypes = [C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
        L.pt_last_host_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.pt_last_error.restype = C.c_char_p
        L.pt_version.restype = C.c_char_p
        _lib = L
    return _lib


def _check(rc):
    if rc != PT_OK:
        raise PtError(rc, lib().pt_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int(0)
    rc = lib().pt_device_count(C.byref(n))
    return n.value if rc == PT_OK else 0


class SceneDesc:
    """Owns the numpy/ctypes storage behind a pt_scene_desc."""

    def __init__(self, spheres, sphere_material, materials, textures, perlin=None, bvh_nodes=None, bvh_root=-1,
                 sky=None):
        self.spheres = np.ascontiguousarray(spheres, dtype=np.float32).reshape(-1, 4)
        self.sphere_material = np.ascontiguousarray(sphere_material, dtype=np.uint32)
        self.materials = (PtMaterial * max(1, len(materials)))()
        for i, (kind, albedo, param, tex) in enumerate(materials):
            m = self.materials[i]
            m.kind, m.param, m.texture = int(kind), float(param), int(tex)
            m.albedo[:] = [float(a) for a in albedo]
        self.n_materials = len(materials)
        self.textures = (PtTexture * max(1, len(textures)))()
        for i, (kind, color, odd, even, scale) in enumerate(textures):
            t = self.textures[i]
            t.kind, t.odd, t.even, t.scale = int(kind), int(odd), int(even), float(scale)
            t.color[:] = [float(c) for c in color]
        self.n_textures = len(textures)
        self.perlin = None
        if perlin is not None:
            randvec, px, py, pz = perlin
            self.perlin = PtPerlin()
            rv = np.ascontiguousarray(randvec, dtype=np.float32).reshape(256, 3)
            C.memmove(C.addressof(self.perlin.randvec), rv.ctypes.data, 256 * 12)
            for name, arr in (("perm_x", px), ("perm_y", py), ("perm_z", pz)):
                a = np.ascontiguousarray(arr, dtype=np.uint32)
                C.memmove(C.addressof(getattr(self.perlin, name)), a.ctypes.data, 1024)
        self.bvh_nodes = None
        self.n_bvh_nodes = 0
        if bvh_nodes is not None and len(bvh_nodes[0]):
            minmax, lr = bvh_nodes
            minmax = np.ascontiguousarray(minmax, dtype=np.float32).reshape(-1, 6)
            lr = np.ascontiguousarray(lr, dtype=np.int32).reshape(-1, 2)
            self.n_bvh_nodes = len(minmax)
            self.bvh_nodes = (PtBvhNode * self.n_bvh_nodes)()
            packed = np.zeros((self.n_bvh_nodes, 8), dtype=np.float32)
            packed[:, :6] = minmax
            packed[:, 6:] = lr.view(np.float32)
            C.memmove(C.addressof(self.bvh_nodes), packed.ctypes.data, self.n_bvh_nodes * 32)
        self.bvh_root = int(bvh_root)
        self.sky = sky

    def struct(self):
        d = PtSceneDesc()
        d.n_spheres = len(self.spheres)
        d.spheres = C.cast(self.spheres.ctypes.data, C.POINTER(PtSphere))
        d.sphere_material = C.cast(self.sphere_material.ctypes.data, C.POINTER(C.c_uint32))
        d.n_materials = self.n_materials
        d.materials = C.cast(self.materials, C.POINTER(PtMaterial))
        d.n_textures = self.n_textures
        d.textures = C.cast(self.textures, C.POINTER(PtTexture))
        d.perlin = C.pointer(self.perlin) if self.perlin is not None else None
        d.n_bvh_nodes = self.n_bvh_nodes
        d.bvh_nodes = C.cast(self.bvh_nodes, C.POINTER(PtBvhNode)) if self.bvh_nodes is not None else None
        d.bvh_root = self.bvh_root if self.n_bvh_nodes else -1
        d.has_sky = 1 if self.sky is not None else 0
        if self.sky is not None:
            d.sky[:] = [float(c) for c in self.sky]
        return d


class WorldDesc(SceneDesc):
    """Owns the storage behind a pt_world_desc: `hitables` is an [n, 16] uint32 array of 64-byte pt_hitable records,
    `transforms` an [m, 24] float32 array of pt_affine (Affine3A, inverse); the tables are SceneDesc's."""

    def __init__(self, hitables, transforms, materials, textures, perlin=None, bvh_nodes=None, bvh_root=-1, sky=None,
                 images=(), nodes=None, node_children=(), root_node=0):
        super().__init__(np.zeros((0, 4), np.float32), np.zeros(0, np.uint32), materials, textures, perlin=perlin,
                         bvh_nodes=bvh_nodes, bvh_root=bvh_root, sky=sky)
        self.hitables = np.ascontiguousarray(hitables, dtype=np.uint32).reshape(-1, 16)
        self.transforms = np.ascontiguousarray(transforms, dtype=np.float32).reshape(-1, 24)
        # optional scene graph (include/ptgpu.h pt_node): [n, 4] uint32 rows (kind, a, b, density as float bits)
        self.nodes = None if nodes is None else np.ascontiguousarray(nodes, dtype=np.uint32).reshape(-1, 4)
        self.node_children = np.ascontiguousarray(node_children, dtype=np.uint32).reshape(-1)
        self.root_node = int(root_node)
        self.image_arrays = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]   # [H, W, 3] each
        self.images = (PtImage * max(1, len(self.image_arrays)))()
        for i, im in enumerate(self.image_arrays):
            self.images[i].width, self.images[i].height = im.shape[1], im.shape[0]
            self.images[i].rgb = C.cast(im.ctypes.data, C.POINTER(C.c_uint8))

    def struct(self):
        d = PtWorldDesc()
        d.n_hitables = len(self.hitables)
        d.hitables = C.cast(self.hitables.ctypes.data, C.POINTER(PtHitable))
        d.n_transforms = len(self.transforms)
        d.transforms = C.cast(self.transforms.ctypes.data, C.POINTER(PtAffine)) if len(self.transforms) else None
        d.n_materials = self.n_materials
        d.materials = C.cast(self.materials, C.POINTER(PtMaterial))
        d.n_textures = self.n_textures
        d.textures = C.cast(self.textures, C.POINTER(PtTexture))
        d.perlin = C.pointer(self.perlin) if self.perlin is not None else None
        d.n_bvh_nodes = self.n_bvh_nodes
        d.bvh_nodes = C.cast(self.bvh_nodes, C.POINTER(PtBvhNode)) if self.bvh_nodes is not None else None
        d.bvh_root = self.bvh_root if self.n_bvh_nodes else -1
        d.has_sky = 1 if self.sky is not None else 0
        if self.sky is not None:
            d.sky[:] = [float(c) for c in self.sky]
        d.n_images = len(self.image_arrays)
        d.images = C.cast(self.images, C.POINTER(PtImage)) if self.image_arrays else None
        if self.nodes is not None:
            d.n_nodes, d.nodes = len(self.nodes), self.nodes.ctypes.data
            d.n_node_children = len(self.node_children)
            d.node_children = C.cast(self.node_children.ctypes.data, C.POINTER(C.c_uint32)) if len(self.node_children) else None
            d.root_node = self.root_node
        return d


class Scene:
    """pt_scene handle (Scene::new, scene.rs:25-31)."""

    def __init__(self, desc, device=0):
        self._h = C.c_void_p()
        self._desc = desc  # keep storage alive during create
        d = desc.struct()
        create = lib().pt_scene_create_world if isinstance(desc, WorldDesc) else lib().pt_scene_create
        _check(create(C.byref(d), device, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().pt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_tuning(self, blocks_per_cu=0, variant=0):
        _check(lib().pt_scene_set_tuning(self._h, blocks_per_cu, variant))

    def set_seed_base(self, seed):
        _check(lib().pt_scene_set_seed_base(self._h, seed))

    def update(self, params, camera, frame_num, buffer):
        """Scene::update (scene.rs:73-121) on a host numpy buffer [H, W, 3] float32 (in/out)."""
        assert buffer.dtype == np.float32 and buffer.flags["C_CONTIGUOUS"]
        assert buffer.size == params.width * params.height * 3
        rc = C.c_uint64(0)
        _check(lib().pt_render(self._h, C.byref(params), C.byref(camera), frame_num, buffer.ctypes.data, C.byref(rc)))
        return rc.value

    def update_device(self, params, camera, frame_num, d_rgb_ptr, d_ray_count_ptr, stream=0):
        _check(lib().pt_render_device(self._h, C.byref(params), C.byref(camera), frame_num, d_rgb_ptr,
                                      d_ray_count_ptr, stream))

    def update_shard_device(self, params, camera, frame_num, shard_index, shard_count, d_rgb_ptr, d_ray_count_ptr,
                            stream=0):
        _check(lib().pt_render_shard_device(self._h, C.byref(params), C.byref(camera), frame_num, shard_index,
                                            shard_count, d_rgb_ptr, d_ray_count_ptr, stream))

    def debug_counters(self, reset=True):
        out = (C.c_uint64 * 4)()
        _check(lib().pt_scene_debug_counters(self._h, out, 1 if reset else 0))
        return dict(misses=out[0], candidates=out[1], overflows=out[2], exact_positives=out[3])

    def coop_counters(self, reset=True):
        """(pixels handed over to idle waves, rays those waves traced) since the last reset (csrc/pt_coop.h)."""
        out = (C.c_uint64 * 2)()
        _check(lib().pt_scene_coop_counters(self._h, out, 1 if reset else 0))
        return dict(pixels=out[0], rays=out[1])

    def tile_rays(self):
        """pt_scene_debug_tile_rays: rays per 8x8 work tile of the last frame as a (tile rows, tiles per row) uint32 array, row 0 = bottom."""
        n, tx = C.c_uint32(0), C.c_uint32(0)
        _check(lib().pt_scene_debug_tile_rays(self._h, None, 0, C.byref(n), C.byref(tx)))
        out = np.zeros(n.value, np.uint32)
        _check(lib().pt_scene_debug_tile_rays(self._h, out.ctypes.data, n.value, C.byref(n), C.byref(tx)))
        return out.reshape(-1, tx.value)

    def closest_hit(self, mode, n_rays, d_rays7_ptr, d_hits8_ptr, t_min=0.001, t_max=3.4028234663852886e38, stream=0):
        """pt_closest_hit: the reference's closest-hit query (QUERY_* mode) for n_rays explicit rays on the device."""
        _check(lib().pt_closest_hit(self._h, mode, n_rays, d_rays7_ptr, t_min, t_max, d_hits8_ptr, stream))

    def traversal_counters(self, reset=True):
        out = (C.c_uint64 * 2)()
        _check(lib().pt_scene_traversal_counters(self._h, out, 1 if reset else 0))
        return dict(nodes=out[0], sphere_tests=out[1])

    def last_kernel_ms(self):
        ms = C.c_float(0)
        _check(lib().pt_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def last_pass_ms(self):
        """Pilot pass + tile sort + frame kernel of the last render (HIP events on the launch stream)."""
        ms = C.c_float(0)
        _check(lib().pt_last_pass_ms(self._h, C.byref(ms)))
        return ms.value

    def update_sharded(self, comm, params, camera, frame_num, d_rgb_full_ptr, d_ray_count_ptr, root=-1, stream=0):
        """Scene::update over the ranks of `comm` (pt_render_sharded): rows y % world == rank, RCCL gather."""
        _check(lib().pt_render_sharded(self._h, comm._h, C.byref(params), C.byref(camera), frame_num, d_rgb_full_ptr,
                                       d_ray_count_ptr, root, stream))

    def build_info(self):
        """The library's own traversal tree: device build time (ms), node count, depth, built on the device?"""
        ms, n, d, dev = C.c_float(0), C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        _check(lib().pt_scene_build_info(self._h, C.byref(ms), C.byref(n), C.byref(d), C.byref(dev)))
        return dict(build_ms=ms.value, n_nodes=n.value, depth=d.value, on_device=bool(dev.value))

    def debug_tree(self):
        """The 4-wide tree's nodes as a [n_nodes, 32] uint32 array (128-byte records)."""
        n = self.build_info()["n_nodes"]
        out = np.zeros((max(n, 1), 32), np.uint32)
        _check(lib().pt_scene_debug_tree(self._h, out.ctypes.data, out.nbytes))
        return out[:n]

    def debug_tree_packed(self):
        """The nodes as the kernels read them: ([n_nodes, 16] uint32 array of 64-byte records, usable flag)."""
        n = self.build_info()["n_nodes"]
        out = np.zeros((max(n, 1), 16), np.uint32)
        ok = C.c_uint32(0)
        _check(lib().pt_scene_debug_tree_packed(self._h, out.ctypes.data, out.nbytes, C.byref(ok)))
        return out[:n], bool(ok.value)

    def last_host_ms(self):
        """(scan / copy-in, GPU wait, copy-out, whole call) in ms of the last update() on a pageable buffer."""
        out = (C.c_float * 4)()
        _check(lib().pt_last_host_ms(self._h, out))
        return tuple(float(x) for x in out)

    def last_kernel_choice(self):
        """The kernel and geometry the most recent render on this handle used (dict of pt_kernel_choice)."""
        c = PtKernelChoice()
        _check(lib().pt_last_kernel_choice(self._h, C.byref(c)))
        return c.as_dict()

    def last_launch_info(self):
        g, b, l = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        _check(lib().pt_last_launch_info(self._h, C.byref(g), C.byref(b), C.byref(l)))
        return g.value, b.value, l.value


def debug_select(desc, params, camera, shard_count=1, blocks_per_cu=0, variant=0):
    """pt_debug_select: the kernel choice pt_render* would make for a description -- host only, no device needed."""
    c = PtKernelChoice()
    d = desc.struct()
    if isinstance(desc, WorldDesc):
        rc = lib().pt_debug_select(None, C.byref(d), C.byref(params), C.byref(camera), shard_count, blocks_per_cu, variant, C.byref(c))
    else:
        rc = lib().pt_debug_select(C.byref(d), None, C.byref(params), C.byref(camera), shard_count, blocks_per_cu, variant, C.byref(c))
    _check(rc)
    return c.as_dict()


def debug_cell_grid(desc):
    """pt_debug_cell_grid: the uniform cell grid a sphere scene would get (host only) as a dict -- cells per axis `n`, box corner `gmin`, cell
    sizes `h`, `d_build`, `half_diag`, the records (n_records x 5 x 4 uint32) and the list indices of the spheres outside the grid. Raises
    PtError(PT_ERR_UNSUPPORTED) for a scene that walks the tree."""
    info = (C.c_uint32 * 16)()
    d = desc.struct()
    _check(lib().pt_debug_cell_grid(C.byref(d), info, None, 0, None))
    iu = np.frombuffer(info, np.uint32).copy()
    fl = iu.view(np.float32)
    rec = np.zeros((int(iu[3]), 5, 4), np.uint32)
    large = np.zeros(16, np.uint32)
    _check(lib().pt_debug_cell_grid(C.byref(d), info, rec.ctypes.data_as(C.POINTER(C.c_uint32)), rec.shape[0], large.ctypes.data_as(C.POINTER(C.c_uint32))))
    return {"n": iu[:3].astype(int), "gmin": fl[8:11].astype(np.float64), "h": fl[11:14].astype(np.float64), "d_build": float(fl[14]), "half_diag": float(fl[15]), "items_per_cell": float(fl[5]), "occupied": float(fl[6]),
            "records": rec, "large": large[:int(iu[4])].astype(int)}


def comm_runtime():
    """(version code, path) of the RCCL the pt_comm_* functions resolved at run time; raises PtError when there is none."""
    v = C.c_int(0)
    buf = C.create_string_buffer(256)
    _check(lib().pt_comm_runtime(C.byref(v), buf, 256))
    return v.value, buf.value.decode()


class Comm:
    """pt_comm handle: the RCCL communicator of the sharded frame path (one rank per GPU)."""

    def __init__(self, handle, rank, world):
        self._h, self.rank, self.world = handle, rank, world

    @staticmethod
    def unique_id():
        """ncclGetUniqueId as bytes (rank 0 calls it; every rank passes the same bytes to Comm.create)."""
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        _check(lib().pt_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def create(cls, uid, rank, world, device):
        h = C.c_void_p()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(uid)
        _check(lib().pt_comm_create(buf, rank, world, device, C.byref(h)))
        return cls(h, rank, world)

    @classmethod
    def create_all(cls, devices):
        """ncclCommInitAll: one communicator per listed device, all in this process (rank i on devices[i])."""
        n = len(devices)
        devs = (C.c_int * n)(*devices)
        hs = (C.c_void_p * n)()
        _check(lib().pt_comm_create_all(devs, n, hs))
        return [cls(C.c_void_p(hs[i]), i, n) for i in range(n)]

    def gather_frame(self, width, height, d_shard_ptr, d_full_ptr, d_ray_count_ptr, root=-1, stream=0):
        _check(lib().pt_comm_gather_frame(self._h, width, height, d_shard_ptr, d_full_ptr, d_ray_count_ptr, root, stream))

    def close(self):
        if self._h:
            lib().pt_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _ptr_array(values):
    return (C.c_void_p * len(values))(*[v if isinstance(v, (int, type(None))) else v.value for v in values])


def render_sharded_all(scenes, comms, params, camera, frame_num, d_full_ptrs, d_ray_count_ptrs, root=-1, streams=None):
    """pt_render_sharded_all: every rank of a Comm.create_all clique from ONE thread -- all ranks' collectives inside one RCCL group."""
    n = len(comms)
    _check(lib().pt_render_sharded_all(_ptr_array([s._h for s in scenes]), _ptr_array([c._h for c in comms]), n, C.byref(params), C.byref(camera), frame_num,
                                       _ptr_array(d_full_ptrs), _ptr_array(d_ray_count_ptrs), root, _ptr_array(streams) if streams else None))


def gather_frame_all(comms, width, height, d_shard_ptrs, d_full_ptrs, d_ray_count_ptrs, root=-1, streams=None):
    """pt_comm_gather_frame_all: the exchange step alone for every rank of the clique, one RCCL group."""
    n = len(comms)
    _check(lib().pt_comm_gather_frame_all(_ptr_array([c._h for c in comms]), n, width, height, _ptr_array(d_shard_ptrs), _ptr_array(d_full_ptrs),
                                          _ptr_array(d_ray_count_ptrs), root, _ptr_array(streams) if streams else None))


def buffer_register(array):
    """pt_buffer_register on a numpy array the caller keeps alive: pt_render then works in place on it."""
    _check(lib().pt_buffer_register(array.ctypes.data, array.nbytes))


def buffer_unregister(array):
    _check(lib().pt_buffer_unregister(array.ctypes.data))


def shard_pack(d_full_ptr, d_shard_ptr, width, height, shard_index, shard_count, stream=0):
    _check(lib().pt_shard_pack(d_full_ptr, d_shard_ptr, width, height, shard_index, shard_count, stream))


def shard_unpack_all(d_gathered_ptr, d_full_ptr, width, height, shard_count, stream=0):
    _check(lib().pt_shard_unpack_all(d_gathered_ptr, d_full_ptr, width, height, shard_count, stream))


def shard_rows(height, shard_index, shard_count):
    return lib().pt_shard_rows(height, shard_index, shard_count)


PROBE_POW5, PROBE_SIN, PROBE_COS, PROBE_RNG, PROBE_LN, PROBE_SWEEP_SQRT, PROBE_SWEEP_DRAWS, PROBE_SWEEP_INVLEN, PROBE_SWEEP_DIV, PROBE_SWEEP_DIVA, PROBE_SWEEP_RECIP = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10


def selftest_probe(probe, values, device=0):
    a = np.ascontiguousarray(values, dtype=np.float32)
    out = np.zeros_like(a)
    _check(lib().pt_selftest_probe(device, probe, a.ctypes.data, out.ctypes.data, a.size))
    return out
