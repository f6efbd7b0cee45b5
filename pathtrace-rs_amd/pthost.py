"""ctypes binding of libpthost.so (host/pthost_c.h): the C++ host that mirrors the
reference's presets / Camera / Params / new_scene / render_offline. Plumbing only."""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PTGPU_BUILD_DIR (development): a build directory other than the shipped _build -- `make -C pathtrace-rs_amd B=_build_dev DEFS=...` puts a
# complete set (libptgpu.so + libpthost.so + CLI, linked to each other by $ORIGIN) there, so A/B runs never overwrite the product.
_BUILD = os.environ.get("PTGPU_BUILD_DIR") or "_build"
LIB_PATH = os.path.join(_BUILD if os.path.isabs(_BUILD) else os.path.join(_HERE, _BUILD), "libpthost.so")

if __package__:
    from . import ptgpu  # noqa: F401
else:  # loaded by file path (the package directory name is not an identifier)
    import importlib.util as _u
    _name = "pathtrace_rs_amd_ptgpu"
    if _name in sys.modules:
        ptgpu = sys.modules[_name]
    else:
        _spec = _u.spec_from_file_location(_name, os.path.join(_HERE, "ptgpu.py"))
        ptgpu = _u.module_from_spec(_spec)
        sys.modules[_name] = ptgpu
        _spec.loader.exec_module(ptgpu)

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError("%s not found: run __graft_entry__.build()" % LIB_PATH)
        ptgpu.lib()  # libpthost links libptgpu (rpath $ORIGIN); load it first so failures are explicit
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.pth_scene_build.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(vp)]
        L.pth_scene_free.argtypes = [vp]
        L.pth_scene_free.restype = None
        L.pth_scene_desc.argtypes = [vp]
        L.pth_scene_desc.restype = C.POINTER(ptgpu.PtSceneDesc)
        L.pth_scene_world_desc.argtypes = [vp]
        L.pth_scene_world_desc.restype = C.POINTER(ptgpu.PtWorldDesc)
        L.pth_scene_is_world.argtypes = [vp]
        L.pth_scene_camera.argtypes = [vp]
        L.pth_scene_camera.restype = C.POINTER(ptgpu.PtCamera)
        L.pth_scene_handle.argtypes = [vp]
        L.pth_scene_handle.restype = vp
        L.pth_scene_build_draws.argtypes = [vp]
        L.pth_scene_build_draws.restype = C.c_uint64
        L.pth_scene_bvh_depth.argtypes = [vp]
        L.pth_scene_bvh_depth.restype = C.c_uint32
        L.pth_render_offline.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                         C.c_int, C.c_char_p, C.c_uint32]
        L.pth_linear_to_srgb.argtypes = [vp, vp]
        L.pth_save_png.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32]
        L.pth_last_error.restype = C.c_char_p
        _lib = L
    return _lib


class HostScene:
    """offline.rs:16-24 on the host: seed-0 rng -> Storage::new -> presets::from_name -> new_scene.

    device=None builds the description only (CPU tests); device>=0 uploads it through the C ABI.
    """

    def __init__(self, preset, width, height, samples=1, use_bvh=False, device=None, quiet=True):
        self._h = C.c_void_p()
        rc = lib().pth_scene_build(preset.encode(), width, height, samples, 1 if use_bvh else 0,
                                   -1 if device is None else int(device), 1 if quiet else 0, C.byref(self._h))
        if rc == 2:
            raise KeyError("unrecognised preset %r" % preset)
        if rc != 0:
            raise RuntimeError(lib().pth_last_error().decode("utf-8", "replace"))
        self.preset, self.width, self.height, self.use_bvh = preset, width, height, bool(use_bvh)

    def close(self):
        if self._h:
            lib().pth_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def desc(self):
        return lib().pth_scene_desc(self._h).contents

    @property
    def world_desc(self):
        return lib().pth_scene_world_desc(self._h).contents

    @property
    def is_world(self):
        """True when the world has non-sphere hitables (traced by the general kernel)."""
        return bool(lib().pth_scene_is_world(self._h))

    @property
    def camera(self):
        return lib().pth_scene_camera(self._h).contents

    @property
    def build_draws(self):
        return lib().pth_scene_build_draws(self._h)

    @property
    def bvh_depth(self):
        return lib().pth_scene_bvh_depth(self._h)

    def device_scene(self):
        """Borrow the device handle as a ptgpu.Scene-like object (not owning)."""
        h = lib().pth_scene_handle(self._h)
        if not h:
            raise RuntimeError("scene was built without a device")
        s = ptgpu.Scene.__new__(ptgpu.Scene)
        s._h = C.c_void_p(h)
        s._desc = None
        s.close = lambda: None  # owned by the host scene
        s._owner = self
        return s

    # numpy views of the flattened description (copies) -- for cross-checks
    def export(self):
        d = self.world_desc
        n = d.n_hitables
        records = np.zeros((0, 16), np.uint32)
        if n:
            records = np.ctypeslib.as_array(C.cast(d.hitables, C.POINTER(C.c_uint32)), shape=(n, 16)).copy()
        transforms = np.zeros((d.n_transforms, 24), np.float32)
        if d.n_transforms:
            transforms = np.ctypeslib.as_array(C.cast(d.transforms, C.POINTER(C.c_float)), shape=(d.n_transforms, 24)).copy()
        sph = mid = None
        if not self.is_world:
            sd = self.desc
            sph = np.ctypeslib.as_array(C.cast(sd.spheres, C.POINTER(C.c_float)), shape=(n, 4)).copy()
            mid = np.ctypeslib.as_array(sd.sphere_material, shape=(n,)).copy()
        mats = np.zeros((d.n_materials, 6), np.float32)
        for i in range(d.n_materials):
            m = d.materials[i]
            mats[i] = [m.kind, m.albedo[0], m.albedo[1], m.albedo[2], m.param, m.texture]
        texs = np.zeros((d.n_textures, 7), np.float32)
        for i in range(d.n_textures):
            t = d.textures[i]
            texs[i] = [t.kind, t.color[0], t.color[1], t.color[2], t.odd, t.even, t.scale]
        perlin = None
        if d.perlin:
            p = d.perlin.contents
            perlin = (np.ctypeslib.as_array(p.randvec).reshape(256, 3).copy(), np.ctypeslib.as_array(p.perm_x).copy(),
                      np.ctypeslib.as_array(p.perm_y).copy(), np.ctypeslib.as_array(p.perm_z).copy())
        nn = d.n_bvh_nodes
        minmax = np.zeros((nn, 6), np.float32)
        lr = np.zeros((nn, 2), np.int32)
        if nn:
            raw = np.ctypeslib.as_array(C.cast(d.bvh_nodes, C.POINTER(C.c_float)), shape=(nn, 8)).copy()
            minmax = raw[:, :6].copy()
            lr = raw[:, 6:].copy().view(np.int32)
        cam = np.ctypeslib.as_array(C.cast(C.pointer(self.camera), C.POINTER(C.c_float)), shape=(24,)).copy()
        return dict(hitables=records, transforms=transforms, spheres=sph, sphere_material=mid, materials=mats, textures=texs, perlin=perlin, bvh_minmax=minmax,
                    bvh_children=lr, bvh_root=d.bvh_root, camera=cam,
                    sky=(np.array(list(d.sky), np.float32) if d.has_sky else None), build_draws=self.build_draws)


def render_offline(preset, width, height, samples, max_depth=10, use_bvh=False, random_seed=False, device=0,
                   output="output.png", frames=1):
    return lib().pth_render_offline(preset.encode(), width, height, samples, max_depth, 1 if use_bvh else 0,
                                    1 if random_seed else 0, device, output.encode(), frames)


def linear_to_srgb(rgb):
    a = np.ascontiguousarray(rgb, np.float32)
    out = np.zeros(3, np.uint8)
    lib().pth_linear_to_srgb(a.ctypes.data, out.ctypes.data)
    return out


def save_png(path, buffer, width, height):
    a = np.ascontiguousarray(buffer, np.float32)
    return lib().pth_save_png(path.encode(), a.ctypes.data, width, height)
