#!/usr/bin/env python3
"""Child process of tests/test_gpu_parity.py::test_sharded_frames_on_an_rccl_test_double: drives the N > 1 paths of the C ABI
(pt_comm_create_all, pt_render_sharded[_all], pt_render_shard_device + pt_comm_gather_frame[_all]) with N ranks on ONE GPU,
behind tests/mock_rccl/mock_rccl.hip (loaded as librccl.so.1 through LD_LIBRARY_PATH: this process must never import torch,
whose own librccl would win). Compares every receiving rank's frame and every rank's ray count with the unsharded render
(scene.rs:90-93, 118-120), over progressive frames. Prints one JSON line."""
import argparse
import ctypes as C
import importlib.util
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class Hip:
    def __init__(self):
        self.L = C.CDLL("libamdhip64.so")
        self.L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.L.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.L.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]

    def ok(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: hipError %d" % (what, rc))

    def zeros(self, nbytes):
        p = C.c_void_p()
        self.ok(self.L.hipMalloc(C.byref(p), max(nbytes, 8)), "hipMalloc")
        self.ok(self.L.hipMemset(p, 0, max(nbytes, 8)), "hipMemset")
        return p

    def to_host(self, p, shape, dtype):
        out = np.zeros(shape, dtype)
        self.ok(self.L.hipMemcpy(out.ctypes.data, p, out.nbytes, 2), "hipMemcpy D2H")
        return out

    def from_host(self, p, arr):
        self.ok(self.L.hipMemcpy(p, arr.ctypes.data, arr.nbytes, 1), "hipMemcpy H2D")

    def sync(self):
        self.ok(self.L.hipDeviceSynchronize(), "hipDeviceSynchronize")

    def stream(self):
        s = C.c_void_p()
        # non-blocking: all ranks share ONE device here, and a NULL-stream operation issued for one rank (a hipMemset while its
        # buffers are sized) must not wait for another rank's stream that is parked inside a collective
        self.ok(self.L.hipStreamCreateWithFlags(C.byref(s), 1), "hipStreamCreateWithFlags")
        return s


def per_rank(fn, n):
    """fn(r) on one thread per rank (ctypes releases the GIL inside the ABI calls); re-raises the first failure."""
    errors = []

    def run(r):
        try:
            fn(r)
        except Exception as e:   # noqa: BLE001
            errors.append((r, e))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise RuntimeError("rank %d: %s" % errors[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--root", type=int, default=-1)
    ap.add_argument("--width", type=int, default=96)
    ap.add_argument("--height", type=int, default=50)
    ap.add_argument("--samples", type=int, default=4)
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--mode", choices=["sharded", "sharded_all", "gather", "gather_all"], default="sharded_all")
    ap.add_argument("--preset", default="random_spheres")
    ap.add_argument("--variant", type=int, default=0, help="pt_scene_set_tuning bits for every rank's scene")
    ap.add_argument("--null-stale", action="store_true", help="ranks that do not receive the frame pass NULL for it from frame 1 on")
    a = ap.parse_args()
    import faulthandler
    faulthandler.dump_traceback_later(40, exit=True)   # a hang shows where every thread is
    assert "torch" not in sys.modules
    ptgpu = _load("pathtrace_rs_amd_ptgpu", "pathtrace-rs_amd/ptgpu.py")
    pthost = _load("pathtrace_rs_amd_pthost", "pathtrace-rs_amd/pthost.py")
    hip = Hip()
    N, W, H, S = a.ranks, a.width, a.height, a.samples
    version, path = ptgpu.comm_runtime()
    p = ptgpu.PtParams(W, H, S, 10, 0, 0)
    # the unsharded reference: progressive frames into one buffer
    ref_scene = pthost.HostScene(a.preset, W, H, samples=S, device=0)
    d_ref, d_ref_rc = hip.zeros(W * H * 12), hip.zeros(8)
    scenes = [pthost.HostScene(a.preset, W, H, samples=S, device=0) for _ in range(N)]
    for sc in scenes:
        sc.device_scene().set_tuning(0, a.variant)
    comms = ptgpu.Comm.create_all([0] * N)
    fulls = [hip.zeros(W * H * 12) for _ in range(N)]
    rcs = [hip.zeros(8) for _ in range(N)]
    shards = [hip.zeros(ptgpu.shard_rows(H, r, N) * W * 12) for r in range(N)]
    streams = [hip.stream() for _ in range(N)]
    checks, mock = 0, C.CDLL(path)
    for frame in range(a.frames):
        ref_scene.device_scene().update_device(p, ref_scene.camera, frame, d_ref, d_ref_rc, 0)
        hip.sync()
        ref = hip.to_host(d_ref, (H, W, 3), np.float32)
        ref_rays = int(hip.to_host(d_ref_rc, (1,), np.uint64)[0])
        receives = [a.root < 0 or a.root == r for r in range(N)]
        full_args = [fulls[r] if (receives[r] or frame == 0 or not a.null_stale) else None for r in range(N)]
        devs = [s.device_scene() for s in scenes]
        if a.mode == "sharded_all":
            ptgpu.render_sharded_all(devs, comms, p, ref_scene.camera, frame, full_args, rcs, a.root, streams)
        elif a.mode == "sharded":
            # the one-rank form: ONE THREAD PER COMMUNICATOR, as include/ptgpu.h requires of a process that drives several ranks
            per_rank(lambda r: devs[r].update_sharded(comms[r], p, ref_scene.camera, frame, full_args[r], rcs[r], a.root, streams[r]), N)
        else:
            for r in range(N):   # the caller's own buffers: previous rows packed by the ABI's kernel, rendered, then gathered
                ptgpu.shard_pack(fulls[r], shards[r], W, H, r, N, streams[r])
                devs[r].update_shard_device(p, ref_scene.camera, frame, r, N, shards[r], rcs[r], streams[r])
            if a.mode == "gather_all":
                ptgpu.gather_frame_all(comms, W, H, shards, full_args, rcs, a.root, streams)
            else:
                per_rank(lambda r: comms[r].gather_frame(W, H, shards[r], full_args[r], rcs[r], a.root, streams[r]), N)
        hip.sync()
        assert mock.mock_rccl_group_depth() == 0, "an RCCL group was left open"
        for r in range(N):
            rays = int(hip.to_host(rcs[r], (1,), np.uint64)[0])
            assert rays == ref_rays, "frame %d rank %d: ray count %d, unsharded %d" % (frame, r, rays, ref_rays)
            checks += 1
            if receives[r]:
                got = hip.to_host(fulls[r], (H, W, 3), np.float32)
                wrong = (got != ref).any(axis=-1)
                bad = int(wrong.sum())
                by_owner = {q: int(wrong[q::N].sum()) for q in range(N) if wrong[q::N].any()}
                assert bad == 0, "frame %d rank %d: %d pixels differ from the unsharded frame (by owning rank: %r; %d of them are all-zero)" % (
                    frame, r, bad, by_owner, int((wrong & (got == 0).all(axis=-1)).sum()))
                checks += 1
        if a.mode.startswith("gather") and a.root >= 0:
            # the gather modes blend against each rank's own full buffer: ranks that did not receive the frame need it for the next one
            for r in range(N):
                if not receives[r]:
                    hip.from_host(fulls[r], ref)
    print(json.dumps({"ok": True, "checks": checks, "rccl_version": version, "rccl_path": path, "ranks": N, "root": a.root, "mode": a.mode}))


if __name__ == "__main__":
    main()
