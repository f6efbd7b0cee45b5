#!/usr/bin/env python3
"""The reference's OWN benchmark units, on both sides. pathtrace-rs measures exactly six things (`cargo bench --features bench`):
one closest-hit query on the centre ray of a 200x100 preset (src/bench.rs:8-26),
    hitable_list::bench::ray_hit                HitableList::ray_hit, random_spheres      (collision/hitable_list.rs:68-75)
    spheres_soa::bench::ray_hit_scalar / _sse4_1 / _avx2     SpheresSoA, random_spheres  (collision/spheres_soa.rs:464-485)
    bvh::bench::random_spheres_ray_hit          BVHNode::ray_hit, random_spheres          (collision/bvh.rs:361-369)
    bvh::bench::ray_hit                         BVHNode::ray_hit, random (moving spheres) (collision/bvh.rs:371-379)
This tool times the same six queries (a) on the oracle -- the C restatement of those functions, ONE thread, ns per call as `b.iter`
reports -- and (b) on the GPU through pt_closest_hit (csrc/pt_query.hip: the same algorithms as written, one ray per lane) as a batch
of identical rays, ns per query = batch time / batch size (HIP events). The fixture ray is the bench's: Params{200, 100}, rng seed 0
continued after the scene build, camera.get_ray(0.5, 0.5). (The bench builds its BVH AFTER drawing the ray; the oracle scene builds
it before: the tree's random split axes differ, the result and the order of magnitude do not.)
Prints one JSON object; --out writes it to a file as well."""
import argparse
import importlib.util
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def fixture_ray(ob, osc):
    """camera.get_ray(0.5, 0.5, rng) with the rng the scene build left behind (bench.rs:22-24)."""
    st = np.zeros(4, np.uint64)
    ob.lib().ora_xoshiro_seed_from_u64(0, st.ctypes.data)
    ex = osc.export()
    for _ in range(int(ex["build_draws"])):   # (a list world's build draws f32s only: one next_u64 each)
        ob.lib().ora_xoshiro_next_u64(st.ctypes.data)
    cam = ex["camera"]
    out = np.zeros(7, np.float32)
    ob.lib().ora_camera_get_ray(cam.ctypes.data, 0.5, 0.5, st.ctypes.data, out.ctypes.data)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1 << 20, help="identical rays per GPU launch")
    ap.add_argument("--reps", type=int, default=200000, help="oracle calls per unit")
    ap.add_argument("--no-gpu", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    units = [("hitable_list::ray_hit", "random_spheres", False, 0, "QUERY_LIST"),
             ("spheres_soa::ray_hit_scalar", "random_spheres", False, 1, "QUERY_SOA_SCALAR"),
             ("spheres_soa::ray_hit_sse4_1", "random_spheres", False, 4, "QUERY_SOA_SSE4_1"),
             ("spheres_soa::ray_hit_avx2", "random_spheres", False, 8, "QUERY_SOA_AVX2"),
             ("bvh::random_spheres_ray_hit", "random_spheres", True, 0, "QUERY_BVH"),
             ("bvh::ray_hit", "random", True, 0, "QUERY_BVH")]
    native = ob.lib(ob.build_native())
    res = {"fixture": "Params 200x100, rng seed 0 continued after the scene build, camera.get_ray(0.5, 0.5) (src/bench.rs:8-26)", "units": []}
    gpu = None
    if not a.no_gpu:
        import torch
        gpu = (torch, _load("pathtrace_rs_amd_ptgpu", "pathtrace-rs_amd/ptgpu.py"), _load("pathtrace_rs_amd_pthost", "pathtrace-rs_amd/pthost.py"))
    for name, preset, bvh, which, mode in units:
        osc = ob.OracleScene(preset, 200, 100, use_bvh=bvh, library=native)
        ray = fixture_ray(ob, ob.OracleScene(preset, 200, 100, use_bvh=False, library=native))   # (the bench draws its ray before it builds the BVH)
        hit = osc.world_ray_hit(ray[0:3], ray[3:6], float(ray[6])) if which == 0 else osc.soa_ray_hit(which, ray[0:3], ray[3:6])
        unit = {"bench": name, "preset": preset, "entries": int(len(osc.export()["hitables"])),
                "hit": None if hit is None else {"t": float(hit[0]), "entry": int(hit[1])},
                "oracle_ns_per_query": osc.bench_ray_hit(which, ray[0:3], ray[3:6], float(ray[6]), a.reps)}
        if gpu:
            torch, ptgpu, pthost = gpu
            hs = pthost.HostScene(preset, 200, 100, samples=1, use_bvh=bvh, device=0)
            rays = torch.from_numpy(np.tile(ray, (a.batch, 1)).astype(np.float32)).cuda()
            hits = torch.zeros((a.batch, 8), dtype=torch.float32, device="cuda")
            sc, m = hs.device_scene(), getattr(ptgpu, mode)
            stream = torch.cuda.current_stream().cuda_stream
            sc.closest_hit(m, a.batch, rays.data_ptr(), hits.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            best = 1e30
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                sc.closest_hit(m, a.batch, rays.data_ptr(), hits.data_ptr(), stream=stream)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            h0 = hits[0].cpu().numpy()
            entry = int(h0[1:2].view(np.uint32)[0])
            unit["gpu_ns_per_query"] = best * 1e6 / a.batch
            unit["gpu_batch"] = a.batch
            unit["gpu_equals_oracle"] = bool((hit is None and entry == 0xffffffff) or (hit is not None and entry == hit[1] and float(h0[0]) == float(hit[0])))
        res["units"].append(unit)
    res["oracle"] = "oracle/ptref.c -O3 -march=native -ffp-contract=off, one thread"
    line = json.dumps(res)
    print(line)
    if a.out:
        open(a.out, "w").write(json.dumps(res, indent=1) + "\n")


if __name__ == "__main__":
    main()
