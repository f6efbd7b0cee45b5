#!/usr/bin/env python3
"""Writes tests/golden/rust_expectations.json: what `cargo run --release -- -O -P <preset> -W .. -H .. -S .. [-B]` of
bitshifter/pathtrace-rs must print and write IF the oracle's restatement (oracle/ptref.c) is faithful -- the "{}rays" count of
offline.rs:36-41, the SHA-256 of the decoded RGB8 pixels of output.png (offline.rs:43-59, rows top-down) and the SHA-256 of the raw
f32 frame buffer (row 0 = bottom, as Scene::update leaves it; needs tools/pin/offline_dump_f32.patch on the Rust side).
Generated HERE from the oracle (CPU); tests/ check that the product (HIP) reproduces every entry. Nothing in this file has been
compared with the Rust binary yet: that is what tools/pin_against_rust.py is for."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob   # noqa: E402

CASES = [   # preset, W, H, S, use_bvh
    ("small", 200, 100, 4, False), ("small", 200, 100, 4, True),                      # BASELINE config 1
    ("random_spheres", 120, 80, 4, False), ("random_spheres", 120, 80, 4, True),
    ("random_spheres", 1200, 800, 64, False),                                          # BASELINE config 3: the metric's frame
    ("two_perlin_spheres", 160, 90, 8, False),                                         # Noise textures: colour passes through f32::sin
    ("random", 300, 200, 8, False),                                                    # MovingSphere
    ("cornell_smoke", 300, 200, 16, False), ("cornell_smoke", 300, 200, 16, True),     # rects, instances, constant media (f32::ln)
    ("simple_light", 160, 90, 8, True),
]


def entry(lib, preset, W, H, S, bvh):
    sc = ob.OracleScene(preset, W, H, use_bvh=bvh, library=lib)
    buf, rays = sc.update(S, 10, 0)
    rgb8 = np.zeros((H, W, 3), np.uint8)
    lib.ora_frame_to_srgb8(buf.ctypes.data, W, H, rgb8.ctypes.data)
    return {"args": ["-O", "-P", preset, "-W", str(W), "-H", str(H), "-S", str(S)] + (["-B"] if bvh else []),
            "rays": int(rays), "rgb8_sha256": hashlib.sha256(rgb8.tobytes()).hexdigest(), "f32_sha256": hashlib.sha256(buf.tobytes()).hexdigest(),
            "libm_sensitive": preset in ("two_perlin_spheres", "simple_light", "cornell_smoke")}


def main():
    lib = ob.lib(ob.build_native())
    out = {"reference": "bitshifter/pathtrace-rs 0.1.2, `cargo run --release -- <args>` (depth 10, fixed seed)",
           "generated_by": "tools/pin/make_rust_expectations.py from oracle/ptref.c (x86-64, glibc, -ffp-contract=off)",
           "status": "NOT yet compared with the Rust binary",
           "cases": [entry(lib, *c) for c in CASES]}
    path = os.path.join(ROOT, "tests", "golden", "rust_expectations.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, "with", len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
