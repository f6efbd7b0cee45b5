#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.

Two kinds of fixture:
 * rng_kat.json  -- EXTERNAL pins: the public xoshiro256+ reference vector (state [1,2,3,4]) and
                    SplitMix64(0) outputs, plus survey-derived values (not produced by the
                    reference binary, which cannot be built here): seed_from_u64 f32 streams, pixel seeds,
                    the random_spheres scene-build ledger.
 * *.npz         -- frames / crops rendered by the CPU oracle (oracle/ptref.c) at the BASELINE.json
                    configs, so the GPU box can check HIP output without re-deriving anything and so
                    any drift of the oracle itself is caught by the CPU suite.
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding as ob  # noqa: E402

RNG_KAT = {
    "source": "public xoshiro256+ reference vector and SplitMix64 outputs; remaining entries survey-derived "
              "(SURVEY.md 8c), NOT produced by the Rust binary",
    "xoshiro256plus_state_1_2_3_4_next_u64": [
        5, 211106232532999, 211106635186183, 9223759065350669058, 9250833439874351877, 13862484359527728515,
        2346507365006083650, 1168864526675804870, 34095955243042024, 3466914240207415127],
    "splitmix64_seed0": ["e220a8397b1dcdaf", "6e789e6aa1b965f4", "06c45d188009454f", "f88bb8a8724c81ec"],
    "seed_from_u64_0_first_f32": [0.8541927337646484, 0.19272810220718384, 0.9754980802536011,
                                  0.31179165840148926, 0.2528002858161926, 0.014432728290557861],
    "seed_from_u64_1_state": ["910a2dec89025cc1", "beeb8da1658eec67", "f893a2eefb32555e", "71c18690ee42c90b"],
    "seed_from_u64_1_first_f32": [0.01092076301574707, 0.8859519958496094, 0.1584458351135254,
                                  0.721820056438446, 0.3475397825241089, 0.14754152297973633],
    "pixel_seeds_frame0": {"1,0": 1973, "0,1": 9277, "599,400": 4892627, "1199,799": 9777951},
    "random_spheres_ledger": {"total_draws": 6027, "lambertian": 393, "metal": 72, "dielectric": 19, "spheres": 488,
                              "first_small_centre": [-10.302682, 0.2, -10.600717],
                              "first_small_albedo": [0.31004485, 0.12408885, 0.118598096]},
}

# (file, preset, W, H, spp, depth, use_bvh, crop (x0, y0, w, h) or None for the full frame)
FRAMES = [
    ("c1_small_200x100_4spp", "small", 200, 100, 4, 10, False, None),                     # BASELINE config 1
    ("c2_aras_1280x720_16spp_crop", "aras", 1280, 720, 16, 10, False, (560, 300, 64, 64)),  # config 2
    ("c3_random_spheres_1200x800_64spp_crop", "random_spheres", 1200, 800, 64, 10, False, (568, 330, 64, 64)),  # 3
    ("c5_perlin_spheres_1920x1080_128spp_crop_bvh", "perlin_spheres", 1920, 1080, 128, 10, True, (940, 420, 32, 32)),
    ("two_perlin_spheres_160x90_8spp", "two_perlin_spheres", 160, 90, 8, 10, False, None),
    # the presets with non-sphere Hitable arms
    ("w_cornell_smoke_300x200_16spp_crop", "cornell_smoke", 300, 200, 16, 10, False, (110, 50, 64, 64)),
    ("w_cornell_300x200_16spp_crop_bvh", "cornell", 300, 200, 16, 10, True, (110, 50, 64, 64)),
    ("w_random_600x400_8spp_crop", "random", 600, 400, 8, 10, False, (270, 150, 64, 48)),
    ("w_simple_light_160x90_8spp_crop", "simple_light", 160, 90, 8, 10, False, (60, 20, 48, 48)),
    ("smallpt_160x120_8spp_crop", "smallpt", 160, 120, 8, 10, False, (48, 30, 64, 64)),
]


def crop_pixels(W, crop):
    x0, y0, w, h = crop
    ys, xs = np.meshgrid(np.arange(y0, y0 + h), np.arange(x0, x0 + w), indexing="ij")
    return (ys * W + xs).reshape(-1).astype(np.uint32)


def main():
    with open(os.path.join(HERE, "rng_kat.json"), "w") as f:
        json.dump(RNG_KAT, f, indent=1)
    for name, preset, W, H, S, depth, bvh, crop in FRAMES:
        sc = ob.OracleScene(preset, W, H, use_bvh=bvh)
        buf = np.zeros((H, W, 3), np.float32)
        if crop is None:
            pixels = np.arange(W * H, dtype=np.uint32)
            _, rays = sc.update(S, depth, 0, buffer=buf, nthreads=0)
        else:
            pixels = crop_pixels(W, crop)
            _, rays = sc.update(S, depth, 0, buffer=buf, nthreads=0, pixels=pixels)
        rgb = buf.reshape(-1, 3)[pixels]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), preset=preset, width=W, height=H, samples=S,
                            depth=depth, use_bvh=bvh, pixels=pixels, rgb=rgb, ray_count=np.uint64(rays),
                            crop=np.array(crop if crop else (0, 0, W, H)))
        print(name, "rays", rays, "mean", rgb.mean(axis=0))


if __name__ == "__main__":
    main()
