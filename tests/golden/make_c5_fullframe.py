#!/usr/bin/env python3
"""Full-frame evidence for BASELINE config 5 (perlin_spheres, 10 002 spheres, BVH world, 1920x1080x128).

The oracle walks the reference's own tree (bvh.rs:37-62: both children, original t_max; ~3 700 node visits per ray on
this scene), so the whole frame costs ~19 core-hours. This script renders it in row blocks, keeps every block's ray
count in a work directory (resumable: finished blocks are skipped), and writes
    tests/golden/c5_perlin_spheres_1920x1080_128spp_fullframe_bvh.npz
with the FULL frame's ray count (`frame_ray_count`, scene.rs:118-120), the ray count of every block of eight rows, the ray count of every
8x8-pixel tile (`tile_rays`, 135 x 240 uint32, row 0 = bottom like the frame: the per-pixel summands of scene.rs:118 added up per tile) and
every 251st pixel's colour (+ `ray_count`, the rays of those pixels alone, as in the other fixtures). The work directory keeps every
pixel's colour and ray count, so a different sampling needs no second render.
    nice -n 19 python tests/golden/make_c5_fullframe.py [--threads 8] [--work /tmp/c5_fullframe_v2]
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding as ob  # noqa: E402

W, H, S, DEPTH, ROWS, STRIDE, TILE = 1920, 1080, 128, 10, 8, 251, 8
NAME = "c5_perlin_spheres_1920x1080_128spp_fullframe_bvh"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--work", default="/tmp/c5_fullframe_v2")
    a = ap.parse_args()
    os.makedirs(a.work, exist_ok=True)
    sc = ob.OracleScene("perlin_spheres", W, H, use_bvh=True)
    sampled = np.arange(0, W * H, STRIDE, dtype=np.uint32)
    for b in range(H // ROWS):
        out = os.path.join(a.work, "block_%03d.npz" % b)
        if os.path.exists(out):
            continue
        pixels = np.arange(b * ROWS * W, (b + 1) * ROWS * W, dtype=np.uint32)
        buf = np.zeros((H, W, 3), np.float32)
        per_pixel = np.zeros(len(pixels), np.uint32)
        _, rays = sc.update(S, DEPTH, 0, buffer=buf, nthreads=a.threads, pixels=pixels, pixel_rays=per_pixel)
        assert int(per_pixel.sum()) == rays
        np.savez(out + ".tmp.npz", rays=np.uint64(rays), pixel_rays=per_pixel.astype(np.uint16).reshape(ROWS, W),
                 rgb_rows=buf[b * ROWS:(b + 1) * ROWS])
        os.replace(out + ".tmp.npz", out)
        print("block", b, "rays", rays, flush=True)
    blocks = [np.load(os.path.join(a.work, "block_%03d.npz" % b)) for b in range(H // ROWS)]
    block_rays = np.array([int(g["rays"]) for g in blocks], np.uint64)
    frame = np.concatenate([g["rgb_rows"] for g in blocks]).astype(np.float32)
    pixel_rays = np.concatenate([g["pixel_rays"] for g in blocks]).astype(np.uint32)
    assert frame.shape == (H, W, 3) and pixel_rays.shape == (H, W)
    tile_rays = pixel_rays.reshape(H // TILE, TILE, W // TILE, TILE).sum(axis=(1, 3)).astype(np.uint32)
    assert int(tile_rays.sum()) == int(block_rays.sum())
    pixels, rgb = sampled, frame.reshape(-1, 3)[sampled]
    # (`ray_count` means in every fixture "the rays of the recorded pixels": rendered once more on their own, which also checks the blocks' colours)
    buf = np.zeros((H, W, 3), np.float32)
    _, sampled_rays = sc.update(S, DEPTH, 0, buffer=buf, nthreads=a.threads, pixels=sampled)
    assert np.array_equal(buf.reshape(-1, 3)[sampled], rgb)
    np.savez_compressed(os.path.join(HERE, NAME + ".npz"), preset="perlin_spheres", width=W, height=H, samples=S, depth=DEPTH,
                        use_bvh=True, pixels=pixels, rgb=rgb, ray_count=np.uint64(sampled_rays), frame_ray_count=np.uint64(block_rays.sum()),
                        block_rows=ROWS, block_rays=block_rays, tile=TILE, tile_rays=tile_rays)
    print(NAME, "frame rays", int(block_rays.sum()), "sampled pixels", len(pixels), "their rays", sampled_rays, "mean", rgb.mean(axis=0))


if __name__ == "__main__":
    main()
