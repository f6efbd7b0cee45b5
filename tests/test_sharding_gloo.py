"""N > 1 path on CPU: two gloo ranks shard a frame by interleaved rows, all_gather, de-interleave.

The renderer itself needs a GPU, so each rank fills its shard with the ORACLE restricted to the
rank's own pixels (tests may use the oracle as a stand-in); what is under test is the product's
sharding module (pathtrace-rs_amd/sharding.py: row ownership, padding, the single collective and
the de-interleave) -- the same code bench.py runs over RCCL.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, W, H, S, preset, out_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import oracle_binding as ob
    from conftest import load_ptgpu
    import importlib.util
    spec = importlib.util.spec_from_file_location("pathtrace_rs_amd_sharding",
                                                  os.path.join(ROOT, "pathtrace-rs_amd", "sharding.py"))
    sharding = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sharding)
    ptgpu = load_ptgpu()

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows = sharding.shard_rows(H, rank, world)
        assert rows == ptgpu.shard_rows(H, rank, world)          # host helper == C ABI pt_shard_rows
        prow = sharding.padded_rows(H, world)
        # this rank's rows, rendered independently (no data-path collective while rendering)
        sc = ob.OracleScene(preset, W, H)
        full = np.zeros((H, W, 3), np.float32)
        px = sharding.owned_pixels(H, W, rank, world).numpy().astype(np.uint32)
        assert len(px) == rows * W
        _, rays = sc.update(S, 10, 0, buffer=full, nthreads=2, pixels=px)
        shard = torch.zeros((prow, W, 3), dtype=torch.float32)
        shard[:rows] = torch.from_numpy(full.reshape(-1, 3)[px].reshape(rows, W, 3))   # compact shard layout
        gathered = torch.empty((world, prow, W, 3), dtype=torch.float32)
        ray_count = torch.tensor([rays], dtype=torch.int64)
        frame = sharding.gather_frame(dist, shard, gathered, ray_count, H)
        if rank == 0:
            np.savez(out_path, frame=frame.numpy(), rays=int(ray_count.item()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _worker_frames(rank, world, port, W, H, S, preset, out_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import oracle_binding as ob
    import importlib.util
    spec = importlib.util.spec_from_file_location("pathtrace_rs_amd_sharding",
                                                  os.path.join(ROOT, "pathtrace-rs_amd", "sharding.py"))
    sharding = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sharding)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank r renders progressive frame r into a ZEROED buffer (what bench.py does with pt_render_device)
        buf, rays = ob.OracleScene(preset, W, H).update(S, 10, rank, nthreads=2)
        gathered = torch.empty((world, H, W, 3), dtype=torch.float32)
        ray_count = torch.tensor([rays], dtype=torch.int64)
        frame = sharding.gather_progressive(dist, torch.from_numpy(buf), gathered, ray_count)
        if rank == 0:
            np.savez(out_path, frame=frame.numpy(), rays=int(ray_count.item()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_progressive_frames_across_ranks_equal_sequential_updates(tmp_path, oracle, world):
    """Weak-scaling mode: frame_num = rank on each rank, one all_gather, blend replayed in order ==
    Scene::update called for frame 0, 1, .. on one buffer (scene.rs:86-87,113-116), bit for bit."""
    W, H, S, preset = 48, 32, 2, "small"
    out = str(tmp_path / "frames.npz")
    mp.spawn(_worker_frames, args=(world, _free_port(), W, H, S, preset, out), nprocs=world, join=True)
    got = np.load(out)
    sc = oracle.OracleScene(preset, W, H)
    ref, total = np.zeros((H, W, 3), np.float32), 0
    for f in range(world):
        _, rays = sc.update(S, 10, f, buffer=ref)
        total += rays
    assert int(got["rays"]) == total
    assert np.array_equal(got["frame"], ref)


@pytest.mark.parametrize("H", [40, 41])     # even split and a ragged last row
def test_two_rank_sharded_frame_equals_full_frame(tmp_path, oracle, H):
    W, S, preset, world = 60, 2, "small", 2
    out = str(tmp_path / "frame.npz")
    mp.spawn(_worker, args=(world, _free_port(), W, H, S, preset, out), nprocs=world, join=True)
    got = np.load(out)
    ref, ref_rays = oracle.OracleScene(preset, W, H).update(S)
    assert int(got["rays"]) == ref_rays                            # all_reduce of per-shard ray counts
    assert np.array_equal(got["frame"], ref)                        # union of shards == full frame, bit for bit


def test_deinterleave_and_ownership_math():
    spec_path = os.path.join(ROOT, "pathtrace-rs_amd", "sharding.py")
    import importlib.util
    spec = importlib.util.spec_from_file_location("pathtrace_rs_amd_sharding_t", spec_path)
    sharding = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sharding)
    for H, W, N in ((8, 3, 8), (800, 5, 8), (13, 4, 3), (5, 2, 8)):
        prow = sharding.padded_rows(H, N)
        frame = torch.arange(H * W * 3, dtype=torch.float32).reshape(H, W, 3)
        gathered = torch.zeros((N, prow, W, 3))
        seen = torch.zeros(H * W, dtype=torch.int32)
        for r in range(N):
            rows = sharding.shard_rows(H, r, N)
            px = sharding.owned_pixels(H, W, r, N)
            seen[px] += 1
            gathered[r, :rows] = frame.reshape(-1, 3)[px].reshape(rows, W, 3)
        assert torch.all(seen == 1)                                # disjoint cover of the frame
        assert torch.equal(sharding.deinterleave(gathered, H), frame)
