"""Multi-GPU frame sharding for Scene::update: rows interleaved over ranks.

Row y of the frame belongs to rank y % N and is local row y // N of that rank's compact shard
buffer (the layout pt_render_shard_device writes). Per-pixel seeds depend only on (x, y, frame)
(scene.rs:99-101), so shards render independently; the only exchange is ONE all_gather of the
float3 shards per frame plus an 8-byte all_reduce of the ray count (scene.rs:118-120). xGMI is
point-to-point, and the message is small (11.5 MB at 1200x800), so one collective, no ring tuning.
"""
import torch


def shard_rows(height, rank, world):
    """Rows owned by `rank` (same as pt_shard_rows)."""
    if world <= 0 or rank >= world or height <= rank:
        return 0
    return (height - rank + world - 1) // world


def padded_rows(height, world):
    """Rows of every rank's gather buffer (ranks with one row less are zero padded)."""
    return (height + world - 1) // world


def owned_pixels(height, width, rank, world):
    """Frame pixel indices (row-major, row 0 = bottom) of a rank's shard, in shard order."""
    if rank >= height:
        return torch.zeros(0, dtype=torch.int64)
    rows = torch.arange(rank, height, world)
    return (rows[:, None] * width + torch.arange(width)[None, :]).reshape(-1)


def deinterleave(gathered, height):
    """gathered: [world, padded_rows, width, 3] -> frame [height, width, 3] (row y = gathered[y % N, y // N])."""
    world, prow, width, ch = gathered.shape
    return gathered.permute(1, 0, 2, 3).reshape(prow * world, width, ch)[:height]


def gather_frame(dist, shard, gathered, ray_count, height):
    """One collective per frame: all_gather the shards, all_reduce the ray count, rebuild the frame.

    shard: [padded_rows, width, 3] float32 (this rank's rows, zero padded); gathered:
    [world, padded_rows, width, 3]; ray_count: int64[1]. Works on any backend (nccl = RCCL on
    ROCm, gloo on CPU)."""
    world, prow, width, ch = gathered.shape
    # rank r's shard lands at gathered[r]; the flat [world*prow, W, 3] view is accepted by nccl and gloo alike
    dist.all_gather_into_tensor(gathered.view(world * prow, width, ch), shard)
    dist.all_reduce(ray_count)
    return deinterleave(gathered, height)
