"""Multi-GPU frame sharding for Scene::update: rows interleaved over ranks.

Row y of the frame belongs to rank y % N and is local row y // N of that rank's compact shard
buffer (the layout pt_render_shard_device writes). Per-pixel seeds depend only on (x, y, frame)
(scene.rs:99-101), so shards render independently; the only exchange is ONE all_gather of the
float3 shards per frame plus an 8-byte all_reduce of the ray count (scene.rs:118-120). xGMI is
point-to-point, and the message is small (11.5 MB at 1200x800), so one collective, no ring tuning.

Weak scaling uses the reference's OTHER data-parallel axis: progressive frames. Scene::update for
frame_num f depends on earlier frames only through the blend `out = out * f/(f+1) + col * 1/(f+1)`
(scene.rs:86-87,113-116), and its seeds depend on (x, y, f) alone, so N GPUs render frames 0..N-1 of
the same scene concurrently (each into a zeroed buffer, which leaves col * mix_new in it), one
all_gather collects them and the blend is replayed in frame order -- bit-identical to N sequential
Scene::update calls (`-F N` in the reference's CLI).
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


def frame_mix_prev(frame_num):
    """scene.rs:86: mix_prev = frame_num as f32 / (frame_num + 1) as f32, one IEEE f32 division."""
    return float(torch.tensor(float(frame_num), dtype=torch.float32) / torch.tensor(float(frame_num + 1), dtype=torch.float32))


def blend_frames(contribs):
    """contribs[f] = Scene::update(frame_num = f) applied to a ZERO buffer, i.e. col_f * mix_new(f)
    (0 * mix_prev + col * mix_new, scene.rs:114-116). Returns the buffer after frames 0..N-1 applied in order:
    out_f = out_{f-1} * mix_prev(f) + contribs[f] -- the same two roundings per channel as the sequential run."""
    acc = torch.zeros_like(contribs[0])
    for f in range(contribs.shape[0]):
        acc = acc * frame_mix_prev(f)
        acc = acc + contribs[f]
    return acc


def gather_progressive(dist, frame_buf, gathered, ray_count):
    """Rank r rendered frame_num = r of the same scene into `frame_buf` [H, W, 3] (zeroed before the launch).
    One all_gather of the frames + the ray-count all_reduce, then the blend in frame order."""
    world, height, width, ch = gathered.shape
    dist.all_gather_into_tensor(gathered.view(world * height, width, ch), frame_buf)
    dist.all_reduce(ray_count)
    return blend_frames(gathered)
