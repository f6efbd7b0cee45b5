// pt_kernels_list.hip -- MFMA list kernels of list worlds (GATE = false): the headline kernels.
#include "pt_kernel.h"
#include "pt_kernels.h"

namespace ptdev {

// ---- work ordering ------------------------------------------------------------------------------
// The frame ends when the slowest lane finishes its last pixel, and a pixel's samples are inherently
// serial (one RNG stream, scene.rs:96-111): a glass pixel needs ~700 dependent ray iterations, most of a
// 15 ms frame. Handing out the expensive tiles FIRST keeps that tail short. Tile costs come from a PILOT
// pass: the same kernel at 1 sample per pixel with throw-away seeds (random_seed path), writing nothing but
// the rays spent per 8x8 tile (~1.5 % of the frame's work). The order only decides WHEN a pixel is rendered,
// never its value.
// `checker_tiles_x` != 0: only the tiles of one colour of a checkerboard were measured (tcol + trow even); a tile of the other colour takes
// the mean of its measured neighbours as its cost (formed here on the fly: nobody else reads the costs of a new view).
// The kernel also zeroes the frame's work counter block (`work_counter`, 16 words): it is the last thing enqueued before the frame kernel.
__device__ __forceinline__ uint32_t ordered_cost(const uint32_t *cost, uint32_t t, uint32_t tiles_x, uint32_t tiles_y) {
    if (tiles_x == 0u) return cost[t];
    const uint32_t ty = t / tiles_x, tx = t - ty * tiles_x;
    if (((tx + ty) & 1u) == 0u) return cost[t];
    uint32_t sum = 0, n = 0;   // (its four neighbours are all of the measured colour)
    if (tx > 0u) sum += cost[t - 1u], n += 1u;
    if (tx + 1u < tiles_x) sum += cost[t + 1u], n += 1u;
    if (ty > 0u) sum += cost[t - tiles_x], n += 1u;
    if (ty + 1u < tiles_y) sum += cost[t + tiles_x], n += 1u;
    return n ? sum / n : 0u;
}
__global__ void pt_tile_order_kernel(uint32_t n_work_tiles, uint32_t *tile_cost, uint32_t cost_scale,
                                     uint32_t *tile_order, uint32_t checker_tiles_x, uint32_t checker_tiles_y, uint32_t *work_counter) {
    __shared__ uint32_t count[64], cursor[64];
    if (threadIdx.x < 64) count[threadIdx.x] = 0;
    if (work_counter && threadIdx.x < 16) work_counter[threadIdx.x] = 0u;
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < n_work_tiles; t += blockDim.x) {
        // (an unmeasured tile's cost is written back for the second pass: its neighbours are measured tiles, which nobody writes -- a measured
        //  tile's own word is left alone, so no thread reads a word another one stores to)
        const uint32_t c = ordered_cost(tile_cost, t, checker_tiles_x, checker_tiles_y);
        if (checker_tiles_x != 0u && (((t / checker_tiles_x) + (t % checker_tiles_x)) & 1u) != 0u) tile_cost[t] = c;
        const uint32_t b = c / cost_scale;
        atomicAdd(&count[b < 63u ? b : 63u], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // most expensive bucket first
        uint32_t acc = 0;
        for (int b = 63; b >= 0; --b) {
            cursor[b] = acc;
            acc += count[b];
        }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < n_work_tiles; t += blockDim.x) {
        const uint32_t b = tile_cost[t] / cost_scale;
        tile_order[atomicAdd(&cursor[b < 63u ? b : 63u], 1u)] = t;
    }
}

// Everything a new view's measuring launch needs zeroed or listed, in ONE launch (they were four fills and a kernel, each behind the other on
// the stream): the work counter block, the ray count, the tile costs, and -- `list` != nullptr -- the tiles of the measured colour.
__global__ void pt_frame_reset_kernel(uint32_t *work_counter, unsigned long long *ray_count, uint32_t *cost, uint32_t n_cost, uint32_t *list, uint32_t tiles_x,
                                      uint32_t tiles_y) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 16u) work_counter[t] = 0u;
    if (t == 0u) *ray_count = 0ull;
    if (t < n_cost) cost[t] = 0u;
    if (list && t < tiles_x * tiles_y) {
        const uint32_t ty = t / tiles_x, tx = t - ty * tiles_x;
        // two rows hold tiles_x measured tiles: ceil(tiles_x / 2) in the even row, the rest in the odd one
        if (((tx + ty) & 1u) == 0u) list[(ty >> 1) * tiles_x + ((ty & 1u) ? (tiles_x + 1u) / 2u : 0u) + (tx >> 1)] = t;
    }
}

// ---- measuring every OTHER tile -----------------------------------------------------------------
// The measuring launch of a new view traces one sample of the tiles of one colour of a checkerboard (tcol + trow even); a tile of the
// other colour takes the mean of its measured neighbours as its cost (ordered_cost above) and is traced from its first sample by the frame
// kernel (KArgs::checker). Config 3: measuring launch + order 0.30 -> 0.23 ms, frame 6.75 -> 6.70 ms; 16 spp 2.23 -> 2.15 ms. (One tile of
// every 2 x 2 block was measured too: 7.1 ms -- the order gets too coarse. tools/checker_ab.sh)

}  // namespace ptdev

namespace pthostside {

void launch_frame_reset(uint32_t *work_counter, unsigned long long *ray_count, uint32_t *cost, uint32_t n_cost, uint32_t *list, uint32_t tiles_x, uint32_t tiles_y,
                        hipStream_t stream) {
    const uint32_t n = std::max(std::max(n_cost, 16u), list ? tiles_x * tiles_y : 0u);
    hipLaunchKernelGGL(pt_frame_reset_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, work_counter, ray_count, cost, n_cost, list, tiles_x, tiles_y);
}

void mfma_list_kernels(bool moving, uint32_t blk, bool verify, bool pool, SphereKernel *frame, SphereKernel *measure) {
    // [moving][256 frame, 256 measure, verify, 768 frame, 768 measure, 1024 frame, 1024 measure]
    static const SphereKernel table[2][8] = {
        {pt_trace_kernel<false, true, true, false, false, false, false>, pt_trace_kernel<false, true, true, false, true, false, false>,
         pt_trace_kernel<false, true, true, true, false, false, false>,
         pt_trace_kernel<false, true, true, false, false, false, false, 768>, pt_trace_kernel<false, true, true, false, true, false, false, 768>,
         pt_trace_kernel<false, true, true, false, false, false, false, 1024>, pt_trace_kernel<false, true, true, false, true, false, false, 1024>,
         pt_trace_kernel<false, true, true, false, false, false, false, 1024, false, true>},
        {pt_trace_kernel<false, true, true, false, false, true, false>, pt_trace_kernel<false, true, true, false, true, true, false>,
         pt_trace_kernel<false, true, true, true, false, true, false>,
         pt_trace_kernel<false, true, true, false, false, true, false, 768>, pt_trace_kernel<false, true, true, false, true, true, false, 768>,
         pt_trace_kernel<false, true, true, false, false, true, false, 1024>, pt_trace_kernel<false, true, true, false, true, true, false, 1024>,
         pt_trace_kernel<false, true, true, false, false, true, false, 1024, false, true>}};
    const SphereKernel *t = table[moving ? 1 : 0];
    const int w = blk == 1024u ? 5 : (blk == 768u ? 3 : 0);
    *frame = verify ? t[2] : ((blk == 1024u && !pool) ? t[7] : t[w]);   // ([7]: the 1024-thread frame kernel without pixel pools)
    *measure = verify ? nullptr : t[w + 1];
}

void launch_tile_order(uint32_t n_work_tiles, uint32_t *tile_cost, uint32_t cost_scale, uint32_t *tile_order, uint32_t checker_tiles_x, uint32_t checker_tiles_y,
                       uint32_t *work_counter, hipStream_t stream) {
    hipLaunchKernelGGL(pt_tile_order_kernel, dim3(1), dim3(1024), 0, stream, n_work_tiles, tile_cost, cost_scale, tile_order, checker_tiles_x, checker_tiles_y, work_counter);
}

}  // namespace pthostside
