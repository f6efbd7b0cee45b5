// pt_args.h -- plain data shared by the host side of the C ABI (scene upload, kernel selection, launch) and the device
// kernels: the kernels' argument blocks, the device record layouts, and the compile-time constants that size the LDS
// carve. No device code here: host-only translation units include this without instantiating a kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_tree4.h"
#include "ptgpu.h"

namespace ptdev {

struct f3 {
    float x, y, z;
};

#ifndef PT_BLOCK
#define PT_BLOCK 256
#endif
#ifndef PT_MINWAVES
#define PT_MINWAVES 2
#endif
#ifndef PT_TREE4_WAVES
#define PT_TREE4_WAVES 4   // waves per SIMD the 4-wide tree kernels are compiled for (128 VGPRs)
#endif
#ifndef PT_TILE_LOG2
#define PT_TILE_LOG2 3
#endif
constexpr uint32_t kTileLog2 = PT_TILE_LOG2;             // work tiles are (1 << kTileLog2)^2 pixels
constexpr uint32_t kTileSide = 1u << kTileLog2, kTilePix = kTileSide * kTileSide;
constexpr int kBlock = PT_BLOCK;  // threads per workgroup (the main loop never synchronises across waves)
// LAZY Noise colours (pt_world.h): 0.5 (1 + sin(scale p.z + 10 turb(p))) (texture.rs:86-88) is formed later -- or never, when the path ends
// in black -- only where it is certainly finite, i.e. where the sine's argument is (f32::sin of a finite number is): |p| < kLazyNoiseReach
// per axis on the device, |scale| <= kLazyNoiseScale and Perlin gradients of at most kLazyNoiseGradient per component on the host. The
// trilinear weights of perlin.rs:66-69 sum to one, so |noise| <= 3 x 4 and |turb| <= 24: the argument stays below 1.1e6. (Far tighter than
// finiteness needs; a point, a scale or a gradient table outside is coloured where it is hit, and `path_odd` sees what comes out.)
constexpr float kLazyNoiseReach = 1.0e4f, kLazyNoiseScale = 1.0e2f, kLazyNoiseGradient = 4.0f;
constexpr uint32_t kGraphDepth = 24u, kGraphFrame = 24u;   // interpreted scene graphs (pt_graph.h): nested Hitable::ray_hit calls per lane, words per call
constexpr uint32_t kWorldNoiseLds = 4096u + 768u + (uint32_t)(kBlock / 64) * 768u;   // general-world kernel, worlds with Noise textures: gradients, permutations, wave_balanced_turb's 192 words per wave
constexpr int kBvhStack = 32;    // per-lane traversal stack entries (LDS)

struct DWideNode;
struct DMat {  // 32 B
    uint32_t kind;
    float a0, a1, a2;
    float param;
    int32_t tex;
    float pad0, pad1;
};
struct DTex {  // 32 B
    uint32_t kind;
    float c0, c1, c2;
    int32_t odd, even;
    float scale;
    float pad;
};

// Per-sphere shading record: everything Material::scatter / emitted needs for the common cases, resolved at
// scene creation so a hit costs one 64-byte fetch instead of the dependent chain
// sphere -> material index -> material -> texture (-> checker children).
//   q0 = (cx, cy, cz, radius)   q1 = (kind, flags, texture id, param as float bits)
//   q2 = colour A (constant albedo / metal albedo / emitted constant / checker ODD)   q3 = checker EVEN colour
constexpr uint32_t kShadeConst = 1u;    // texture is Constant: colour A
constexpr uint32_t kShadeChecker2 = 2u; // texture is Checker of two Constants: odd = A, even = q3
constexpr uint32_t kShadeNoise = 4u;    // texture is Noise: scale in A.x (texture.rs:86-89)

struct DCamera {  // camera.rs:8-19
    f3 origin, lower_left_corner, horizontal, vertical, u, v, w;
    float time0, time1, lens_radius;
};

// Frame parameters the main loop needs in vector registers are staged in LDS once per workgroup (one ds_read_b128 per
// group where they are used). Left as kernel arguments they are ~45 SGPRs that the compiler keeps live across the whole
// loop and spills into VGPR lanes (v_readlane / v_writelane around every use).
//   [0] clip_min.xyz, cull_u0   [1] clip_max.xyz, cull_inv_cell   [2] c0.xyz, rs2   [3] m0, gamma, inv_nx, inv_ny
//   [4..9] DCamera (24 floats, camera.rs:8-19 order)   [10] inv_ns, mix_prev, mix_new, prev_zero (0 / 1)   [11] sky.xyz, has_sky
//   [12] as u32 bits: cull_axis, cull_always, max_depth, samples   [13] tile culling's per-ray reach: 2 kappa, kappa (2 Rs^2 + r_max^2) + r_min^2, r_min, -
//   [14] tile culling's second axis: cull_u0_2, cull_inv_cell_2, cull_axis2 (u32 bits), -
constexpr uint32_t kLdsParamBytes = 15u * 16u;

// Bit of a ray's 32-bit tile mask (pt_prefilter.h, intersect_list_mfma) that fragment row `row` of a tile ends up in. A lane of
// v_mfma_f32_32x32x16_f16 holds rows (r & 3) + 8 (r >> 2) + 4 * (lane >> 5) in accumulator registers r = 0..15; the kernel shifts
// the signs in so that register r lands at bit 15 - r, the low half of the wave supplying bits 0..15 and the high half 16..31.
// tile_sphere is stored in this BIT order, so a candidate bit indexes it directly.
constexpr uint32_t tile_bit_of_row(uint32_t row) { return (15u - ((row & 3u) | ((row >> 3) << 2))) + 16u * ((row >> 2) & 1u); }
constexpr int kCullCells = 128;   // resolution of the tile-culling lookups along each of the two axes (two pairs of tables: 2 KB of LDS)

struct KArgs {
    // scene (HBM resident)
    const float4 *spheres;       // cx, cy, cz, radius
    const float4 *shade;         // [4*n_spheres] per-sphere shading record (DShadeRec): ONE 64-byte fetch per hit
    const float4 *spheres_r2;    // cx, cy, cz, radius*radius (sphere.rs:36), scan layout, padded to n_spheres_pad
    const uint32_t *sphere_mat;  // material index per sphere
    const float4 *motion;        // MOVING kernels: [2*n_spheres] (dx, dy, dz, inv_time_delta), (time_start, is_moving, -, -)
    const DMat *mats;
    const DTex *texs;
    const float4 *perlin_vec;    // 256 gradients (xyz, pad)
    const uint32_t *perlin_perm; // 768 entries: perm_x | perm_y | perm_z
    const DWideNode *wnodes;  // binary internal tree (variant bit 2048): children's AABBs inside the parent
    const DNode4Q *nodes4;    // 4-wide internal tree (default of the tree kernels) as packed 64-byte nodes, root = node 0
    const float4 *slotrec;    // its leaves: [4 * (node * 4 + slot)] = sphere | gate min, chain count | gate max, chain offset | (rank bits, sphere index): ONE 64-byte fetch per exact test
    const uint32_t *rank_sphere;     // BVH worlds: sphere of each DFS leaf rank (inverse of leaf_rank; decodes the hit key)
    const float4 *shade_rank;        // BVH worlds, 4-wide tree: the shading records in DFS-rank order (the hit key carries the rank)
    const uint32_t *leaf_rank;       // DFS (lhs before rhs) order of each sphere's leaf, for equal-t ties
    float root_min[3], root_max[3];
    uint32_t n_nodes, nodes_in_lds, bvh_stack_entries;
    // BVH mode acceleration structure built by pt_scene_create (the caller's tree only defines the RESULT)
    const float4 *gate;          // [2*n_spheres] AABB (min, max) of each sphere's parent node in the caller's tree;
                                 // .w of the pair = count / offset of further ancestors in gate_chain (count ~0: never hit)
    const float4 *gate_chain;    // ancestors whose test is not implied by the box below them (inverted boxes only)
    const uint32_t *bvh_large;   // spheres kept out of the internal tree (huge radius): tested for every ray
    uint32_t n_bvh_large;
    // uniform cell grid (pt_grid.h; pt_host.h GridPlan): non-NULL grid_cells selects it in the 4-wide tree kernels
    const uint4 *grid_cells;     // [n_records][5]: four spheres (cx, cy, cz, radius) + (index x 4 | link in the last word)
    const float4 *grid_rec;      // per SPHERE, the 64-byte record of an exact test (slotrec's layout)
    const uint32_t *grid_large;  // spheres outside the grid: tested for every ray
    uint32_t n_grid_large;
    uint32_t grid_n[3];
    float grid_min[3], grid_h[3], grid_inv_h[3];   // (cells are cubes, except along an axis with ONE cell: that one spans the spheres' whole extent)
    float grid_centre[3], grid_half_diag, grid_d_build;   // a ray whose origin is farther than d_build - half_diag from the centre walks the tree instead
    uint32_t n_spheres;
    uint32_t n_spheres_pad;      // multiple of kScanUnroll; padding entries can never be hit
    // MFMA discriminant prefilter (list mode, see "MFMA prefilter" below); n_tiles == 0 disables it
    const uint4 *afrag;          // [n_tiles][2 chunks][64 lanes] x 8 f16: sphere-feature A fragments
    const uint16_t *tile_sphere; // [n_tiles*32] sphere index behind each bit of a tile mask (tile_bit_of_row), 0xffff = padding row
    const uint32_t *large;       // spheres outside the prefilter's range: tested exactly for every ray
    uint32_t large0;             // large[0] (0xffffffff: there is none): the first of them is tested without a load of the list or a branch
    uint32_t n_tiles, n_large;
    float c0[3];                 // feature-space origin (f32-exact), radius bound of the prefiltered set
    float rs2;                   // Rs^2, Rs >= max(|c - c0| + |r|) over prefiltered spheres
    float m0, gamma;             // margin = a * (m0 + gamma * (|o - c0|^2 + Rs^2))
    // tile culling (DESIGN.md "tile culling"): tiles hold spheres sorted along cull_axis; a wave runs only the tiles
    // some lane's ray segment (origin .. nearest exact hit so far, clipped to the sorted spheres' box) can overlap
    const uint32_t *cull_tab;    // per axis: [kCullCells] tiles reaching up to cell c or beyond | [kCullCells] tiles starting at cell c or before
                                 // (first the sort axis, then the second axis: 4 x kCullCells words)
    uint32_t cull_axis;          // 0..2; 3 = culling off
    uint32_t cull_always;        // tiles that are always run (they hold spheres outside the sorted set)
    float cull_u0, cull_inv_cell;
    uint32_t cull_axis2;         // the axis the strips of the sort axis are sorted along (tiles are boxes in two axes); tables of all ones when unused
    float cull_u0_2, cull_inv_cell_2;
    float clip_min[3], clip_max[3];  // box of the sorted spheres, padded by 2e-3 + 1e-5 |.| (lane_tile_mask adds each ray's own reach)
    float cull_reach[3];             // 2 kappa, kappa (2 Rs^2 + r_max^2) + r_min^2, r_min (lane_tile_mask)
    uint32_t verify;             // debug: count exact-positive pairs the prefilter did not flag
    unsigned long long *debug;   // [4] misses, candidates, overflow fallbacks, exact positives
    unsigned long long *wave_end; // optional (PTGPU_TIMING=1): wall clock at which each wave left the main loop
    int32_t bvh_root;
    uint32_t has_sky;
    f3 sky;
    uint32_t has_noise;
    // frame
    DCamera cam;
    uint32_t width, height, samples, max_depth, frame_num;
    float inv_nx, inv_ny, inv_ns, mix_prev, mix_new;  // scene.rs:82-87 (computed on the host in f32)
    uint32_t random_seed;
    // A frame in two launches (phase 1: the first samples of every pixel, in natural order, MEASURING the tiles; phase 2: the
    // rest, ordered by those costs). A pixel's samples are one serial RNG stream: phase 1 parks (xoshiro state, colour sum) in
    // px_state (12 dwords per pixel) where phase 2 picks them up; 0 = the whole frame in one launch.
    uint32_t phase;
    uint32_t checker;            // phase 2: only the tiles with an even tcol + trow were measured (their pixels continue at sample 1); the others start here
    uint4 *px_state;
    uint32_t first_static;       // 0, or the number of items handed out statically as the waves' first fetches (grid x 1024)
    uint32_t refill_min;         // lanes that must be waiting before a wave fetches new pixels (4; 8 below 32 spp)
    // 1024-thread frame kernels: per-wave pool of ready-to-start pixels in LDS (pt_kernel.h POOL): entries per wave, byte offset of the pools in
    // the dynamic LDS, the fair share (items left per wave of the grid) below which claims stop filling the pool, floor(2^32 / waves of the grid)
    uint32_t pool_slots, pool_off, pool_tail, pool_waves_magic;
    uint32_t grid_park_max, grid_park_after;   // cell-grid kernels: at most this many lanes still walking park their walk (0: never), after this many rounds of a call (pt_grid.h)
    uint32_t ready_min;          // 4-wide tree: lanes with a finished traversal before the wave leaves the traversal loop to shade
    uint32_t drain_at;           // 4-wide tree: a lane holding more than this many leaf candidates triggers the wave's drain
    uint64_t seed_base;
    // sharding: rows y with y % shard_count == shard_index, compact buffer
    uint32_t shard_index, shard_count, local_rows;
    uint32_t tiles_x, n_items;  // 8x8 tiles over (width x local_rows); n_items = tiles * 64
    uint32_t tiles_x_magic;     // floor(2^32 / tiles_x): tile / tiles_x = umulhi(tile, magic) (+ 1 after one correction step)
    // outputs / work queue
    float *rgb;
    uint32_t prev_zero;          // the caller vouches that the buffer holds +0.0f everywhere: the blend uses 0.0f instead of loading it
    unsigned long long *ray_count;
    uint32_t *work_counter;      // [0] next work item; the other words of its 64-byte block: hand-over protocol of the cooperative mode (pt_coop.h)
    // wave-cooperative mode of the wide list kernels (pt_coop.h): pixels handed from waves still in their main loop to waves that have left it
    uint64_t *tail_box;          // [tail_cap][16] one mailbox per wave of the grid: a handed-over pixel (64 B), its state word, the exit word
    uint32_t tail_cap;           // waves of the grid; 0: the mode is off for this launch
    uint32_t tail_dry0;          // the frame has no more work items than the grid has lanes: the list is dry once every wave has fetched
    uint32_t tail_gen;           // this launch's stamp in the mailboxes' words (never 0, below 2^30)
    uint32_t tail_live_max;      // a wave with at most this many live pixels hands over to any idle worker it finds ...
    uint32_t tail_streak;        // ... any wave does after this many probes in a row that found an idle worker
    uint32_t tail_period_mask;   // a wave with more live pixels probes when (iteration & mask) == 0 (3: every fourth)
    float tail_min_est;          // a pixel with fewer estimated rays left is not worth the hand-over's global round trips
    uint32_t tail_dbg;           // -DPT_DEVKNOBS builds (PTGPU_COOP_DBG): 1 no workers, 2 no probes, 4 no started / done counting
    const uint32_t *tile_order;  // optional permutation of the 8x8 work tiles (expensive tiles first)
    uint32_t *tile_cost;         // optional: rays spent per work tile (accumulated when a pixel completes): the pilot pass' result,
                                 // or what a frame kernel measures for the next frame of the same view
    float *gstack;  // global path-stack fallback when max_depth*3*kBlock*4 exceeds the LDS budget
    uint32_t stack_in_lds;       // 256-thread kernels: number of attenuation-stack slots (3 per level; 1 on the 4-wide tree kernels, WST below) kept in LDS; the rest in gstack
    uint32_t lds_sphere_bytes;  // offsets of the dynamic LDS carve
};

// ---- LDS queue geometry (the host sizes the carve with the same constants the kernels index it with) ----
constexpr int kScanUnroll = 8;
constexpr int kQueueCap = 20;  // per-lane candidate slots (u16) of the exact scan, drained above kQueueCap - kScanUnroll

constexpr int kEntCap = 4;      // u32 tile masks a lane can hold between two drains (a lane only queues tiles of its own mask)
// Phase 2 is balanced over the wave: the lanes' candidates are expanded into one list of (ray, sphere) pairs per wave and
// every lane takes one PAIR per round, whoever's ray it belongs to (a lane-owns-its-candidates loop ran 4.0 rounds per
// bounce at 23 % lane utilisation: 59 candidates per wave, unevenly spread). Per wave: the pair list and one 64-bit
// (t, tie-break) key per ray that the pairs' exact tests are reduced into with ds_min_u64.
constexpr int kPairCap = 192;
constexpr uint32_t kWavePairBytes = kPairCap * 4u + 64u * 8u;
__host__ __device__ constexpr uint32_t mfma_queue_bytes(uint32_t blk) { return (uint32_t)kEntCap * blk * 4u + (blk / 64u) * kWavePairBytes; }
__host__ __device__ constexpr uint32_t scan_queue_bytes(uint32_t blk) { return ((uint32_t)(kQueueCap + 1) * blk * 2u + 15u) / 16u * 16u; }

constexpr int kReadyMin = 56;  // shade as soon as this many lanes of the wave have a finished traversal
#ifndef PT_SHARE_MIN
#define PT_SHARE_MIN 4
#endif
constexpr int kShareMin = PT_SHARE_MIN;   // 4-wide tree with work sharing: lanes without traversal work before subtrees change hands

#ifndef PT_LEAFQ
#define PT_LEAFQ 8
#endif
constexpr int kLeafQ = PT_LEAFQ;   // per-lane candidate slots; drained when a lane holds more than kLeafQ - 4
constexpr uint32_t kPairLaneShift = 26u;   // pair = owner lane << 26 | leaf slot (node * 4 + slot; the tree has < 65536 nodes)
__host__ __device__ constexpr uint32_t tree4_queue_bytes(uint32_t blk) { return (uint32_t)kLeafQ * blk * 4u + (blk / 64u) * kWavePairBytes; }
// cell-grid kernels (pt_grid.h): the last few lanes of a wave still walking PARK their walk (8 words) here and finish it in the wave's next call
constexpr uint32_t kGridParkMax = 8u, kGridParkWords = 8u;
__host__ __device__ constexpr uint32_t grid_park_bytes(uint32_t blk) { return (blk / 64u) * kGridParkMax * kGridParkWords * 4u; }

// ---- binary internal tree node (variant bit 2048 / trees beyond the 4-wide format) ----
struct DWideNode {  // 64 B
    float lmin[3], lmax[3];  // lhs inner node: box CENTRE, HALF extent; lhs leaf: the sphere (centre, lmax[0] = radius)
    float rmin[3], rmax[3];  // same for rhs
    int32_t lhs, rhs;        // >= 0 inner node, < 0 ~sphere
    uint32_t pad0, pad1;     // 1 / smallest |radius| below lhs / rhs (float bits)
};

// ---- argument block of the general-world kernel (pt_world.h) ----
struct WArgs {
    const pt_hitable *hit;   // [n_hit] HitableList order, the 64-byte C-ABI records
    const pt_affine *xf;     // Instance transforms (Affine3A, inverse)
    const pt_bvh_node *nodes;
    const DMat *mats;
    const DTex *texs;
    const float4 *perlin_vec;
    const uint32_t *perlin_perm;
    const uint4 *image_table;   // Texture::Image sources: (byte offset, width, height, -) per image
    const uint8_t *image_bytes;
    uint32_t has_image;         // some texture is an Image: rect hits then compute (u, v) (rect.rs:97-98)
    uint32_t atts_finite;       // every material colour is finite (a path that ends in black then needs no fold)
    // scene graphs that do not flatten (pt_graph.h): pt_node rows, HitableList children, the root; `nodes` then holds the graph's BVHNode rows
    const uint4 *gnodes;
    const uint32_t *gchildren;
    uint32_t groot;
    float *gframes;             // kGraphDepth x kGraphFrame words per lane of the grid
    uint32_t n_hit, n_xf;
    int32_t bvh_root;        // >= 0: BVHNode::ray_hit over `nodes`; < 0: HitableList::ray_hit
    uint32_t bvh_stack_entries;
    uint32_t has_sky;
    f3 sky;
    uint32_t has_noise;
    DCamera cam;
    uint32_t width, height, samples, max_depth, frame_num;
    float inv_nx, inv_ny, inv_ns, mix_prev, mix_new;
    uint32_t random_seed;
    uint32_t refill_min;
    uint64_t seed_base;
    uint32_t shard_index, shard_count, local_rows;
    uint32_t tiles_x, n_items;
    uint32_t tiles_x_magic;
    uint32_t prev_zero;   // as KArgs::prev_zero
    float *rgb;
    unsigned long long *ray_count;
    uint32_t *work_counter;
    float *gstack;
    uint32_t stack_in_lds;
    // heavy-first work order, as in pt_trace_kernel (pt_kernel.h KArgs): 8x8 tiles in the order of `tile_order` (nullptr: natural),
    // rays per tile accumulated into `tile_cost` when a pixel completes, and a frame in two launches -- phase 1 traces the
    // first sample of every pixel and parks (xoshiro state, colour sum) in px_state, phase 2 continues from there; 0 = one launch
    const uint32_t *tile_order;
    uint32_t *tile_cost;
    uint32_t phase;
    uint4 *px_state;
};

}  // namespace ptdev
