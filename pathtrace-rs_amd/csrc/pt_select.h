// pt_select.h -- which kernel renders a frame, and with what launch geometry: a PURE function of the scene's traits (fixed at
// pt_scene_create), the frame's pt_params / camera shutter, and the tuning word. No HIP call, no allocation, no state: the
// launch path (pt_launch.hip) executes the choice, and tests/test_host_cpu.py enumerates it for every preset and world class
// through pt_debug_select (no GPU needed). DESIGN.md section 4 explains WHY each rule is what it is; this file is the
// table.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>

#include "pt_args.h"
#include "pt_devknobs.h"

namespace ptsel {

using namespace ptdev;

constexpr uint32_t kLdsBudget = 160u * 1024u;      // LDS per CU on gfx950
constexpr uint32_t kPoolSlots = 32u, kPoolSlotsMin = 8u;   // wide kernels: entries of a wave's pixel pool (pt_kernel.h POOL), and the fewest a wide kernel is chosen with
constexpr uint32_t kTwoLaunchMinSamples = 12u;     // frames of a new view with at least this many samples measure their tiles with their own
                                                   // first sample (random_spheres 1200x800: -11 % at 8 spp, +2 % at 12, +8 % at 16, +5 % at 64)
constexpr uint32_t kMinOrderedTiles = 256u;        // below this many work tiles the order is not worth its launches
constexpr uint32_t kWideBlock = 768u;              // MFMA list kernels: one workgroup of 12 waves per CU when 16 do not fit
constexpr uint32_t kLdsPerBlockMax = 96u * 1024u;  // exact-scan kernels: leave room for >= 1 co-resident block
constexpr uint32_t kListTreeMin = 768u;            // = 24 MFMA tiles: beyond that the fragments no longer fit beside the rest; list worlds walk the tree
constexpr uint32_t kMaxMfmaTiles = 24u;

// (the bits of the tuning word that select code paths: pt_devknobs.h)

// Facts about a scene that kernel selection may look at. Filled by pt_scene_create*; never changes afterwards
// (bin_* : the binary internal tree is built the first time a launch needs it).
struct SceneTraits {
    bool is_world = false;         // general world (Rect / Cuboid / Instance / ConstantMedium entries): pt_world_kernel
    // ---- sphere scenes (Sphere, or Sphere + MovingSphere entries)
    uint32_t n_spheres = 0;
    bool has_caller_bvh = false;   // created with BVH nodes: pt_params.use_bvh is allowed
    uint32_t n_tiles = 0;          // MFMA prefilter tiles of 32 spheres (0: the prefilter cannot take this scene)
    bool palette_ok = false;       // every attenuation is a per-sphere constant or one of two checker colours (wide kernels' palette stack)
    bool word_ok = false;          // every attenuation fits one stack word (4-wide tree kernels)
    bool has_noise = false;
    bool has_motion = false;       // some entry is a MovingSphere; swept bounds cover ray times in [time_lo, time_hi]
    float time_lo = 0.f, time_hi = 0.f;
    uint32_t n_nodes4 = 0, depth4 = 0;   // the 4-wide internal tree
    bool tree4_packed = false;
    bool grid_ok = false;          // the scene has a uniform cell grid (pt_host.h GridPlan): the tree kernels walk it instead of the tree
    uint32_t bin_nodes = 0, bin_depth = 0;   // the binary internal tree once built (0: not yet)
    // ---- general worlds
    uint32_t n_hitables = 0, n_world_xf = 0, ref_bvh_depth = 0;
    bool has_media = false, has_image = false;
    bool atts_finite = false;      // every material colour is finite: a path that ends in black needs no fold (pt_world.h)
    bool noise_finite = false;     // ... and every Noise texture's scale and the Perlin gradients are such that its colour is finite wherever |p| < 1e4 (pt_args.h kLazyNoise*)
    bool has_chains = false;       // some entry sits below several Instance levels, or below Instances around its medium (scene graphs)
    bool is_graph = false;         // a scene graph that does not flatten: interpreted (pt_graph.h)
};

enum class Family : uint32_t { World = 0, TreeBinary = 1, Tree4 = 2, Mfma = 3, ScanLds = 4, ScanHbm = 5 };
enum class Order : uint32_t { Natural = 0, Measured = 1 };   // Measured: heavy tiles first (two launches for a new view, last frame's costs for a repeated one)

struct KernelChoice {
    Family family = Family::ScanHbm;
    bool ref_bvh = false;       // BVHNode::ray_hit semantics requested (gates + DFS-rank ties, or the caller's tree for a world)
    bool moving = false;        // MOVING instantiation
    bool gate = false;          // GATE instantiation: a BVH world on the MFMA list kernel
    bool verify = false;
    // general-world kernel: <BVH, HIT_LDS, OCC, MEDIA>
    bool world_hit_lds = false, world_media = false, world_chains = false;
    bool world_graph = false;   // the scan is the interpreted walk of a scene graph (pt_graph.h)
    bool world_lazy = false;    // Noise colours of scatters are formed when a lit path ends, wave-balanced (pt_world.h LAZY)
    uint32_t world_occ = 3;
    uint32_t block = 256;       // threads per workgroup
    uint32_t lds_bytes = 0;     // dynamic LDS per workgroup
    uint32_t bpc = 1;           // resident workgroups per CU the LDS and the policy allow (the launch clamps by registers)
    bool bpc_forced = false;    // pt_scene_set_tuning gave the number: no register clamp
    Order order = Order::Natural;
    bool needs_binary_tree = false;
    bool gstack = false;        // some attenuation-stack levels live in HBM: the launch sizes the global stack
    // values the kernels' LDS carve is driven by (copied into KArgs / WArgs)
    uint32_t sph_bytes = 0, n_tiles = 0, stack_in_lds = 0, nodes_in_lds = 0, bvh_stack_entries = 0, cull_off = 0;
    uint32_t refill_min = 4;
    uint32_t pool_slots = 0, pool_off = 0;   // wide frame kernels: entries of each wave's pixel pool, byte offset of the pools in the LDS carve
    bool grid = false;          // Tree4 family: the traversal structure is the uniform cell grid (pt_grid.h), not the 4-wide tree
    bool coop = false;          // wide list kernels: idle waves finish pixels handed over by busy ones, 64 lanes per ray (pt_coop.h)
};

struct Knobs {                  // tuning word + the development overrides (-1 / 0: library default)
    uint32_t variant = 0, blocks_per_cu = 0;
    int refill = -1, pool = -1;
    bool world_occ3 = false, world_occ4 = false;   // (development knobs: cap the general-world kernel's waves per SIMD)
};

inline bool shutter_inside(const SceneTraits &t, float time0, float time1) {
    const float lo = std::min(time0, time1), hi = std::max(time0, time1);
    return std::isfinite(lo) && std::isfinite(hi) && lo >= t.time_lo && hi <= t.time_hi;
}

inline uint32_t scan_pad(uint32_t n) { return (n + (uint32_t)kScanUnroll - 1u) / (uint32_t)kScanUnroll * (uint32_t)kScanUnroll; }

// Does this frame take the heavy-first work order?
inline Order order_for(uint32_t n_work_tiles, uint32_t samples, uint32_t variant, bool verify) {
    if (n_work_tiles < kMinOrderedTiles || samples < kTwoLaunchMinSamples || (variant & kVarNaturalOrder) || verify) return Order::Natural;
    return Order::Measured;
}

inline uint32_t work_tiles(uint32_t width, uint32_t local_rows) {
    return ((width + kTileSide - 1u) / kTileSide) * ((local_rows + kTileSide - 1u) / kTileSide);
}

// ---- general worlds ---------------------------------------------------------------------------------------
inline void select_world(const SceneTraits &t, const pt_params &p, uint32_t local_rows, const Knobs &k, KernelChoice &c) {
    c = KernelChoice{};
    c.family = Family::World;
    c.ref_bvh = p.use_bvh != 0;
    c.block = (uint32_t)kBlock;
    c.bvh_stack_entries = t.ref_bvh_depth + 2u;
    uint32_t lds = t.has_noise ? kWorldNoiseLds : 0u;
    if (c.ref_bvh) lds += c.bvh_stack_entries * (uint32_t)kBlock * 4u;
    c.world_graph = t.is_graph;
    c.world_hit_lds = !t.is_graph && t.n_hitables * 64u + t.n_world_xf * 96u <= 40960u;   // records + transforms staged in LDS
    if (c.world_hit_lds) lds += t.n_hitables * 64u + t.n_world_xf * 96u;
    c.world_chains = t.has_chains;
    c.world_lazy = t.has_noise && t.noise_finite && c.world_hit_lds && !t.has_chains && p.max_depth <= 64u && !(k.variant & kVarWorldEager);
    const uint64_t path_bytes = (uint64_t)p.max_depth * (c.world_lazy ? 4ull : 3ull) * (uint32_t)kBlock * 4ull;
    c.stack_in_lds = (lds + path_bytes <= 60u * 1024u) ? 1u : 0u;
    if (c.stack_in_lds) lds += (uint32_t)path_bytes;
    c.gstack = !c.stack_in_lds;
    // four waves per SIMD (128 VGPRs, no (u, v) in the hit record) for worlds without noise / image textures
    const bool occ4 = !t.has_noise && !t.has_image && !t.has_chains && k.blocks_per_cu == 0 && 4u * lds <= kLdsBudget && !k.world_occ3;
    // ... and five (96 VGPRs, a handful of spills) where the LDS lets five workgroups share a CU: list worlds at depth 10 (cornell +7.5 %,
    // cornell_smoke +9 %); a BVH world's traversal stack does not leave the room
    c.world_occ = occ4 ? ((c.world_hit_lds && 5u * lds <= kLdsBudget && !k.world_occ4) ? 5u : 4u) : 3u;
    // worlds whose records do not fit LDS (more than ~600 hitables) share the MEDIA = true code
    c.world_media = t.has_media || !c.world_hit_lds || t.has_motion || t.has_chains;
    c.lds_bytes = lds;
    uint32_t bpc = k.blocks_per_cu ? k.blocks_per_cu : c.world_occ;
    c.bpc_forced = k.blocks_per_cu != 0;
    const uint32_t lds_limit = lds ? (kLdsBudget / lds) : 8u;
    if (bpc > lds_limit) bpc = lds_limit ? lds_limit : 1u;
    c.bpc = std::min(bpc, 8u);
    c.refill_min = k.refill >= 0 ? (uint32_t)k.refill : (p.samples < 32u ? 8u : 4u);
    c.order = order_for(work_tiles(p.width, local_rows), p.samples, k.variant, false);
}

// ---- sphere scenes ----------------------------------------------------------------------------------------
// `tree4_wg_by_regs`: workgroups of 256 threads per CU the 4-wide tree kernel's register count allows (4 at 128 VGPRs).
inline void select_spheres(const SceneTraits &t, const pt_params &p, float time0, float time1, uint32_t local_rows, const Knobs &k,
                           uint32_t tree4_wg_by_regs, KernelChoice &c) {
    const uint32_t v = k.variant;
    const bool ref_bvh = p.use_bvh != 0;
    // list mode walks the internal tree instead of scanning when the scan would be the slower option: more than kListTreeMin
    // spheres, or more than the scan's 16-bit candidate indices address
    const bool list_tree = !ref_bvh && (v & (kVarExactScan | kVarNoListTree)) == 0 && (t.n_spheres > kListTreeMin || t.n_spheres > 0xfff0u);
    // A BVH WORLD is a list world plus two rules applied when a hit is accepted (ancestor-AABB gate, DFS-rank ties), so scenes
    // the MFMA prefilter can take run on the list kernel in BVH mode as well
    const uint32_t n_pad = scan_pad(t.n_spheres);
    const bool mfma_fits = (v & (kVarScanFromHbm | kVarExactScan)) == 0 && n_pad * 16u <= 64u * 1024u && t.n_tiles > 0 && t.n_tiles <= kMaxMfmaTiles;
    const bool bvh = (ref_bvh && !(mfma_fits && (v & kVarBvhOnTree) == 0)) || list_tree;   // kernel flavour: tree traversal
    // Sphere + MovingSphere worlds: MOVING instantiations exist for the MFMA list kernel and the tree kernels, and their swept
    // bounds cover ray times in [time_lo, time_hi] only; anything else is traced by the general kernel (the records ride along)
    const bool moving = t.has_motion && shutter_inside(t, time0, time1) && (bvh || mfma_fits) && (v & kVarGeneralMoving) == 0;
    if (t.has_motion && !moving) {
        select_world(t, p, local_rows, k, c);
        return;
    }
    c = KernelChoice{};
    c.ref_bvh = ref_bvh;
    c.moving = moving;
    c.verify = (v & kVarVerify) != 0;
    bool sph_lds = false;
    if (!bvh) {
        c.sph_bytes = n_pad * 16u;
        sph_lds = (v & kVarScanFromHbm) == 0 && c.sph_bytes <= 64u * 1024u;
        if (!sph_lds) c.sph_bytes = 0;
    }
    const bool mfma = !bvh && sph_lds && t.n_tiles > 0 && t.n_tiles <= kMaxMfmaTiles && (v & kVarExactScan) == 0;
    c.n_tiles = mfma ? t.n_tiles : 0u;
    c.cull_off = (v & kVarNoCulling) ? 1u : 0u;
    uint32_t lds = c.sph_bytes + kLdsParamBytes;
    if (t.has_noise) lds += 4096u + 768u;
    // 4-wide tree (default; variant bit 2048: the binary tree): its stack entries are 16-bit node numbers
    const bool tree4 = bvh && (v & kVarBinaryTree) == 0 && t.n_nodes4 < 65536u && t.word_ok && t.tree4_packed;
    c.needs_binary_tree = bvh && !tree4;
    // a visit pushes at most three siblings, and only above the bottom level (3 (depth - 1) entries at most); a bottom node
    // still WRITES its three slots (uncounted), hence + 3
    c.bvh_stack_entries = tree4 ? (3u * (t.depth4 ? t.depth4 - 1u : 0u) + 3u) : (t.bin_depth + 2u);
    if (bvh) lds += c.bvh_stack_entries * (uint32_t)kBlock * (tree4 ? 2u : 4u);
    if (tree4) lds += tree4_queue_bytes((uint32_t)kBlock);
    const bool grid = tree4 && t.grid_ok && (v & kVarNoGrid) == 0;
    if (grid) lds += grid_park_bytes((uint32_t)kBlock);
    // binary-tree nodes go to LDS only while FOUR workgroups still fit on the CU (with two levels of attenuation stack each)
    c.nodes_in_lds = (bvh && !tree4 && (v & kVarScanFromHbm) == 0 && lds + t.bin_nodes * 64u + 2u * 3u * (uint32_t)kBlock * 4u <= kLdsBudget / 4u) ? 1u : 0u;
    if (c.nodes_in_lds) lds += t.bin_nodes * 64u;
    // Workgroup size. The MFMA list kernels run ONE 1024- (or 768-) thread workgroup per CU when everything fits: 16-bit palette
    // codes on the attenuation stack + the shading records (+ a BVH world's gates and ranks, + MovingSphere records) in LDS
    const uint32_t stack_levels = p.max_depth > 1u ? p.max_depth - 1u : 1u;
    uint32_t blk = (uint32_t)kBlock;
    const auto wide_extra = [&](uint32_t b) {
        return ((uint64_t)t.n_spheres + 1ull) * 64ull + (((uint64_t)stack_levels * 2ull * b + 15ull) & ~15ull) +
               (ref_bvh ? (uint64_t)t.n_spheres * 32ull + (((uint64_t)t.n_spheres * 4ull + 15ull) & ~15ull) : 0ull) +
               (moving ? (uint64_t)t.n_spheres * 32ull : 0ull);
    };
    const uint32_t mfma_tables = t.n_tiles * 2048u + ((t.n_tiles * 64u + 15u) & ~15u) + 16u * (uint32_t)kCullCells;
    if (mfma && t.palette_ok && (v & kVarStackInHbm) == 0 && !c.verify && k.blocks_per_cu == 0) {
        const auto wide_lds = [&](uint32_t b) { return (uint64_t)lds + mfma_queue_bytes(b) + mfma_tables + wide_extra(b); };
        if (wide_lds(1024u) <= kLdsBudget) blk = 1024u;
        else if (wide_lds(kWideBlock) <= kLdsBudget) blk = kWideBlock;
    }
    const bool wide = blk != (uint32_t)kBlock;
    if (!bvh) lds += mfma ? mfma_queue_bytes(blk) : scan_queue_bytes(blk);
    if (mfma) lds += mfma_tables;
    const uint32_t slots = tree4 ? 1u : 3u;   // attenuation-stack slots per level (4-wide tree kernels: one word)
    const uint64_t path_bytes = (uint64_t)stack_levels * slots * blk * 4ull;
    // Stack levels kept in LDS. Wide kernels: all of them (palette codes, accounted below). Tree kernels run four workgroups
    // per CU: as many levels as fit next to four of them, deeper ones in HBM/L2. 256-thread MFMA kernels: none (their LDS goes
    // to the fragments; 3 resident workgroups beat 1 with an LDS stack). Exact-scan kernels: all or nothing.
    uint32_t lds_levels = 0;
    bool scan4 = false;
    if (wide) {
        lds_levels = 0;
    } else if (bvh && (v & kVarStackInHbm) == 0) {
        const uint32_t wg_regs = tree4 ? tree4_wg_by_regs : 4u;
        const uint32_t per_block = kLdsBudget / std::min(tree4 ? (uint32_t)PT_TREE4_WAVES : 4u, std::max(1u, wg_regs));
        if (per_block > lds) lds_levels = std::min<uint32_t>(stack_levels, (per_block - lds) / (slots * blk * 4u));
    } else if (!bvh && !mfma && (v & kVarStackInHbm) == 0) {
        // Exact-scan kernels (108 VGPRs: four waves per SIMD): FOUR workgroups per CU when all levels -- or all but the deepest two,
        // which few paths reach and which then live in HBM/L2 -- fit beside four of them (smallpt +10 %, small +7 % over three);
        // otherwise all levels or none, three workgroups
        const uint32_t per4 = kLdsBudget / 4u, level_bytes = slots * blk * 4u;
        const uint32_t fit4 = per4 > lds ? std::min<uint32_t>(stack_levels, (per4 - lds) / level_bytes) : 0u;
        if (k.blocks_per_cu == 0 && fit4 + 2u >= stack_levels && fit4 != 0u) lds_levels = fit4, scan4 = true;
        else if (lds + path_bytes <= kLdsPerBlockMax) lds_levels = stack_levels;
    }
    c.stack_in_lds = lds_levels * slots;
    lds += lds_levels * slots * blk * 4u;
    if (wide) lds += (uint32_t)wide_extra(blk);
    c.gstack = !wide && lds_levels < stack_levels;
    // 1024-thread frame kernels: a pool of ready-to-start pixels per wave (48 bytes an entry; pt_kernel.h POOL) where the LDS left over holds at
    // least kPoolSlotsMin entries per wave; otherwise (and for the 768-thread kernels, and under kVarNoPool) the kernel that batches its refills.
    // A power of two, so that a claim is whole rows of one 8x8 work tile: 12, 20 or 43 entries measured 2-4 % SLOWER than 8, 16 or 32.
    // ... and only frames with more than two pixels per lane of an MI355X (256 CUs x 1024): below that nearly every pixel is handed out by the
    // waves' static first claims and the kernel without the pool code is the faster one by 1-2 % (the shards of a multi-GPU frame)
    const uint64_t frame_items = (uint64_t)work_tiles(p.width, local_rows) * kTilePix;
    if (blk == 1024u && !c.verify && (v & kVarNoPool) == 0 && k.pool != 0 && (frame_items > 2ull * 256ull * 1024ull || k.pool > 0)) {
        const uint32_t waves = blk / 64u, room = kLdsBudget > lds ? (kLdsBudget - lds) / (48u * waves) : 0u;
        uint32_t slots = std::min<uint32_t>(std::min<uint32_t>(k.pool > 0 ? (uint32_t)k.pool : kPoolSlots, 64u), room);
        if (k.pool <= 0) while (slots & (slots - 1u)) slots &= slots - 1u;
        if (slots >= kPoolSlotsMin || (k.pool > 0 && slots > 0u)) {
            c.pool_slots = slots, c.pool_off = lds;
            lds += slots * 48u * waves;
        }
    }
    c.lds_bytes = lds;
    c.block = blk;
    c.family = bvh ? (tree4 ? Family::Tree4 : Family::TreeBinary) : (mfma ? Family::Mfma : (sph_lds ? Family::ScanLds : Family::ScanHbm));
    c.gate = mfma && ref_bvh;
    c.grid = grid;
    // cooperative hand-over: the wide frame kernels; a path's attenuations live one level per lane there (depth <= 64), a lane's spheres in 8 register sets (<= 512 spheres)
    c.coop = wide && !c.verify && p.max_depth <= 64u && t.n_spheres <= 512u && (v & kVarNoCoop) == 0;
    // refills are batched: 4 waiting lanes for long pixels, 8 when pixels are short (< 32 spp); 16-wave workgroups batch harder
    c.refill_min = p.samples < 32u ? 8u : 4u;
    if (blk == 1024u && p.samples >= 32u) c.refill_min = 12u;
    // (kernels with pixel pools batch the same way once the list is nearly empty -- or from the start, when a frame has about a pixel per lane)
    if (k.refill >= 0) c.refill_min = (uint32_t)k.refill;
    // persistent grid: CUs x resident workgroups
    uint32_t bpc = k.blocks_per_cu;
    c.bpc_forced = bpc != 0;
    if (bpc == 0) bpc = wide ? 1u : (tree4 ? (uint32_t)PT_TREE4_WAVES : ((bvh || scan4) ? 4u : 3u));
    const uint32_t lds_limit = lds ? (kLdsBudget / lds) : 8u;
    if (bpc > lds_limit) bpc = lds_limit ? lds_limit : 1u;
    c.bpc = std::min(bpc, 8u);
    // the ScanHbm kernel has no measuring twin; everything else orders its work heavy-first when the frame is big enough
    c.order = c.family == Family::ScanHbm ? Order::Natural : order_for(work_tiles(p.width, local_rows), p.samples, v, c.verify);
}

inline void select_kernel(const SceneTraits &t, const pt_params &p, float time0, float time1, uint32_t local_rows, const Knobs &k,
                          uint32_t tree4_wg_by_regs, KernelChoice &c) {
    if (t.is_world) select_world(t, p, local_rows, k, c);
    else select_spheres(t, p, time0, time1, local_rows, k, tree4_wg_by_regs, c);
}

}  // namespace ptsel
