// pt_kernel.h -- persistent-threads path-tracing kernel for gfx950 (MI355X).
//
// One pixel per lane; every lane owns its pixel's xoshiro256+ stream
// (scene.rs:96-102) and walks samples and bounces iteratively. When a path
// terminates the lane regenerates the next camera ray in place; when the
// pixel's samples are exhausted the lane pulls the next pixel from a global
// work counter (one wave-aggregated atomic per refill).
//
// Closest hit (the pieces live in their own headers):
//   pt_sphere.h     Sphere::ray_hit in its exact forms; the exact VALU scan of a HitableList
//   pt_prefilter.h  list worlds on the wide kernels: MFMA prefilter (64 rays x 32 spheres x K = 32 per tile), tile culling, balanced
//                   exact phase 2; the ancestor-AABB gate / DFS-rank rule that makes the same kernel serve BVH worlds
//   pt_tree.h       the internal trees: binary (variant), and the 4-wide packed tree with whole-wave work-sharing traversal
//   pt_grid.h       the uniform cell grid (3D-DDA) the tree kernels walk instead when the scene has one
//   pt_texture.h    Texture::value, Perlin noise, wave-balanced turbulence
//   pt_coop.h       the wave-cooperative mode (one pixel per wave) behind the hand-over of the wide kernels
// DESIGN.md section 4 has the derivations and the error budget.
#pragma once
#ifndef PT_REJ_CAP
#define PT_REJ_CAP 3   // tries of the shared rejection loop per trip before a lane puts the rest off to the next trip (0: until every lane is done; NOTES.md has 2, 3, 4 and 6 measured)
#endif
#include "pt_args.h"
#include "pt_device.h"
#include "pt_tree4.h"
#include "ptgpu.h"

#include "pt_texture.h"
#include "pt_sphere.h"
#include "pt_prefilter.h"   // (also: cross-lane helpers, gates and accept rules the tree kernels share)
#include "pt_tree.h"
#include "pt_grid.h"    // the uniform cell grid the tree kernels walk when the scene has one
#include "pt_coop.h"   // the wave-cooperative mode of the wide list kernels (one pixel per wave), built from the pieces above
namespace ptdev {

// SPH_LDS: list-mode sphere scan reads the (cx,cy,cz,r^2) table from LDS
// (staged once per workgroup); otherwise from HBM/L2 through wave-uniform loads.
// PILOT: the measuring launch that precedes the frame kernel of a new view (own symbol so profiles keep the two apart; A.phase == 1):
// the FIRST sample of every pixel, for real -- it parks each pixel's RNG stream and colour sum for the frame kernel (A.phase == 2)
// and counts the rays per tile.
// MOVING: the world also holds MovingSphere entries (moving_sphere.rs): rays keep their time (camera.rs:59) and
// every exact sphere test / normal uses the centre at that time; prefilter fragments and internal-tree boxes
// were built over the motion's whole sweep.
// GATE: a BVH world on the MFMA list kernel (ancestor-AABB gate + DFS-rank ties at hit acceptance).
// GRID: the 4-wide tree kernels' flavour that walks the scene's uniform cell grid (pt_grid.h) instead of the tree.
// NOPOOL: a 1024-thread frame kernel without the per-wave pixel pools (see POOL below).
// BLK: threads per workgroup. 256 (three workgroups per CU) everywhere except the MFMA list kernels, which run ONE
// 768-thread workgroup per CU when the scene allows: the sphere fragments are then staged once per CU instead of three
// times, and the LDS that frees holds the per-lane attenuation stacks (no HBM traffic for them).
//
// MAP of the kernel body (search for the quoted banner):
//   prologue      LDS carve, staging of tables / fragments / frame parameters, per-lane state
//   main loop, one trip = one ray per live lane ("for (;;)"), its ONLY exit at the very end:
//     "---- refill"                 finished pixels are written, free lanes claim pixels (batched), parked streams are reloaded
//     "TAIL: once the list is dry"  a look into one other wave's mailbox (pt_coop.h), read at the end of the trip
//     "---- camera.rs:56-68"        next sample's camera ray + the sphere draws a Metal scatter still owes, one shared rejection loop
//     "---- hitable.rs:39-65"       closest hit: 4-wide tree (bvh4_trace) | binary tree | MFMA prefilter + balanced exact tests | exact scan
//     "---- scene.rs:49-71"         one level of ray_trace: material, scatter, attenuation push; on termination fold + sample count
//     "Hand-over (pt_coop.h)"       a pixel between two samples goes to an idle worker
//   epilogue      the wave retires or becomes a worker (coop_worker), ray count reduction, development counters
// The stages share ~40 loop-carried registers per lane and are kept in one function body on purpose: every attempt to carry that state
// through a struct or across call boundaries has cost registers in a kernel that lives at its 128-VGPR limit (NOTES.md).
#ifdef PT_BBPROF   // tools/bbprof.py: the instrumented assembly keeps its counter registers above the compiler's
#include "pt_bbprof.h"
#define PT_BBPROF_ATTR __attribute__((amdgpu_num_sgpr(100)))
#else
#define PT_BBPROF_ATTR
#endif
typedef const KArgs __attribute__((address_space(4))) *KArgsK;   // the kernel's argument block where it lives: the kernarg segment
template <bool BVH, bool SPH_LDS, bool MFMA, bool VERIFY, bool PILOT, bool MOVING = false, bool GATE = false, int BLK = kBlock, bool GRID = false, bool NOPOOL = false>
__global__ __launch_bounds__(BLK, (BLK == kBlock) ? ((BVH && SPH_LDS) ? PT_TREE4_WAVES : PT_MINWAVES) : 1) PT_BBPROF_ATTR void pt_trace_kernel(const KArgs A) {
    static_assert(BLK == kBlock || (MFMA && !BVH), "only the MFMA list kernels take another workgroup size");
    static_assert(!GRID || (BVH && SPH_LDS), "the uniform cell grid (pt_grid.h) is a traversal structure of the 4-wide tree kernels");
    constexpr bool TREE4 = BVH && SPH_LDS;   // tree kernels: SPH_LDS selects the 4-wide tree (false: the binary one, variant bit 2048)
    // Wide (one workgroup per CU) MFMA kernels: every attenuation a path can pick up is one of a finite PALETTE -- a sphere's
    // constant / metal albedo, one of its two checker colours, or white (dielectric) -- so the per-lane attenuation stack
    // holds 16-bit codes (sphere index | even-checker bit; 0x7fff = white) instead of three floats, and the fold reads the
    // colours back from the per-sphere shading records, which live in LDS here. 18 instead of 108 bytes of LDS per lane.
    // (launch() only picks a wide kernel for scenes whose textures are all Constant or Checker-of-two-Constants.)
    constexpr bool PAL = (BLK != kBlock);
    // The wide frame kernels hand pixels over to waves that have run out of work (pt_coop.h); A.tail_cap == 0 switches it off.
    constexpr bool TAIL = PAL && !PILOT && !VERIFY;
    // The wide frame kernels keep a per-wave POOL of ready-to-start pixels in LDS (A.pool_slots entries of 48 bytes: RNG stream, colour sum,
    // coordinates): a lane that finishes a pixel takes the next one from it in the same trip, and the global round trips of a claim (work
    // counter -> tile order -> parked stream) are paid once per pool_slots pixels by the whole wave instead of making freed lanes wait for
    // a batch ("---- refill" below; A.pool_slots == 0: the batched refill of the other kernels).
    // (the 1024-thread frame kernels; NOPOOL = true keeps round 5's batched refill for scenes whose LDS has no room for pools and for the tuning bit
    //  kVarNoPool. 768-thread kernels -- three waves per SIMD at up to 149 registers -- measured 1.6 % slower with pools: they keep the batch.)
    constexpr bool POOL = TAIL && BLK == 1024 && !NOPOOL;
    static_assert(!NOPOOL || (TAIL && BLK == 1024), "NOPOOL only distinguishes the 1024-thread frame kernels");
    // 4-wide tree kernels: ONE 32-bit word per attenuation-stack level -- the grey value of a Noise texture as its float bits
    // (texture.rs:86-89 yields (v, v, v)), or a palette code like PAL's for everything else: 0xFFE00000 | even-checker bit << 20 |
    // record index (0xFFFFF = white). No arithmetic produces such a NaN pattern (canonical NaNs are 0x7FC00000 / 0xFFC00000).
    // A quarter of the LDS (and of the HBM traffic of the levels that do not fit) of three floats, and one register instead
    // of three for the first bounce. (launch() only walks this tree for scenes whose textures are Noise, Constant or
    // Checker-of-two-Constants; others use the binary tree kernel with the float stack.)
    constexpr bool WST = TREE4;
    constexpr uint32_t kWstCode = 0xFFE00000u, kWstWhite = kWstCode | 0xFFFFFu;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS carve (all offsets multiples of 16)
    float4 *s_sph = reinterpret_cast<float4 *>(smem);  // list mode: n_spheres x (cx,cy,cz,r^2)
    unsigned char *p = smem + A.lds_sphere_bytes;
    float4 *s_pvec = reinterpret_cast<float4 *>(p);     // perlin gradients (4 KB) when has_noise
    uint8_t *s_perm = p + (A.has_noise ? 4096 : 0);
    p += A.has_noise ? (4096 + 768) : 0;
    uint32_t *s_bvh = reinterpret_cast<uint32_t *>(p);
    p += BVH ? (A.bvh_stack_entries * BLK * (TREE4 ? 2 : 4)) : 0;   // (4-wide tree: 16-bit entries, an even number of them)
    DWideNode *s_nodes = reinterpret_cast<DWideNode *>(p);
    p += (BVH && A.nodes_in_lds) ? A.n_nodes * 64u : 0u;
    uint16_t *s_queue = reinterpret_cast<uint16_t *>(p);  // exact scan: [kQueueCap+1][BLK] u16; MFMA: [kEntCap][BLK] u32 tile masks + per-wave pair lists
    //                                                   4-wide tree: [kLeafQ][BLK] u32 leaf candidates + per-wave pair lists
    uint32_t *w_pairs = reinterpret_cast<uint32_t *>(p + (TREE4 ? kLeafQ : kEntCap) * BLK * 4 + (threadIdx.x >> 6) * kWavePairBytes);
    unsigned long long *w_keys = reinterpret_cast<unsigned long long *>(w_pairs + kPairCap);
    p += BVH ? (TREE4 ? tree4_queue_bytes(BLK) : 0u) : (MFMA ? mfma_queue_bytes(BLK) : scan_queue_bytes(BLK));
    uint32_t *w_park = reinterpret_cast<uint32_t *>(p) + (threadIdx.x >> 6) * (kGridParkMax * kGridParkWords);   // GRID: this wave's parked walks (pt_grid.h)
    p += GRID ? grid_park_bytes(BLK) : 0u;
    uint4 *s_afrag = reinterpret_cast<uint4 *>(p);        // MFMA: [n_tiles][2][64] x 16 B
    p += MFMA ? A.n_tiles * 2048u : 0u;
    uint16_t *s_tile_sphere = reinterpret_cast<uint16_t *>(p);
    p += MFMA ? ((A.n_tiles * 64u + 15u) & ~15u) : 0u;
    uint32_t *s_cull = reinterpret_cast<uint32_t *>(p);   // MFMA: tile-culling tables, 2 axes x 2 x kCullCells words
    p += MFMA ? 16u * kCullCells : 0u;

    const float4 *s_par = reinterpret_cast<const float4 *>(p);   // frame parameters (kLdsParamBytes)
    p += kLdsParamBytes;
    float4 *s_shade = reinterpret_cast<float4 *>(p);    // PAL: the per-sphere shading records (64 B each)
    p += PAL ? (A.n_spheres + 1u) * 64u : 0u;
    float4 *s_gate = reinterpret_cast<float4 *>(p);     // PAL && GATE: gate boxes (2 float4 per sphere) and DFS ranks of a BVH world
    p += (PAL && GATE) ? A.n_spheres * 32u : 0u;
    uint32_t *s_rank = reinterpret_cast<uint32_t *>(p);
    p += (PAL && GATE) ? ((A.n_spheres * 4u + 15u) & ~15u) : 0u;
    float4 *s_motion = reinterpret_cast<float4 *>(p);   // PAL && MOVING: the MovingSphere records (2 float4 per sphere)
    p += (PAL && MOVING) ? A.n_spheres * 32u : 0u;
    const float4 *const mot = (PAL && MOVING) ? (const float4 *)s_motion : A.motion;
    float *s_path = reinterpret_cast<float *>(p);       // [max_depth][3][BLK] attenuation stack (PAL: u16 [max_depth][BLK] palette codes)
    uint16_t *s_pal = reinterpret_cast<uint16_t *>(p);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t wave_id = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: lives in a scalar register

    if (!BVH && SPH_LDS) {
        for (uint32_t k = tid; k < A.n_spheres_pad; k += BLK) s_sph[k] = A.spheres_r2[k];
    }
    if (MFMA) {
        for (uint32_t k = tid; k < A.n_tiles * 128u; k += BLK) s_afrag[k] = A.afrag[k];
        for (uint32_t k = tid; k < A.n_tiles * 32u; k += BLK) s_tile_sphere[k] = A.tile_sphere[k];
        if (A.cull_axis < 3u)
            for (uint32_t k = tid; k < 4u * kCullCells; k += BLK) s_cull[k] = A.cull_tab[k];
    }
    if (PAL) {
        for (uint32_t k = tid; k < A.n_spheres * 4u; k += BLK) s_shade[k] = A.shade[k];
        if (tid < 4) s_shade[A.n_spheres * 4u + tid] = make_float4(1.f, 1.f, 1.f, 0.f);   // the white entry (Dielectric, material.rs:117)
        if (GATE) {
            for (uint32_t k = tid; k < A.n_spheres * 2u; k += BLK) s_gate[k] = A.gate[k];
            for (uint32_t k = tid; k < A.n_spheres; k += BLK) s_rank[k] = A.leaf_rank[k];
        }
        if (MOVING)
            for (uint32_t k = tid; k < A.n_spheres * 2u; k += BLK) s_motion[k] = A.motion[k];
    }
    if (BVH && A.nodes_in_lds) {
        const uint4 *src = reinterpret_cast<const uint4 *>(A.wnodes);
        uint4 *dst = reinterpret_cast<uint4 *>(s_nodes);
        for (uint32_t k = tid; k < A.n_nodes * 4u; k += BLK) dst[k] = src[k];
    }
    if (tid == 0) {
        float4 *w = const_cast<float4 *>(s_par);
        w[0] = make_float4(A.clip_min[0], A.clip_min[1], A.clip_min[2], A.cull_u0);
        w[1] = make_float4(A.clip_max[0], A.clip_max[1], A.clip_max[2], A.cull_inv_cell);
        w[2] = make_float4(A.c0[0], A.c0[1], A.c0[2], A.rs2);
        w[3] = make_float4(A.m0, A.gamma, A.inv_nx, A.inv_ny);
        w[4] = make_float4(A.cam.origin.x, A.cam.origin.y, A.cam.origin.z, A.cam.lower_left_corner.x);
        w[5] = make_float4(A.cam.lower_left_corner.y, A.cam.lower_left_corner.z, A.cam.horizontal.x, A.cam.horizontal.y);
        w[6] = make_float4(A.cam.horizontal.z, A.cam.vertical.x, A.cam.vertical.y, A.cam.vertical.z);
        w[7] = make_float4(A.cam.u.x, A.cam.u.y, A.cam.u.z, A.cam.v.x);
        w[8] = make_float4(A.cam.v.y, A.cam.v.z, A.cam.w.x, A.cam.w.y);
        w[9] = make_float4(A.cam.w.z, A.cam.time0, A.cam.time1, A.cam.lens_radius);
        w[10] = make_float4(A.inv_ns, A.mix_prev, A.mix_new, A.prev_zero ? 1.0f : 0.0f);
        w[11] = make_float4(A.sky.x, A.sky.y, A.sky.z, A.has_sky ? 1.0f : 0.0f);
        w[12] = make_float4(__uint_as_float(A.cull_axis), __uint_as_float(A.cull_always), __uint_as_float(A.max_depth), __uint_as_float(A.samples));
        w[13] = make_float4(A.cull_reach[0], A.cull_reach[1], A.cull_reach[2], 0.0f);
        w[14] = make_float4(A.cull_u0_2, A.cull_inv_cell_2, __uint_as_float(A.cull_axis2), 0.0f);
    }
    if (A.has_noise) {
        for (int k = tid; k < 256; k += BLK) s_pvec[k] = A.perlin_vec[k];
        for (int k = tid; k < 768; k += BLK) s_perm[k] = (uint8_t)A.perlin_perm[k];
    }
    __syncthreads();
    if (TAIL && A.tail_cap != 0u && tid == 0 && PT_COOP_DBG(4u)) atomicAdd(&A.work_counter[kCtlStarted], 1u);   // (counted per workgroup: pt_coop.h "end")
    if (TAIL && kCoopBusyPrio != 0 && A.tail_cap != 0u) __builtin_amdgcn_s_setprio(kCoopBusyPrio);   // (pt_coop.h: workers drop to 0)

    PerlinLds pn{s_pvec, s_perm, false};
    // attenuation stack of this lane: the 768-thread kernels always have it in LDS; the 256-thread ones keep the first
    // A.stack_in_lds slots (3 per level) in LDS and deeper, rarely reached levels in HBM/L2 (what fits next to four
    // resident workgroups of a tree kernel)
    float *gpath = A.gstack + (size_t)blockIdx.x * A.max_depth * 3 * BLK + tid;
    auto path_ld = [&](uint32_t slot) -> float {
        if (BLK != kBlock || slot < A.stack_in_lds) return s_path[slot * BLK + tid];
        return gpath[slot * BLK];
    };
    auto path_st = [&](uint32_t slot, float v) {
        if (BLK != kBlock || slot < A.stack_in_lds)
            s_path[slot * BLK + tid] = v;
        else
            gpath[slot * BLK] = v;
    };

#ifdef PT_SECTIONS
    // development aid (-DPT_SECTIONS): per-wave cycle counts of the main loop's sections (s_memtime deltas at
    // wave-uniform points), summed into debug[16..23]; ptgpu.hip prints the shares after each launch
    unsigned long long sec_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sec_last = __builtin_readcyclecounter();
#define PT_SEC(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); sec_t[i] += now_ - sec_last; sec_last = now_; } while (0)
#else
#define PT_SEC(i) do { } while (0)
#endif
    bool have = false, exhausted = false, need_cam = true, trav_new = false, finished = false;
    bool pend_metal = false;    // the lane's Metal scatter waits for its sphere sample (drawn with the next camera rays' lens samples)
    float metal_fuzz = 0.0f;
    uint32_t pix_rays = 0;
    BvhTrav trav{0u, 0u, 0, kMaxT, -1, 0u, false};
    Steal4 steal4{0u, 0u};
    unsigned long long grid_parked = 0ull;   // GRID, wave-uniform: lanes whose walk is parked in w_park (pt_grid.h), to be resumed by the next call
    // per-lane bookkeeping, packed (every register counts: the 4-waves-per-SIMD kernels are compiled for 128 VGPRs):
    //   pxy = pixel column | local row << 16 (launch() keeps width and height below 65536)
    //   sd  = bounce depth (12 bits) | sample number << 12 (launch(): max_depth <= 4095, samples < 2^20)
    uint32_t pxy = 0, sd = 0;
    unsigned long long wave_rays = 0;   // scene.rs:57 ray_count of this WAVE (wave-uniform: lives in scalar registers)
#define PT_DEPTH (sd & 0xfffu)
    Rng rng{0, 0, 0, 0};
    f3 col = mk3(0.f, 0.f, 0.f), o = mk3(0.f, 0.f, 0.f), d = mk3(0.f, 0.f, 0.f);
    f3 att0 = mk3(1.f, 1.f, 1.f);   // attenuation of the first bounce (deeper ones: the stack)
    uint32_t att0c = 0u;            // PAL: its palette code; WST: its word
    const float4 *shade = PAL ? (const float4 *)s_shade : ((TREE4 && !MOVING && A.gate) ? A.shade_rank : A.shade);
    const uint32_t kWhite = A.n_spheres;   // PAL: code of (1, 1, 1): one extra record behind the spheres'
    auto word_colour = [&](uint32_t w) -> f3 {   // WST: the attenuation behind a stack word
        if ((w & kWstCode) != kWstCode) {
            const float v = __uint_as_float(w);
            return mk3(v, v, v);
        }
        if (w == kWstWhite) return mk3(1.f, 1.f, 1.f);
        const float4 q = shade[4u * (w & 0xFFFFFu) + 2u + ((w >> 20) & 1u)];
        return mk3(q.x, q.y, q.z);
    };
    auto palette_colour = [&](uint32_t code) -> f3 {   // PAL: the colour behind a stack entry
        const float4 q = s_shade[4u * (code & 0x7fffu) + 2u + (code >> 15)];
        return mk3(q.x, q.y, q.z);
    };
    float rtime = 0.f;  // ray.time (only MOVING kernels read it)
    bool first_claim = true;   // (wave-uniform: every lane of a wave takes part in its first fetch)
#ifdef PT_CAMSKIP
    bool cam_skipped = false;   // (wave-uniform)
#endif
    // POOL, one wave-uniform word: bits 0-6 the pool's first live entry, 7-13 how many are live, 14-20 items left of the wave's 64 static first
    // ones, 21 the work list has nothing more for this wave, 22-31 the wave's FAIR SHARE of what the list still held at its last claim
    // (items left / waves of the grid, capped at 1023): a claim parks at most half of it, so the pools never hold more than half of what is
    // left -- an item parked here is one an idle lane elsewhere cannot take -- and the claims shrink with the list until they take what is
    // wanted and no more (a frame of two pixels per lane kept 60 % of its dynamic items parked under a fixed claim size: +5 % frame time)
    uint32_t pool_st = (POOL && A.first_static != 0u) ? (64u << 14) : 0u;
    if (POOL) {
        const uint32_t share0 = __umulhi(A.n_items - A.first_static, A.pool_waves_magic);
        pool_st |= (share0 < 1023u ? share0 : 1023u) << 22;
    }
#define PT_POOL_LEFT ((pool_st >> 7) & 127u)
#define PT_POOL_DRY ((pool_st >> 21) & 1u)
#define PT_POOL_SHARE (pool_st >> 22)
#define PT_POOL_EXACT (PT_POOL_SHARE < A.pool_tail)
    bool tail_dry = TAIL && A.tail_dry0 != 0u;     // TAIL, wave-uniform: the work list has run dry (from then on pixels may be handed over)
    uint32_t tail_it = 0, tail_streak = 0, tail_rand = (blockIdx.x * (BLK / 64) + wave_id) * 2654435761u + 12345u;   // (wave-uniform)
#ifdef PT_DEVKNOBS
    if (A.wave_end && lane == 0) atomicMin(&A.wave_end[65535], (unsigned long long)wall_clock64());   // (the launch's first wave: origin of the hand-over log's timeline)
#endif
#if defined(PT_SECTIONS) || defined(PT_WAVEDBG)
#define PT_WAVE_DETAIL 1   // development builds: per-wave iteration counts, first / last pixel, moment the work list ran dry
    uint32_t dbg_first_pxy = 0xffffffffu;
#ifdef PT_WAVEDBG
    unsigned long long dbgc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    unsigned long long dbg_exh_iter = 0, dbg_iters = 0, dbg_last_refill = 0, dbg_start = A.wave_end ? wall_clock64() : 0ull;   // (PTGPU_TIMING)
#endif

    for (;;) {
        // ---- refill: lanes without a pixel get one
        // (this section reads its arguments -- frame buffer, parked streams, work order, counters, frame geometry: two dozen scalars it needs once per
        //  pixel -- through RF(field): in the frame kernels the kernel's argument block behind an OPAQUE pointer to the kernarg segment, so that they are
        //  loaded where they are used instead of being kept in scalar registers across the whole loop, which left the hot loops reading spilled scalars
        //  back with v_readlane (headline kernel: 77 -> 35 spilled SGPRs, +2 %). The measuring and verify twins read the plain block.)
        constexpr bool kOpaqueArgs = !PILOT && !VERIFY;
        const KArgsK Rp = [] { KArgsK q = (KArgsK)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(q)); return q; }();
#define RF(field) (kOpaqueArgs ? Rp->field : A.field)
        const unsigned long long want = wave_ballot(!have && !exhausted);
        // scene.rs:113-116: a finished pixel is averaged, blended into the frame and its rays are booked on its work tile
        auto write_finished_pixel = [&]() {
            finished = false;
            const float4 pf = s_par[10];   // inv_ns, mix_prev, mix_new
            if (PILOT && RF(phase) == 1u) {   // to be continued: park the stream and the sum
                uint4 *st = RF(px_state) + 3u * (size_t)((pxy >> 16) * RF(width) + (pxy & 0xffffu));
                st[0] = make_uint4((uint32_t)rng.s0, (uint32_t)(rng.s0 >> 32), (uint32_t)rng.s1, (uint32_t)(rng.s1 >> 32));
                st[1] = make_uint4((uint32_t)rng.s2, (uint32_t)(rng.s2 >> 32), (uint32_t)rng.s3, (uint32_t)(rng.s3 >> 32));
                st[2] = make_uint4(__float_as_uint(col.x), __float_as_uint(col.y), __float_as_uint(col.z), 0u);
            }
            col = scale3(col, pf.x);
            if (!PILOT) {
                float *out = RF(rgb) + ((pxy >> 16) * RF(width) + (pxy & 0xffffu)) * 3u;
                // (prev_zero: pt_render found the host buffer all +0.0f and did not upload it -- same products, same sums)
                const bool pz = pf.w != 0.0f;   // (KArgs::prev_zero, through the LDS parameter block like its neighbours)
                const float p0 = pz ? 0.0f : out[0], p1 = pz ? 0.0f : out[1], p2 = pz ? 0.0f : out[2];
                out[0] = p0 * pf.y + col.x * pf.z;
                out[1] = p1 * pf.y + col.y * pf.z;
                out[2] = p2 * pf.y + col.z * pf.z;
            }
            // (frame kernels: the NEXT frame's work order; the pixel's work tile is recomputed from its coordinates)
            if (PILOT || RF(tile_cost)) atomicAdd(&RF(tile_cost)[((pxy >> 16) >> kTileLog2) * RF(tiles_x) + ((pxy & 0xffffu) >> kTileLog2)], pix_rays);
        };
        // the sample number a pixel starts this launch with. Phase 2 of a frame whose measuring launch traced every OTHER tile (KArgs::checker):
        // a pixel of an unmeasured tile starts here, one sample behind the others -- its sample number starts at -1 (20 bits), so that it too
        // is done when the number reaches s_par[12].w = samples - 1
        auto tile_parked = [&](uint32_t tcol, uint32_t trow) -> bool { return !PILOT && RF(phase) == 2u && (RF(checker) == 0u || ((tcol + trow) & 1u) == 0u); };
        auto start_sd = [&](bool parked) -> uint32_t { return (!PILOT && RF(phase) == 2u && !parked) ? 0xfffff000u : 0u; };
        // work item -> the pixel and the state it starts with (false: the item lies beyond the frame's edge)
        auto start_item = [&](uint32_t item, uint32_t &pxy_o, bool &parked_o, Rng &rng_o, f3 &col_o) -> bool {
            const uint32_t in = item & (kTilePix - 1u);
            const uint32_t tile = RF(tile_order) ? RF(tile_order)[item >> (2u * kTileLog2)] : (item >> (2u * kTileLog2));
            // tile / tiles_x by the host's magic (a u32 division costs ~25 instructions and a hoisted reciprocal register):
            // umulhi underestimates the quotient by at most one for any tile < 2^32
            uint32_t trow = __umulhi(tile, RF(tiles_x_magic)), tcol = tile - trow * RF(tiles_x);
            if (tcol >= RF(tiles_x)) trow += 1u, tcol -= RF(tiles_x);
            const uint32_t x = tcol * kTileSide + (in & (kTileSide - 1u));
            const uint32_t ly = trow * kTileSide + (in >> kTileLog2);
            if (!(x < RF(width) && ly < RF(local_rows))) return false;
            pxy_o = x | (ly << 16);
            const uint32_t px = x, py = ly * RF(shard_count) + RF(shard_index);
            parked_o = tile_parked(tcol, trow);
            if (parked_o) {   // continue the stream and the sum phase 1 parked
                const uint4 *st = RF(px_state) + 3u * (size_t)(ly * RF(width) + x);
                const uint4 a = st[0], b = st[1], c = st[2];
                rng_o.s0 = (uint64_t)a.x | ((uint64_t)a.y << 32), rng_o.s1 = (uint64_t)a.z | ((uint64_t)a.w << 32);
                rng_o.s2 = (uint64_t)b.x | ((uint64_t)b.y << 32), rng_o.s3 = (uint64_t)b.z | ((uint64_t)b.w << 32);
                col_o = mk3(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z));
            } else {
                col_o = mk3(0.f, 0.f, 0.f);
                // scene.rs:96-102
                uint64_t seed = ((uint64_t)px * 1973ull + (uint64_t)py * 9277ull + (uint64_t)RF(frame_num) * 26699ull) | 1ull;
                if (RF(random_seed)) {
                    uint64_t h = RF(seed_base) ^ (seed * 0x9e3779b97f4a7c15ULL);
                    seed = splitmix64_next(h);
                }
                rng_seed_from_u64(rng_o, seed);
            }
            return true;
        };
        if (POOL) {
            // Wide frame kernels: freed lanes do not wait for a batch. The wave keeps a POOL of ready-to-start pixels in LDS (RF(pool_slots) entries of
            // 48 bytes per wave: RNG stream, colour sum, coordinates); a lane that finishes takes the next entry in the same trip, and the global
            // round trips of a claim (work counter -> tile order -> parked stream) are paid once per pool_slots pixels, by all lanes together. Near
            // the list's end (the wave's fair share of it below RF(pool_tail) items) claims take what is wanted and no more, batched by RF(refill_min) like the other kernels'.
            if (want != 0ull) {   // (wave-uniform)
                if (!have && finished) write_finished_pixel();
                uint32_t wv_here = wave_id;
                asm volatile("" : "+s"(wv_here));   // (the pool's address is formed here: hoisted out of the loop it is one more live scalar)
                uint4 *const w_pool = reinterpret_cast<uint4 *>(smem + RF(pool_off)) + wv_here * (RF(pool_slots) * 3u);
                const uint32_t wn = (uint32_t)__popcll(want);
                if (PT_POOL_LEFT == 0u && !PT_POOL_DRY && (!PT_POOL_EXACT || wn >= RF(refill_min) || wave_ballot(have) == 0ull)) {
                    // ---- claim: the whole wave fetches up to pool_slots work items and parks their start states in its pool
                    uint32_t n = RF(pool_slots), base;
                    const uint32_t static_left = (pool_st >> 14) & 127u;
                    if (static_left != 0u) {
                        // A SIMD's arbiter serves its OLDEST wave first: the first waves of a 16-wave workgroup advance up to twice as fast
                        // as the last ones ("The end of a frame"). The head of the heavy-first list -- the pixels whose
                        // serial sample chains decide when the frame ends -- therefore goes to them: a wave's first 64 items are fixed by
                        // its age class (wave >> 2) instead of by the race for the counter, which starts behind these items.
                        const uint32_t cls = wv_here >> 2, idx = blockIdx.x * 4u + (wv_here & 3u);
                        n = n < static_left ? n : static_left;
                        base = (cls * gridDim.x * 4u + idx) * 64u + (64u - static_left);
                        pool_st -= n << 14;
                    } else {
                        // half the fair share, as a power of two (a claim is then whole rows of one 8x8 work tile), but never less than is wanted now
                        const uint32_t half = PT_POOL_SHARE >> 1;
                        const uint32_t fill = (PT_POOL_EXACT || half == 0u) ? 0u : (0x80000000u >> __builtin_clz(half));
                        const uint32_t want_now = fill > wn ? fill : wn;
                        n = n < want_now ? n : want_now;
                        uint32_t b = 0;
                        if (lane == 0) b = atomicAdd(RF(work_counter), n);
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)b) + RF(first_static);
                    }
                    if (base + n >= RF(n_items)) pool_st |= 1u << 21;
                    if (static_left == 0u) {   // what the list holds now, per wave of the grid
                        const uint32_t share = __umulhi(base + n < RF(n_items) ? RF(n_items) - (base + n) : 0u, RF(pool_waves_magic));
                        pool_st = (pool_st & 0x3fffffu) | ((share < 1023u ? share : 1023u) << 22);
                    }
                    const uint32_t item = base + (uint32_t)lane;
                    uint32_t q = 0;
                    bool parked_q = false;
                    Rng r{0, 0, 0, 0};
                    f3 c = mk3(0.f, 0.f, 0.f);
                    const bool ok = (uint32_t)lane < n && item < RF(n_items) && start_item(item, q, parked_q, r, c);
                    const unsigned long long m = wave_ballot(ok);
                    if (ok) {
                        uint4 *e = w_pool + 3u * __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        e[0] = make_uint4((uint32_t)r.s0, (uint32_t)(r.s0 >> 32), (uint32_t)r.s1, (uint32_t)(r.s1 >> 32));
                        e[1] = make_uint4((uint32_t)r.s2, (uint32_t)(r.s2 >> 32), (uint32_t)r.s3, (uint32_t)(r.s3 >> 32));
                        e[2] = make_uint4(__float_as_uint(c.x), __float_as_uint(c.y), __float_as_uint(c.z), q);
                    }
                    pool_st = (pool_st & ~0x3fffu) | ((uint32_t)__popcll(m) << 7);   // first entry 0, that many live
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                // ---- serve: waiting lanes take the pool's next entries, in order (a lane the pool cannot serve waits for the next claim)
                const uint32_t pool_head = pool_st & 127u, pool_left = PT_POOL_LEFT;
                if (!have && !exhausted) {
                    const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
                    if (rk < pool_left) {
                        const uint4 *e = w_pool + 3u * (pool_head + rk);
                        const uint4 a = e[0], b = e[1], c = e[2];
                        rng.s0 = (uint64_t)a.x | ((uint64_t)a.y << 32), rng.s1 = (uint64_t)a.z | ((uint64_t)a.w << 32);
                        rng.s2 = (uint64_t)b.x | ((uint64_t)b.y << 32), rng.s3 = (uint64_t)b.z | ((uint64_t)b.w << 32);
                        col = mk3(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z));
                        pxy = c.w;
                        sd = start_sd(tile_parked((pxy & 0xffffu) >> kTileLog2, (pxy >> 16) >> kTileLog2));
                        pix_rays = 0;
                        have = true;
                        need_cam = true;
#ifdef PT_WAVE_DETAIL
                        if (dbg_first_pxy == 0xffffffffu) dbg_first_pxy = pxy;
#endif
                    } else if (PT_POOL_DRY) {
                        exhausted = true;
#ifdef PT_WAVE_DETAIL
                        if (RF(wave_end) && dbg_exh_iter == 0) dbg_exh_iter = dbg_iters, dbg_last_refill = wall_clock64();
#endif
                    }
                }
                const uint32_t took = wn < pool_left ? wn : pool_left;
                pool_st += took - (took << 7);   // head += took, left -= took
            }
        } else {
            // ---- the other kernels: one wave-aggregated atomic for all lanes that need a pixel. The refill code (a global atomic
            // round trip and four SplitMix64 steps of 64-bit multiplies) runs for the whole wave whenever ANY lane needs
            // it, so lanes wait until RF(refill_min) of them do (or nobody has work left): fewer, fuller refills.
            const bool refill_now = __popcll(want) >= (int)RF(refill_min) || wave_ballot(have) == 0ull;
            if (!have && !exhausted && refill_now) {
                if (finished) write_finished_pixel();
                const unsigned long long m = wave_ballot(1);
                const int leader = __ffsll((long long)m) - 1;
                uint32_t base = 0;
                if (RF(first_static) != 0u && first_claim) {
                    // (the waves' first 64 items by age class: see the pool's claim above)
                    const uint32_t wv = wave_id, cls = wv >> 2, idx = blockIdx.x * 4u + (wv & 3u);
                    base = (cls * gridDim.x * 4u + idx) * 64u;
                } else {
                    if (lane == leader) base = atomicAdd(RF(work_counter), (uint32_t)__popcll(m));
                    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader) + RF(first_static);
                }
                first_claim = false;
                const uint32_t item = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));   // lanes of m below this one
                if (item >= RF(n_items)) {
                    exhausted = true;
#ifdef PT_WAVE_DETAIL
                    if (RF(wave_end) && dbg_exh_iter == 0) dbg_exh_iter = dbg_iters, dbg_last_refill = wall_clock64();
#endif
                } else {
                    pix_rays = 0;
                    bool parked = false;
                    if (start_item(item, pxy, parked, rng, col)) {
                        have = true;
#ifdef PT_WAVE_DETAIL
                        if (dbg_first_pxy == 0xffffffffu) dbg_first_pxy = pxy;
#endif
                        need_cam = true;
                        sd = start_sd(parked);
                    }
                }
            }
        }
#undef RF
        // TAIL: once the list is dry, every few iterations a look into ONE other wave's mailbox (pt_coop.h): a load of a line nobody
        // else polls, issued here and read at the end of the iteration, so its latency hides behind the iteration's work
        uint64_t tail_probe = 0;
        uint32_t tail_target = 0;
        bool tail_polled = false;   // (wave-uniform)
        if (TAIL && A.tail_cap != 0u) {
            tail_dry = tail_dry || (POOL && PT_POOL_DRY) || wave_any(exhausted);
            tail_polled = tail_dry && PT_COOP_DBG(2u) && ((tail_it & A.tail_period_mask) == 0u || (uint32_t)__popcll(wave_ballot(have)) <= A.tail_live_max);
            if (tail_polled) {
                tail_rand = tail_rand * 1664525u + 1013904223u;
                tail_target = __umulhi(tail_rand, A.tail_cap);
                tail_probe = wt_load(A.tail_box + 16u * (size_t)tail_target + 8);
            }
            tail_it += 1u;
        }
#ifdef PT_WAVE_DETAIL
        if (A.wave_end) dbg_iters += 1;
#endif
        PT_SEC(0);

        // ---- camera.rs:56-68 + scene.rs:107-108: start the next sample -- and, in the same rejection loop, the
        // random_in_unit_sphere a Metal scatter of the LAST iteration still owes (material.rs:77).
        // Both are "draw until the point lies inside the unit ball" loops (math.rs:6-13 in the plane, math.rs:15-26 in space), and
        // a wave runs such a loop as often as its unluckiest lane needs: 2.8 trips for the lens, 3.1 for the metal lanes, one
        // after the other. A lane is in at most one of the two roles here -- a path that scattered off metal continues, so it
        // needs no camera ray -- and its draws keep their order (the sphere's three draws were the lane's next ones anyway), so
        // both loops become ONE whose trips cost the maximum instead of the sum. Same arithmetic per role: x = 2a - 1 etc.,
        // (x x + y y) + z z with z = 0 in the plane, which is the reference's (x x + y y) + 0.
        const bool cam_role = have && need_cam, met_role = have && pend_metal;
#ifdef PT_CAMSKIP
        // (timing experiment, round 5's verdict item 8 -- NOTES.md "camera stage every other trip": when fewer than PT_CAMSKIP lanes need the stage it
        //  is skipped for ONE trip: those lanes idle through the trip -- they hold no finished ray -- and the stage's ~360 instructions are issued less often)
        const bool cam_run = (uint32_t)__popcll(wave_ballot(cam_role || met_role)) >= (uint32_t)(PT_CAMSKIP) || cam_skipped;
        cam_skipped = !cam_run && wave_any(cam_role || met_role);
        if ((cam_role || met_role) && cam_run) {
#else
        if (cam_role || met_role) {
#endif
            const float4 c0 = s_par[4], c1 = s_par[5], c2 = s_par[6], c3 = s_par[7], c4 = s_par[8], c5 = s_par[9], pn2 = s_par[3];
            const f3 cam_origin = mk3(c0.x, c0.y, c0.z), cam_llc = mk3(c0.w, c1.x, c1.y), cam_horizontal = mk3(c1.z, c1.w, c2.x),
                     cam_vertical = mk3(c2.y, c2.z, c2.w), cam_u = mk3(c3.x, c3.y, c3.z), cam_v = mk3(c3.w, c4.x, c4.y);
            const float cam_time0 = c5.y, cam_time1 = c5.z, cam_lens_radius = c5.w;
            const uint32_t px = pxy & 0xffffu, py = (pxy >> 16) * A.shard_count + A.shard_index;
            float u = 0.f, v = 0.f;
#if PT_REJ_CAP
            // A lane that has not found its point after PT_REJ_CAP tries of the shared loop PUTS OFF the rest to the next trip (the loop ran ~4 trips per
            // wave-trip for its last one or two lanes -- mostly Metal's ball in the cube, 48 % rejected): it keeps its role, sits this trip out, and a
            // camera lane keeps its two jitter draws in its dead origin registers behind a marker no direction can be. The lane's draws keep their order.
            constexpr uint32_t kRejMarker = 0x7fc0dea5u;
            const bool cam_resumed = cam_role && __float_as_uint(d.z) == kRejMarker;
            if (cam_role && cam_resumed) u = o.x, v = o.y;
            if (cam_role && !cam_resumed) {
#else
            if (cam_role) {   // scene.rs:107-108: the jitter draws come before the lens draws
#endif
                u = rng_plus(rng, (float)px) * pn2.z;
                v = rng_plus(rng, (float)py) * pn2.w;
            }
            float sx, sy, sz;
#if PT_REJ_CAP
            bool inside_ball = false;
            for (uint32_t tries = 0; tries < (uint32_t)(PT_REJ_CAP); ++tries) {
                sx = rng_pm1(rng), sy = rng_pm1(rng);   // math.rs:8 / math.rs:17-21: 2 * draw - 1
                sz = 0.0f;
                if (met_role) sz = rng_pm1(rng);
                inside_ball = ((sx * sx + sy * sy) + sz * sz) < 1.0f;
                if (inside_ball) break;
            }
            if (!inside_ball) {
                if (cam_role) o.x = u, o.y = v, d.z = __uint_as_float(kRejMarker);
            } else {
#else
            for (;;) {   // (a plain divergent loop: a lane leaves when its point is inside, the wave when its last lane has)
                sx = rng_pm1(rng), sy = rng_pm1(rng);   // math.rs:8 / math.rs:17-21: 2 * draw - 1
                sz = 0.0f;
                if (met_role) sz = rng_pm1(rng);
                if (((sx * sx + sy * sy) + sz * sz) < 1.0f) break;
            }
            {
#endif
            f3 vec;
            if (cam_role) {
                const float rdx = cam_lens_radius * sx, rdy = cam_lens_radius * sy;
                const f3 offset = add3(scale3(cam_u, rdx), scale3(cam_v, rdy));
                const float tdraw = rng_f32(rng);  // camera.rs:59 time draw (plain spheres ignore ray.time)
                if (MOVING) rtime = cam_time0 + tdraw * (cam_time1 - cam_time0);
                vec = sub3(sub3(add3(add3(cam_llc, scale3(cam_horizontal, u)), scale3(cam_vertical, v)), cam_origin), offset);
                o = add3(cam_origin, offset);
                sd &= ~0xfffu;   // depth = 0
                need_cam = false;
                trav_new = true;
            } else {
                // material.rs:82: reflected + fuzz * random_in_unit_sphere (the fuzz waited in the lane's idle candidate-queue slot)
                const float fuzz = MFMA ? __uint_as_float(reinterpret_cast<const uint32_t *>(s_queue)[tid]) : metal_fuzz;
                vec = add3(d, scale3(mk3(sx, sy, sz), fuzz));
            }
            d = normalize3(vec);   // camera.rs:66 / material.rs:84, once for both roles
            pend_metal = false;
            }
        }

        PT_SEC(1);
#ifdef PT_CULLSTATS
        {   // development aid: who holds a ray this trip -- debug[96] trips, [97] lanes with a ray, [98] lanes waiting for the batched refill, [99] lanes with no work left
            const unsigned long long hv = wave_ballot(have), wt = wave_ballot(!have && !exhausted), ex = wave_ballot(!have && exhausted);
            if (lane == 0) {
                atomicAdd(&A.debug[96], 1ull);
                atomicAdd(&A.debug[97], (unsigned long long)__popcll(hv));
                atomicAdd(&A.debug[98], (unsigned long long)__popcll(wt));
                atomicAdd(&A.debug[99], (unsigned long long)__popcll(ex));
                atomicAdd(&A.debug[104 + (__popcll(hv) >> 3)], 1ull);   // histogram of live lanes per trip, in eighths of a wave ([112]: all 64)
            }
        }
#endif
        // ---- hitable.rs:39-65: closest hit (inactive lanes carry a null ray)
#if defined(PT_CAMSKIP) || PT_REJ_CAP
        const bool sat_out = have && (need_cam || pend_metal);   // (lanes whose camera ray / Metal sample was put off sit this trip out; restored below)
        have = have && !sat_out;
#endif
        const f3 ro = have ? o : mk3(0.f, 0.f, 0.f);
        const f3 rd = have ? d : mk3(0.f, 0.f, 0.f);
        const float a = dot3(rd, rd);  // sphere.rs:34
        // its reciprocal, once per ray, for the quotients of the exact sphere tests (pt_device.h DivA; the exact-scan and binary-tree kernels divide in full)
        const bool short_div = (MFMA && !BVH) || TREE4;
        const DivA av{a, short_div ? recip_unit_range(a) : 0.0f, short_div && wave_ballot(have && !in_unit_range(a)) == 0ull};
        float t_hit;
        int idx;
        if (TREE4) {
            // (every ray of the wave is finished when this returns: no traversal state is carried into the next trip)
            if (GRID) grid_trace<MOVING, VERIFY, BLK>(A, reinterpret_cast<uint16_t *>(s_bvh), reinterpret_cast<uint32_t *>(s_queue), w_pairs, w_keys, w_park, grid_parked, ro, rd, av, rtime, have, steal4
#ifdef PT_SECTIONS
                                                         , sec_t
#endif
                                                         );
            else bvh4_trace<MOVING, VERIFY, BLK>(A, reinterpret_cast<uint16_t *>(s_bvh), reinterpret_cast<uint32_t *>(s_queue), w_pairs, w_keys, ro, rd, av, rtime, have, steal4
#ifdef PT_SECTIONS
                                            , sec_t
#endif
                                            );
            trav_new = false;
            idx = -1, t_hit = kMaxT;
            if (have && !(GRID && ((grid_parked >> lane) & 1ull))) {   // (lanes without a ray hold a stale or never-written key; a parked walk has no result yet)
                const unsigned long long key = w_keys[lane];
                const uint32_t low = (uint32_t)key;
                // BVH world: the key carries the leaf's DFS rank; shading reads the rank-ordered copy of the records, so
                // the sphere index itself (one more dependent load) is only needed for a moving sphere's motion record
                if (key != ~0ull) idx = (int)(A.gate ? (MOVING ? A.rank_sphere[0xffffffffu - low] : (0xffffffffu - low)) : low);
                t_hit = __uint_as_float((uint32_t)(key >> 32));
            }
        } else if (BVH) {
            if (have && trav_new) {
                bvh_start<MOVING>(A, s_bvh, o, d, dot3(d, d), rtime, trav);
                trav_new = false;
            }
            if (A.nodes_in_lds)
                bvh_run<true, MOVING, VERIFY>(A, s_bvh, s_nodes, ro, rd, a, rtime, have, trav);
            else
                bvh_run<false, MOVING, VERIFY>(A, s_bvh, A.wnodes, ro, rd, a, rtime, have, trav);
            idx = trav.idx;
            t_hit = trav.best;
        } else if (MFMA)
            idx = intersect_list_mfma<VERIFY, MOVING, GATE, BLK>(A, (PAL && GATE) ? GateSrc{s_gate, s_rank} : GateSrc{A.gate, A.leaf_rank}, mot, s_par, SPH_LDS ? (const float4 *)s_sph : A.spheres_r2, s_afrag, s_tile_sphere, s_cull,
                                                      s_queue, w_pairs, w_keys, ro, rd, av, have, rtime, t_hit
#ifdef PT_SECTIONS
                                                      , sec_t
#elif defined(PT_WAVEDBG)
                                                      , dbgc
#endif
                                                      );
        else
            idx = intersect_list(SPH_LDS ? (const float4 *)s_sph : A.spheres_r2, (int)A.n_spheres_pad, s_queue + tid, ro, rd,
                                 a, t_hit);

        PT_SEC(2);
        // ---- scene.rs:49-71 one level of ray_trace (BVH mode: only lanes whose traversal has finished)
        const bool shading = have && !(BVH && !TREE4 && trav.active) && !(GRID && ((grid_parked >> lane) & 1ull));   // (GRID: a lane whose walk was parked finishes its ray in the next trip)
        wave_rays += (unsigned long long)__popcll(wave_ballot(shading));   // scene.rs:57 `ray_count += 1` for every lane shaded below
        // 4-wide tree kernels: Texture::Noise of the lanes that will scatter off a noise-textured Lambertian, evaluated for the
        // whole wave at once (wave_balanced_turb): the shading below is divergent, and a third of its lanes (sky misses) idle
        float turb_pre = 0.0f;
        if (WST && A.has_noise) {
            bool need = false;
            f3 np = mk3(0.f, 0.f, 0.f);
            if (shading && idx >= 0) {
                const float4 q1n = shade[4 * idx + 1];
                need = __float_as_uint(q1n.x) == (uint32_t)PT_MAT_LAMBERTIAN && (__float_as_uint(q1n.y) & kShadeNoise) != 0u &&
                       PT_DEPTH < __float_as_uint(s_par[12].z);
                np = add3(o, scale3(d, t_hit));   // ray.rs:24-26, the same point the shading computes
            }
            // (the eight gradients of an octave are fetched before its arithmetic here: outside the divergent shading block the
            //  32 registers that takes are free -- 128 VGPRs, no spill; +0.8 %)
            turb_pre = wave_balanced_turb(PerlinLds{s_pvec, s_perm, true}, w_pairs, need, np);
        }
        if (shading) {
            pix_rays += 1;
            bool terminal = true;
            f3 V;
            if (idx < 0) {
                // scene.rs:40-47
                const float4 psky = s_par[11];
                if (psky.w != 0.0f) {
                    V = mk3(psky.x, psky.y, psky.z);
                } else {
                    const float t = 0.5f * (d.y + 1.0f);
                    const float w1 = 1.0f - t;
                    V = mk3(w1 + (t * 0.5f) * 0.3f, w1 + (t * 0.7f) * 0.3f, w1 + (t * 1.0f) * 0.3f);
                }
            } else {
                const float4 sp = sphere_at_m<MOVING>(mot, idx, shade[4 * idx], rtime), q1 = shade[4 * idx + 1],
                             qa = shade[4 * idx + 2], qb = shade[4 * idx + 3];
                const f3 centre = mk3(sp.x, sp.y, sp.z);
                // ray.rs:24-26. The hit point IS the next ray's origin when the path scatters, and a path that ends here gets a new
                // origin from the camera: the lane's origin is advanced in place (no second copy of the point kept alive)
                o = add3(o, scale3(d, t_hit));
                const f3 point = o;
                const f3 normal = divs3_known(sub3(point, centre), sp.w, qa.w);    // sphere.rs:42 (qa.w: 1 / radius from the host)
                struct { uint32_t kind, flags; int32_t tex; float param; } m = {
                    __float_as_uint(q1.x), __float_as_uint(q1.y), (int32_t)__float_as_uint(q1.z), q1.w};
                // Texture::value for this sphere's texture (texture.rs:74-91), inlined for the resolved cases
                auto surface_colour = [&]() -> f3 {
                    if (m.flags & kShadeConst) return mk3(qa.x, qa.y, qa.z);
                    if (m.flags & kShadeChecker2) {
                        const bool odd = checker_is_odd(10.0f * point.x, 10.0f * point.y, 10.0f * point.z);
                        return odd ? mk3(qa.x, qa.y, qa.z) : mk3(qb.x, qb.y, qb.z);
                    }
                    if (m.flags & kShadeNoise) {   // texture.rs:86-89, resolved here: no dependent fetch of the texture record
                        const float v1 = 1.0f + sin_colour(qa.x * point.z + 10.0f * perlin_turb(pn, point));
                        return mk3(0.5f * v1, 0.5f * v1, 0.5f * v1);
                    }
                    return texture_value(A.texs, pn, m.tex, point);
                };
                f3 emitted = mk3(0.f, 0.f, 0.f);                        // material.rs:161-167
                if (m.kind == PT_MAT_DIFFUSE_LIGHT) emitted = surface_colour();
                bool scattered = false;
                f3 att = mk3(1.f, 1.f, 1.f);
                uint32_t attc = WST ? kWstWhite : kWhite;    // PAL / WST: code of `att` (white unless a branch says otherwise)
                if (PT_DEPTH < __float_as_uint(s_par[12].z)) {   // max_depth
                    // every scatter ends in `.normalize()` of some vector (material.rs:63,84,112,119): the branches
                    // only produce that vector, the normalisation is issued once for the whole wave
                    f3 raw = d;
                    if (m.kind == PT_MAT_LAMBERTIAN) {  // material.rs:52-67
                        const f3 target = add3(add3(point, normal), random_unit_vector(rng));
                        if (PAL) {
                            const bool even = (m.flags & kShadeChecker2) && !checker_is_odd(10.0f * point.x, 10.0f * point.y, 10.0f * point.z);
                            attc = (uint32_t)idx | (even ? 0x8000u : 0u);
                        } else if (WST) {
                            if (m.flags & kShadeNoise) {   // texture.rs:86-89 with the turbulence evaluated above
                                const float v1 = 1.0f + sin_colour(qa.x * point.z + 10.0f * turb_pre);
                                attc = __float_as_uint(0.5f * v1);
                            } else {
                                const bool even = (m.flags & kShadeChecker2) && !checker_is_odd(10.0f * point.x, 10.0f * point.y, 10.0f * point.z);
                                attc = kWstCode | (uint32_t)idx | (even ? (1u << 20) : 0u);
                            }
                        } else {
                            att = surface_colour();
                        }
                        raw = sub3(target, point);
                        scattered = true;
                    } else if (m.kind == PT_MAT_METAL) {  // material.rs:69-89
                        const f3 reflected = reflect3(d, normal);
                        if (dot3(reflected, normal) > 0.0f) {
                            att = mk3(qa.x, qa.y, qa.z);
                            attc = (WST ? kWstCode : 0u) | (uint32_t)idx;
                            raw = reflected, pend_metal = true;   // sampled at the top of the next iteration
                            if (MFMA) reinterpret_cast<uint32_t *>(s_queue)[tid] = __float_as_uint(m.param);
                            else metal_fuzz = m.param;
                            scattered = true;
                        }
                    } else if (m.kind == PT_MAT_DIELECTRIC) {  // material.rs:91-124
                        const float ref_idx = m.param;
                        const float rdotn = dot3(d, normal);
                        f3 outward_normal;
                        float ni_over_nt, cosine;
                        if (rdotn > 0.0f) {
                            cosine = rdotn / length3(d);
                            cosine = sqrt_exact(1.0f - ref_idx * ref_idx * (1.0f - cosine * cosine));
                            outward_normal = neg3(normal);
                            ni_over_nt = ref_idx;
                        } else {
                            cosine = -rdotn / length3(d);
                            outward_normal = normal;
                            ni_over_nt = qb.y;   // 1.0 / ref_idx (f32, from the host: pt_prep.hip)
                        }
                        f3 refracted;
                        bool use_refract = false;
                        if (refract3(d, outward_normal, ni_over_nt, refracted)) {
                            const float reflect_prob = qb.x + (1.0f - qb.x) * pow5_ref(1.0f - cosine);   // math.rs:76-80, r0 from the host
                            if (rng_f32(rng) > reflect_prob) use_refract = true;
                        }
                        raw = use_refract ? refracted : reflect3(d, normal);
                        scattered = true;
                    }
                    // (the direction is replaced in place as well: a path that does not scatter ends, and its lane's next ray is a camera ray)
                    if (scattered) d = pend_metal ? raw : normalize3(raw);
                }
                if (scattered) {
                    // the first scatter's attenuation stays in registers (measured best: 0 levels -1.5 %, 2 levels -2.3 %;
                    // it also removes 40 % of the stack's HBM writes); deeper levels go to the per-lane stack, level d
                    // at slot d - 1
                    if (PAL) {
                        if (PT_DEPTH == 0u) att0c = attc;
                        else (s_pal + tid)[(PT_DEPTH - 1u) * BLK] = (uint16_t)attc;
                    } else if (WST) {
                        if (PT_DEPTH == 0u) att0c = attc;
                        else path_st(PT_DEPTH - 1u, __uint_as_float(attc));
                    } else if (PT_DEPTH == 0u) {
                        att0 = att;
                    } else {
                        path_st((PT_DEPTH - 1u) * 3u + 0u, att.x);
                        path_st((PT_DEPTH - 1u) * 3u + 1u, att.y);
                        path_st((PT_DEPTH - 1u) * 3u + 2u, att.z);
                    }
                    sd += 1u;   // depth += 1
                    terminal = false;
                    trav_new = true;
                } else {
                    V = emitted;
                }
            }
            if (terminal) {
                // scene.rs:62-64 unwound: emitted(=0) + attenuation * deeper, innermost first
                if (PAL) {
                    // three levels per trip: the codes, then the colours, are fetched together (two LDS round trips per
                    // trip instead of two per level); the products keep the innermost-first order
                    // (the lane's column is re-derived here, behind an opaque copy: hoisted out of the main loop this address is one
                    //  register too many for the 128 the kernel may use, and it was the last value the compiler spilled)
                    uint32_t tid_here = (uint32_t)tid;
                    asm volatile("" : "+v"(tid_here));
                    const uint16_t *my_pal = s_pal + tid_here;
                    for (int k = (int)PT_DEPTH - 1; k >= 1; k -= 3) {
                        const uint32_t ca = my_pal[(uint32_t)(k - 1) * BLK];
                        const uint32_t cb = my_pal[(uint32_t)(k >= 2 ? k - 2 : 0) * BLK];
                        const uint32_t cc = my_pal[(uint32_t)(k >= 3 ? k - 3 : 0) * BLK];
                        const f3 qa3 = palette_colour(ca), qb3 = palette_colour(cb), qc3 = palette_colour(cc);
                        V = mk3(0.0f + qa3.x * V.x, 0.0f + qa3.y * V.y, 0.0f + qa3.z * V.z);
                        if (k >= 2) V = mk3(0.0f + qb3.x * V.x, 0.0f + qb3.y * V.y, 0.0f + qb3.z * V.z);
                        if (k >= 3) V = mk3(0.0f + qc3.x * V.x, 0.0f + qc3.y * V.y, 0.0f + qc3.z * V.z);
                    }
                }
                if (WST) {
                    // three levels per trip, as above: the words, then the colours behind them, are fetched together (a level per trip ran
                    // 8 trips per wave-iteration on config 5, four lanes switched on, each trip one dependent LDS round trip)
                    for (int k = (int)PT_DEPTH - 1; k >= 1; k -= 3) {
                        const uint32_t wa = __float_as_uint(path_ld((uint32_t)(k - 1)));
                        const uint32_t wb = __float_as_uint(path_ld((uint32_t)(k >= 2 ? k - 2 : 0)));
                        const uint32_t wc = __float_as_uint(path_ld((uint32_t)(k >= 3 ? k - 3 : 0)));
                        const f3 ca = word_colour(wa), cb = word_colour(wb), cc = word_colour(wc);
                        V = mk3(0.0f + ca.x * V.x, 0.0f + ca.y * V.y, 0.0f + ca.z * V.z);
                        if (k >= 2) V = mk3(0.0f + cb.x * V.x, 0.0f + cb.y * V.y, 0.0f + cb.z * V.z);
                        if (k >= 3) V = mk3(0.0f + cc.x * V.x, 0.0f + cc.y * V.y, 0.0f + cc.z * V.z);
                    }
                }
                for (int k = (PAL || WST) ? 0 : (int)PT_DEPTH - 1; k >= 1; --k) {
                    if (PAL || WST) {
                    } else {
                        V.x = 0.0f + path_ld((uint32_t)(k - 1) * 3u + 0u) * V.x;
                        V.y = 0.0f + path_ld((uint32_t)(k - 1) * 3u + 1u) * V.y;
                        V.z = 0.0f + path_ld((uint32_t)(k - 1) * 3u + 2u) * V.z;
                    }
                }
                if (PT_DEPTH > 0u) {
                    const f3 c0 = PAL ? palette_colour(att0c) : (WST ? word_colour(att0c) : att0);
                    V = mk3(0.0f + c0.x * V.x, 0.0f + c0.y * V.y, 0.0f + c0.z * V.z);
                }
                col = add3(col, V);  // scene.rs:110
                sd += 0x1000u;   // sample += 1
                need_cam = true;
                if ((sd >> 12) == __float_as_uint(s_par[12].w)) {   // samples
                    // the pixel is written when the lane fetches its next one (the refill below is batched over
                    // several lanes, and so is this read-modify-write of the frame buffer)
                    have = false;
                    finished = true;
                }
            }
        }
        PT_SEC(3);
#if defined(PT_CAMSKIP) || PT_REJ_CAP
        have = have || sat_out;
#endif
        if (TAIL && tail_polled) {
            // Hand-over (pt_coop.h): the list is dry and the probed wave is an idle worker -- the lane at a sample boundary with the
            // most estimated work left parks its pixel (RNG stream, colour sum, counters: the pixel's whole state between two samples)
            // in that worker's mailbox, to be finished with all 64 lanes on each of its rays. At most one pixel per wave and iteration.
            const bool idle = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tail_probe) == ((A.tail_gen << 2) | kBoxIdle) &&
                              (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(tail_probe >> 32)) == (A.tail_gen >> 30);
            tail_streak = idle ? tail_streak + 1u : 0u;
            const uint32_t live = (uint32_t)__popcll(wave_ballot(have));
            if (idle && (live <= A.tail_live_max || tail_streak >= A.tail_streak)) {
#if PT_REJ_CAP
                const bool cand = have && need_cam && __float_as_uint(d.z) != 0x7fc0dea5u;   // between two samples (a lane that has put off its lens point is INSIDE one: its stream is past the jitter draws)
#else
                const bool cand = have && need_cam;   // between two samples
#endif
                const uint32_t done_s = sd >> 12;
                const float est = (float)pix_rays * (float)(__float_as_uint(s_par[12].w) - done_s) * __builtin_amdgcn_rcpf((float)done_s);   // rays per sample so far x samples left
                const uint32_t eb = (cand && est >= A.tail_min_est) ? __float_as_uint(est) : 0u;   // (NaN before the first sample: not >=)
                const uint32_t mx = wave_max_u32(eb);
                if (mx != 0u && eb == mx && lane == __builtin_ctzll(wave_ballot(eb == mx))) {
                    if (coop_hand_over(A, tail_target, rng, col, pxy, done_s, pix_rays)) have = false;   // (not `finished`: nothing is written, the lane simply holds no pixel any more)
                }
            }
        }
        // The loop's only exit, at its very end. (A wave whose refill brought no pixel -- beyond the frame's edge, or the list ran dry --
        // used to skip the body with `continue` / leave with `break` from here up there. The compiler's structurizer turns such an edge
        // into a flag tested after the body, which keeps every loop-carried register's start-of-iteration value alive THROUGH the body:
        // a second home for ~28 registers and ~45 copies per wave-iteration. An idle trip through the body is harmless: no lane has a ray.)
        if (wave_ballot(have) == 0ull && wave_ballot(!exhausted) == 0ull) break;
    }

    if (TAIL && A.tail_cap != 0u) {
        // this wave hands nothing over any more (its mailbox stores were drained before their flags): count it, and become a worker --
        // unless it is the last one out of a main loop: then nobody can hand anything over, and every worker is told so
        uint32_t last = 0u;
        if (lane == 0 && PT_COOP_DBG(4u)) {
            // (the waves of a workgroup count in LDS -- the one spare word of the parameter block, zero since it was staged -- and only
            // the last of them touches the global counters: 4 096 waves counting on one word cost a 2 ms frame 0.2 ms)
            uint32_t *wg_done = reinterpret_cast<uint32_t *>(const_cast<float4 *>(s_par) + 13) + 3;
            if (atomicAdd(wg_done, 1u) + 1u == (uint32_t)(BLK / 64))
                last = (atomicAdd(&A.work_counter[kCtlDone], 1u) + 1u == ctl_load(A.work_counter + kCtlStarted)) ? 1u : 0u;
        }
        if (__builtin_amdgcn_readfirstlane((int)last) != 0) coop_broadcast_exit(A, A.tail_cap);
        else if (PT_COOP_DBG(1u))
            coop_worker<MOVING, GATE>(A, GATE ? GateSrc{s_gate, s_rank} : GateSrc{nullptr, nullptr}, mot, s_par, s_sph, s_shade, pn, blockIdx.x * (BLK / 64) + wave_id, wave_rays);
    }
#ifdef PT_SECTIONS
    PT_SEC(4);
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&A.debug[16 + i], sec_t[i]);
    if (A.wave_end && lane == 0)
        for (int q = 0; q < 8; ++q) A.wave_end[(5 + q) * (gridDim.x * (BLK / 64)) + blockIdx.x * (BLK / 64) + wave_id] = sec_t[q];
#endif
    if (A.wave_end && lane == 0) {
        const uint32_t w = blockIdx.x * (BLK / 64) + wave_id, nw = gridDim.x * (BLK / 64);
        A.wave_end[w] = wall_clock64();
        // (per-lane values of lane 0 would miss other lanes' exhaustion: take the wave's earliest)
#ifdef PT_WAVE_DETAIL
        A.wave_end[nw + w] = dbg_iters | (dbg_exh_iter << 32), A.wave_end[2 * nw + w] = dbg_last_refill, A.wave_end[3 * nw + w] = dbg_start;
        A.wave_end[4 * nw + w] = dbg_first_pxy | ((unsigned long long)pxy << 32);
#else
        (void)nw;
#endif
#ifdef PT_WAVEDBG
        for (int q = 0; q < 4; ++q) A.wave_end[(5 + q) * nw + w] = dbgc[q];
#endif
    }
    if (BVH && VERIFY) {   // traversal counters (accumulate over the lane's whole life: never reset per ray)
        atomicAdd(&A.debug[8], (unsigned long long)(TREE4 ? steal4.visits : trav.visits));
        atomicAdd(&A.debug[9], (unsigned long long)(TREE4 ? steal4.leaves : trav.leaves) + (lane == 0 ? wave_rays * A.n_bvh_large : 0ull));
    }
    // scene.rs:118 ray_count: wave reduce, one atomic per wave
    if (lane == 0) atomicAdd(A.ray_count, wave_rays);
#undef PT_DEPTH
#undef PT_POOL_LEFT
#undef PT_POOL_DRY
#undef PT_POOL_EXACT
#undef PT_POOL_SHARE
}

}  // namespace ptdev
