// pt_launch.hip -- one Scene::update (scene.rs:73-121) enqueued on a stream: validate the frame, ask pt_select.h which kernel
// and geometry, fill the argument block, order the work (heavy tiles first), launch. Buffers that grow with the frame are
// sized by the ensure_* helpers; nothing here decides WHICH code runs.
#include "pt_kernels.h"

#include <mutex>

namespace pthostside {

namespace {

f3 to3(const float *p) { return f3{p[0], p[1], p[2]}; }

// Workgroups of `blk` threads one CU can hold by registers: the unified VGPR file gives min(8, 512 / alloc) waves per SIMD,
// alloc = the kernel's VGPR count rounded up to the granule of 8 (MI355X_MICROARCH.md "Register files"). A persistent grid
// must not exceed it: a workgroup that does not fit only starts when another one retires.
uint32_t blocks_per_cu_by_registers(const void *kern, uint32_t blk) {
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, kern) != hipSuccess || attr.numRegs <= 0) return 8u;
    const uint32_t alloc = ((uint32_t)attr.numRegs + 7u) / 8u * 8u;
    const uint32_t waves_per_simd = std::min<uint32_t>(8u, 512u / alloc);
    return std::max<uint32_t>(1u, waves_per_simd * 4u / (blk / 64u));
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-function, PROCESS-wide attribute: the limit of a kernel is only ever
// raised, and remembered per (device, kernel) for all scenes.
int raise_lds_limit(int device, const void *kern, uint32_t lds) {
    struct Entry { int device; const void *kern; uint32_t lds; };
    static std::mutex mu;
    static std::vector<Entry> seen;
    std::lock_guard<std::mutex> lock(mu);
    for (Entry &e : seen)
        if (e.device == device && e.kern == kern) {
            if (e.lds >= lds) return PT_OK;
            HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            e.lds = lds;
            return PT_OK;
        }
    HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    seen.push_back(Entry{device, kern, lds});
    return PT_OK;
}

// The per-frame arguments both kernels share (KArgs and WArgs use the same member names).
template <typename Args>
void fill_frame_args(Args &X, const pt_scene *s, const pt_params *params, const pt_camera *cam, uint32_t frame_num, uint32_t shard_index, uint32_t shard_count,
                     float *d_rgb, uint64_t *d_ray_count, const ptsel::KernelChoice &c) {
    X.has_sky = s->has_sky;
    X.sky = to3(s->sky);
    X.has_noise = s->tr.has_noise ? 1u : 0u;
    X.cam.origin = to3(cam->origin);
    X.cam.lower_left_corner = to3(cam->lower_left_corner);
    X.cam.horizontal = to3(cam->horizontal);
    X.cam.vertical = to3(cam->vertical);
    X.cam.u = to3(cam->u);
    X.cam.v = to3(cam->v);
    X.cam.w = to3(cam->w);
    X.cam.time0 = cam->time0;
    X.cam.time1 = cam->time1;
    X.cam.lens_radius = cam->lens_radius;
    X.width = params->width;
    X.height = params->height;
    X.samples = params->samples;
    X.max_depth = params->max_depth;
    X.frame_num = frame_num;
    {   // scene.rs:82-87, evaluated in f32 exactly like the reference
        const volatile float one = 1.0f;
        X.inv_nx = one / (float)params->width;
        X.inv_ny = one / (float)params->height;
        X.inv_ns = one / (float)params->samples;
        const volatile float mp = (float)frame_num / (float)(frame_num + 1u);
        X.mix_prev = mp;
        X.mix_new = one - mp;
    }
    X.random_seed = params->random_seed;
    X.refill_min = c.refill_min;
    X.seed_base = s->seed_base;
    X.shard_index = shard_index;
    X.shard_count = shard_count;
    X.local_rows = pt_shard_rows(params->height, shard_index, shard_count);
    X.tiles_x = (params->width + kTileSide - 1u) / kTileSide;
    X.tiles_x_magic = X.tiles_x > 1u ? (uint32_t)(0x100000000ull / X.tiles_x) : 0xffffffffu;   // (tiles_x == 1: umulhi gives tile - 1, corrected by the kernel's one step)
    X.n_items = X.tiles_x * ((X.local_rows + kTileSide - 1u) / kTileSide) * kTilePix;
    X.rgb = d_rgb;
    X.prev_zero = 0;
    X.ray_count = reinterpret_cast<unsigned long long *>(d_ray_count);
    X.work_counter = s->d_work_counter;
}

// The binary internal tree (variant bit 2048 for A/B runs, scenes whose attenuations do not fit the tree kernels' stack words,
// trees the packed format cannot hold) is built on the host the first time a launch needs it.
int ensure_binary_tree(pt_scene *s) {
    if (s->binary_built) return PT_OK;
    pt_scene_desc d{};
    d.n_spheres = (uint32_t)s->h_spheres.size();
    d.spheres = s->h_spheres.data();
    AccelBuild acc = build_accel(&d, s->h_motion.empty() ? nullptr : s->h_motion.data(), s->h_t_lo, s->h_t_hi);
    if (acc.depth + 2 > (uint32_t)kBvhStack) return fail(PT_ERR_UNSUPPORTED, "internal BVH depth %u exceeds the traversal stack", acc.depth);
    if (int rc = upload(&s->d_wnodes, acc.nodes.data(), acc.nodes.size())) return rc;
    s->bin_root = acc.root;
    s->tr.bin_depth = acc.depth;
    s->tr.bin_nodes = (uint32_t)acc.nodes.size();
    s->binary_built = true;
    return PT_OK;
}

int ensure_tile_buf(pt_scene *s, uint32_t n_work_tiles) {
    if (n_work_tiles <= s->d_tile_cap) return PT_OK;
    (void)hipFree(s->d_tile_buf);
    s->d_tile_buf = nullptr, s->d_tile_cap = 0, s->hint_valid = false;
    HIP_TRY(hipMalloc((void **)&s->d_tile_buf, (8 + 3 * (size_t)n_work_tiles) * sizeof(uint32_t)));
    s->d_tile_cap = n_work_tiles;
    return PT_OK;
}

int ensure_px_state(pt_scene *s, size_t pixels) {
    if (pixels <= s->d_px_state_pixels) return PT_OK;
    (void)hipFree(s->d_px_state);
    s->d_px_state = nullptr, s->d_px_state_pixels = 0;
    HIP_TRY(hipMalloc((void **)&s->d_px_state, pixels * 48u));
    s->d_px_state_pixels = pixels;
    return PT_OK;
}

// cooperative mode (pt_coop.h): one mailbox per wave of the grid; its words are stamped with the launch's generation, so they are
// cleared only here
int ensure_tail(pt_scene *s, uint32_t waves, hipStream_t stream) {
    if (waves <= s->tail_cap) return PT_OK;
    (void)hipFree(s->d_tail_box);
    s->d_tail_box = nullptr, s->tail_cap = 0;
    HIP_TRY(hipMalloc((void **)&s->d_tail_box, (size_t)waves * 128u));
    HIP_TRY(hipMemsetAsync(s->d_tail_box, 0, (size_t)waves * 128u, stream));   // (on the launch's stream: a NULL-stream memset is not ordered against a non-blocking one)
    s->tail_cap = waves;
    return PT_OK;
}

int ensure_gstack(pt_scene *s, size_t need_floats) {
    if (need_floats <= s->d_gstack_floats) return PT_OK;
    (void)hipFree(s->d_gstack);
    s->d_gstack = nullptr, s->d_gstack_floats = 0;
    HIP_TRY(hipMalloc((void **)&s->d_gstack, need_floats * sizeof(float)));
    s->d_gstack_floats = need_floats;
    return PT_OK;
}

// Heavy-first work order, shared by both kernel families. A repeated view is ordered by the rays
// each tile took in its last frame (measured by the frame kernel itself); a new view runs as TWO launches: the first (the
// family's measuring twin, phase 1) traces the first sample of every pixel in natural order, counts the rays per tile and parks
// each pixel's RNG stream and colour sum, the second continues, ordered by those costs. Only WHEN a pixel is rendered depends
// on any of this, never its value. `A` is the frame kernel's argument block (updated in place), `measure` launches phase 1.
inline void set_checker(KArgs &A, bool on) { A.checker = on ? 1u : 0u; }
inline void set_checker(WArgs &, bool) {}   // (the general-world kernel measures every tile)
// The caller has NOT zeroed the work counter block and the ray count: a new view's reset kernel does it together with the tile costs and the
// list of measured tiles (one launch where there were four fills and a kernel), a repeated view's two fills here; the order kernel zeroes the
// work counter again for the frame kernel.
template <typename Args, typename LaunchMeasure>
int order_work(pt_scene *s, Args &A, const pt_params *params, const pt_camera *cam, uint32_t shard_index, uint32_t shard_count, hipStream_t stream,
               uint32_t measure_refill, LaunchMeasure measure, bool checker = false) {
    const uint32_t n_work_tiles = A.n_items / kTilePix;
    if (int rc = ensure_tile_buf(s, n_work_tiles)) return rc;
    uint32_t *cost = s->d_tile_buf + 8, *order = cost + s->d_tile_cap, *measured = order + s->d_tile_cap;
    pt_scene::ViewKey key{};
    key.params = *params, key.cam = *cam, key.shard_index = shard_index, key.shard_count = shard_count, key.variant = s->variant, key.n_tiles = n_work_tiles;
    const bool reuse = s->hint_valid && (s->variant & ptsel::kVarMeasureEveryFrame) == 0 && memcmp(&key, &s->hint_key, sizeof key) == 0;
    uint32_t measured_scale = params->samples * (params->max_depth + 1u);
    if (reuse) {
        // the last frame of this view measured every tile: order by that (64 buckets over samples x (depth + 1) x 64 pixels)
        HIP_TRY(hipMemsetAsync(A.ray_count, 0, sizeof(uint64_t), stream));
        launch_tile_order(n_work_tiles, measured, s->hint_scale, order, 0u, 0u, s->d_work_counter, stream);
    } else {
        if (int rc = ensure_px_state(s, (size_t)A.width * A.local_rows)) return rc;
        Args A1 = A;
        A1.samples = 1, A1.phase = 1, A1.px_state = s->d_px_state, A1.tile_cost = cost;
        const uint32_t tiles_y = n_work_tiles / A.tiles_x;
        // (checker: every other tile -- pt_kernels_list.hip; the list of the measured colour lives where the frame's order will be written)
        launch_frame_reset(s->d_work_counter, A.ray_count, cost, n_work_tiles, checker ? order : nullptr, A.tiles_x, tiles_y, stream);
        if (checker) {
            A1.tile_order = order;
            A1.n_items = ((tiles_y >> 1) * A.tiles_x + ((tiles_y & 1u) ? (A.tiles_x + 1u) / 2u : 0u)) * kTilePix;
        }
        // one sample per pixel: refills dominate, batch them hard (`measure_refill` lanes must be waiting: the caller's choice)
        A1.refill_min = dev_knobs().phase1_refill > 0 ? (uint32_t)dev_knobs().phase1_refill : measure_refill;
        measure(A1);
        launch_tile_order(n_work_tiles, cost, params->max_depth + 1u, order, checker ? A.tiles_x : 0u, tiles_y, s->d_work_counter, stream);
        A.samples = params->samples - 1u, A.phase = 2, A.px_state = s->d_px_state;
        set_checker(A, checker);
        measured_scale = A.samples * (params->max_depth + 1u);
    }
    HIP_TRY(hipGetLastError());
    A.tile_order = order;
    s->tile_rays_info.n_tiles = n_work_tiles, s->tile_rays_info.tiles_x = A.tiles_x;
    s->tile_rays_info.first_sample_in_cost = !reuse, s->tile_rays_info.checker = !reuse && checker;
    s->tile_rays_info.frame_counted = (s->variant & ptsel::kVarMeasureEveryFrame) == 0;
    if ((s->variant & ptsel::kVarMeasureEveryFrame) == 0) {   // this frame measures the tiles for the next one (after the order kernel has read the old values)
        HIP_TRY(hipMemsetAsync(measured, 0, (size_t)n_work_tiles * sizeof(uint32_t), stream));
        A.tile_cost = measured;
        s->hint_key = key, s->hint_valid = true, s->hint_scale = measured_scale;
    }
    return PT_OK;
}

int launch_world(pt_scene *s, const ptsel::KernelChoice &c, const pt_params *params, const pt_camera *cam, uint32_t frame_num, uint32_t shard_index,
                 uint32_t shard_count, float *d_rgb, uint64_t *d_ray_count, hipStream_t stream, const BeforeFrame *before_frame) {
    WArgs W;
    memset(&W, 0, sizeof W);
    W.hit = s->d_hitables;
    W.xf = s->d_transforms;
    W.nodes = s->d_ref_nodes;
    W.mats = s->d_mats;
    W.texs = s->d_texs;
    W.perlin_vec = s->d_perlin_vec;
    W.perlin_perm = s->d_perlin_perm;
    W.image_table = s->d_image_table;
    W.image_bytes = s->d_image_bytes;
    W.has_image = s->tr.has_image ? 1u : 0u;
    W.atts_finite = s->tr.atts_finite ? 1u : 0u;
    W.n_hit = s->tr.n_hitables;
    W.n_xf = s->tr.n_world_xf;
    W.bvh_root = c.ref_bvh ? s->bvh_root : -1;
    W.bvh_stack_entries = c.bvh_stack_entries;
    W.stack_in_lds = c.stack_in_lds;
    fill_frame_args(W, s, params, cam, frame_num, shard_index, shard_count, d_rgb, d_ray_count, c);
    if (c.order != ptsel::Order::Measured || W.n_items == 0) {   // (an ordered frame: order_work zeroes both, with what else the view needs)
        HIP_TRY(hipMemsetAsync(s->d_work_counter, 0, 64, stream));
        HIP_TRY(hipMemsetAsync(d_ray_count, 0, sizeof(uint64_t), stream));
    }
    if (W.n_items == 0) {
        s->ev_valid = false;
        return PT_OK;
    }
    const WorldKernel wk = world_kernel_for(c);
    uint32_t bpc = c.bpc;
    if (!c.bpc_forced) bpc = std::min(bpc, blocks_per_cu_by_registers(reinterpret_cast<const void *>(wk), c.block));
    uint32_t grid = (uint32_t)s->num_cus * bpc;
    const uint32_t need = (W.n_items + c.block - 1) / c.block;
    if (grid > need) grid = need;
    if (c.gstack) {
        if (int rc = ensure_gstack(s, (size_t)grid * params->max_depth * (c.world_lazy ? 4ull : 3ull) * c.block)) return rc;
        W.gstack = s->d_gstack;
    }
    if (c.world_graph) {   // the interpreted walk's frames: kGraphDepth x kGraphFrame words per lane (pt_graph.h)
        const size_t need_floats = (size_t)grid * ptdev::kGraphDepth * ptdev::kGraphFrame * c.block;
        if (need_floats > s->d_gframes_floats) {
            (void)hipFree(s->d_gframes);
            s->d_gframes = nullptr, s->d_gframes_floats = 0;
            HIP_TRY(hipMalloc((void **)&s->d_gframes, need_floats * sizeof(float)));
            s->d_gframes_floats = need_floats;
        }
        W.gnodes = s->d_gnodes, W.gchildren = s->d_gchildren, W.groot = s->groot, W.gframes = s->d_gframes;
    }
    if (int rc = raise_lds_limit(s->device, reinterpret_cast<const void *>(wk), c.lds_bytes)) return rc;
    HIP_TRY(hipEventRecord(s->ev_pass, stream));
    if (c.order == ptsel::Order::Measured)
        if (int rc = order_work(s, W, params, cam, shard_index, shard_count, stream, 48u,
                                [&](const WArgs &W1) { hipLaunchKernelGGL(wk, dim3(grid), dim3(c.block), c.lds_bytes, stream, W1); }))
            return rc;
    if (before_frame) {
        bool zero = false;
        if (int rc = (*before_frame)(&zero)) return rc;
        W.prev_zero = zero ? 1u : 0u;
    }
    HIP_TRY(hipEventRecord(s->ev_start, stream));
    hipLaunchKernelGGL(wk, dim3(grid), dim3(c.block), c.lds_bytes, stream, W);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->ev_stop, stream));
    s->ev_valid = true;
    s->last_grid = grid, s->last_block = c.block, s->last_lds = c.lds_bytes;
    return PT_OK;
}

void report_dev_aids(pt_scene *s, const KArgs &A, uint32_t grid, uint32_t blk, hipStream_t stream);

}  // namespace

WorldKernel world_kernel_for(const ptsel::KernelChoice &c) { return world_kernel(c.ref_bvh, c.world_hit_lds, c.world_occ, c.world_media, c.world_chains, c.world_lazy, c.world_graph); }

void sphere_kernels_for(const ptsel::KernelChoice &c, SphereKernel *frame, SphereKernel *measure) {
    switch (c.family) {
    case ptsel::Family::Tree4: tree_kernels(true, c.moving, c.verify, c.grid, frame, measure); break;
    case ptsel::Family::TreeBinary: tree_kernels(false, c.moving, c.verify, false, frame, measure); break;
    case ptsel::Family::Mfma:
        if (c.gate) mfma_gate_kernels(c.moving, c.block, c.verify, c.pool_slots != 0u, frame, measure);
        else mfma_list_kernels(c.moving, c.block, c.verify, c.pool_slots != 0u, frame, measure);
        break;
    case ptsel::Family::ScanLds: scan_kernels(true, frame, measure); break;
    default: scan_kernels(false, frame, measure); break;
    }
}

const char *kernel_name(const ptsel::KernelChoice &c, char *buf, size_t cap) {
    static const char *fam[] = {"world", "tree-binary", "tree4", "mfma", "scan-lds", "scan-hbm"};
    if (c.family == ptsel::Family::World)
        snprintf(buf, cap, "world<bvh=%d,hit_lds=%d,occ=%u,media=%d%s>", (int)c.ref_bvh, (int)c.world_hit_lds, c.world_occ, (int)c.world_media, c.world_graph ? ",graph" : (c.world_chains ? ",chains" : (c.world_lazy ? ",lazy" : "")));
    else
        snprintf(buf, cap, "%s<blk=%u%s%s%s>", c.grid ? "grid" : fam[(uint32_t)c.family], c.block, c.moving ? ",moving" : "", c.gate ? ",gate" : "", c.verify ? ",verify" : (c.pool_slots ? ",pool" : ""));
    return buf;
}

int launch(pt_scene *s, const pt_params *params, const pt_camera *cam, uint32_t frame_num, uint32_t shard_index, uint32_t shard_count, float *d_rgb,
           uint64_t *d_ray_count, hipStream_t stream, const BeforeFrame *before_frame) {
    if (!s || !params || !cam || !d_rgb || !d_ray_count) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    if (params->width == 0 || params->height == 0 || params->samples == 0) return fail(PT_ERR_INVALID_ARG, "width/height/samples must be non-zero");
    if ((uint64_t)params->width * params->height > 0x3fffffffull) return fail(PT_ERR_INVALID_ARG, "frame too large");
    // every kernel packs a lane's pixel into one register; the sphere kernels its (depth, sample) counters as well
    if (params->width > 0xffffu || params->height > 0xffffu) return fail(PT_ERR_UNSUPPORTED, "width and height must be below 65536");
    if (!s->tr.is_world && (params->max_depth > 0xfffu || params->samples > 0xfffffu))
        return fail(PT_ERR_UNSUPPORTED, "sphere kernels take max_depth < 4096, samples < 2^20");
    if (shard_count == 0 || shard_index >= shard_count) return fail(PT_ERR_INVALID_ARG, "bad shard %u/%u", shard_index, shard_count);
    if (params->use_bvh && s->bvh_root < 0) return fail(PT_ERR_UNSUPPORTED, "use_bvh requested but the scene was created without BVH nodes");
    // (an interpreted scene graph's BVHNode rows are part of the graph, not a tree over the world: the same refusal as pt_debug_select's)
    if (params->use_bvh && !s->tr.has_caller_bvh) return fail(PT_ERR_UNSUPPORTED, "use_bvh requested but the description has no BVH nodes");
    HIP_TRY(hipSetDevice(s->device));
    s->tile_rays_info.n_tiles = 0;   // (pt_scene_debug_tile_rays answers for THIS frame only: order_work fills it in when the frame counts rays per tile)

    // ---- which kernel, which geometry (pt_select.h) ----
    ptsel::Knobs knobs;
    knobs.variant = s->variant, knobs.blocks_per_cu = s->blocks_per_cu, knobs.refill = dev_knobs().refill, knobs.pool = dev_knobs().pool, knobs.world_occ3 = dev_knobs().world_occ3, knobs.world_occ4 = dev_knobs().world_occ4;
    const uint32_t local_rows = pt_shard_rows(params->height, shard_index, shard_count);
    const auto tree4_regs = [&] {
        return blocks_per_cu_by_registers(reinterpret_cast<const void *>(tree4_kernel_for_registers(s->tr.has_motion)), (uint32_t)kBlock);
    };
    ptsel::KernelChoice c;
    ptsel::select_kernel(s->tr, *params, cam->time0, cam->time1, local_rows, knobs, s->tr.is_world ? 4u : tree4_regs(), c);
    if (c.needs_binary_tree && !s->binary_built) {   // built on first use; its size enters the LDS carve
        if (int rc = ensure_binary_tree(s)) return rc;
        ptsel::select_kernel(s->tr, *params, cam->time0, cam->time1, local_rows, knobs, tree4_regs(), c);
    }
    s->last_choice = c;
    if (dev_knobs().debug) {
        char name[96];
        fprintf(stderr, "[ptgpu launch] %s lds %u bpc %u stack_in_lds %u order %u\n", kernel_name(c, name, sizeof name), c.lds_bytes, c.bpc, c.stack_in_lds, (uint32_t)c.order);
    }
    if (c.family == ptsel::Family::World) return launch_world(s, c, params, cam, frame_num, shard_index, shard_count, d_rgb, d_ray_count, stream, before_frame);

    // ---- argument block ----
    const bool bvh = c.family == ptsel::Family::Tree4 || c.family == ptsel::Family::TreeBinary, tree4 = c.family == ptsel::Family::Tree4;
    KArgs A;
    memset(&A, 0, sizeof A);
    A.spheres = s->d_spheres;
    A.spheres_r2 = s->d_spheres_r2;
    A.shade = s->d_shade;
    A.sphere_mat = s->d_sphere_mat;
    A.motion = s->d_motion;
    A.mats = s->d_mats;
    A.texs = s->d_texs;
    A.perlin_vec = s->d_perlin_vec;
    A.perlin_perm = s->d_perlin_perm;
    A.gate = c.ref_bvh ? s->d_gate : nullptr;   // list semantics: no ancestor-AABB gate, ties to the lower index
    A.gate_chain = s->d_gate_chain;
    A.bvh_large = s->d_bvh_large;
    A.n_bvh_large = s->n_bvh_large;
    A.nodes4 = s->d_nodes4q;
    if (c.grid) {
        const pthostside_grid_geom &q = s->grid_geom;
        A.grid_cells = s->d_grid_cells, A.grid_rec = s->d_grid_rec, A.grid_large = s->d_grid_large, A.n_grid_large = q.n_large;
        for (int k = 0; k < 3; ++k) A.grid_n[k] = q.n[k], A.grid_min[k] = q.gmin[k], A.grid_centre[k] = q.centre[k];
        for (int k = 0; k < 3; ++k) A.grid_h[k] = q.ha[k], A.grid_inv_h[k] = 1.0f / q.ha[k];
        A.grid_half_diag = q.half_diag, A.grid_d_build = q.d_build;
    }
    A.slotrec = s->d_slotrec;
    A.rank_sphere = s->d_rank_sphere;
    A.shade_rank = s->d_shade_rank;
    A.leaf_rank = s->d_leaf_rank;
    A.n_spheres = s->tr.n_spheres;
    A.n_spheres_pad = ptsel::scan_pad(s->tr.n_spheres);
    fill_frame_args(A, s, params, cam, frame_num, shard_index, shard_count, d_rgb, d_ray_count, c);
    SphereKernel kern = nullptr, measure_kern = nullptr;
    sphere_kernels_for(c, &kern, &measure_kern);
    const bool ordered = c.order == ptsel::Order::Measured && measure_kern != nullptr;
    if (!ordered || A.n_items == 0) {   // (an ordered frame: order_work zeroes both, with what else the view needs)
        HIP_TRY(hipMemsetAsync(s->d_work_counter, 0, 64, stream));
        HIP_TRY(hipMemsetAsync(d_ray_count, 0, sizeof(uint64_t), stream));
    }
    if (A.n_items == 0) {
        s->ev_valid = false;
        return PT_OK;
    }
    A.afrag = s->d_afrag;
    A.tile_sphere = s->d_tile_sphere;
    A.cull_tab = s->d_cull_tab;
    A.cull_axis = c.cull_off ? 3u : s->cull_axis;
    A.cull_always = s->cull_always;
    A.cull_u0 = s->cull_u0, A.cull_inv_cell = s->cull_inv_cell;
    A.cull_axis2 = s->cull_axis2, A.cull_u0_2 = s->cull_u0_2, A.cull_inv_cell_2 = s->cull_inv_cell_2;
    memcpy(A.clip_min, s->clip_min, 12), memcpy(A.clip_max, s->clip_max, 12);
    if (A.cull_axis < 3u) {
        // Per-RAY reach of the reference's f32 discriminant error (pt_prefilter.h lane_tile_mask): the kernel pads the clip box and
        // the segment's extent along the sort axis by sqrt(r_min^2 + kappa (2 |o - c0|^2 + 2 Rs^2 + r_max^2)) - r_min for the
        // ray at hand, so nothing here depends on where the camera is. The constants are rounded up.
        const double kappa = 4.0 * 1.3e-6;
        const double k1 = kappa * (2.0 * (double)s->rs_small * s->rs_small + (double)s->cull_rmax * s->cull_rmax) + (double)s->cull_rmin * s->cull_rmin;
        A.cull_reach[0] = std::nextafter((float)(2.0 * kappa), 3.0e38f);
        A.cull_reach[1] = std::nextafter((float)k1, 3.0e38f);
        A.cull_reach[2] = std::nextafter(s->cull_rmin, 0.0f);
    }
    A.large = s->d_large;
    A.n_tiles = c.n_tiles;
    A.n_large = s->n_large;
    A.large0 = s->large0;
    memcpy(A.c0, s->c0, sizeof A.c0);
    A.rs2 = s->rs2;
    A.m0 = s->m0;
    A.gamma = s->gamma;
    A.verify = (c.verify ? 1u : 0u) | ((s->variant & ptsel::kVarNoStack) ? 2u : 0u);
    A.debug = s->d_debug;
    // (cell-grid kernels: the last six lanes of a wave still walking after two rounds park their walk and finish it in the next call. tools/park_ab.sh,
    //  config 5 kernel time: off 59.6 ms; 3 / 4 / 5 / 6 / 8 / 12 / 16 lanes 55.9 / 55.5 / 55.4 / 55.2 / 55.4 / 55.9 / 56.7; after 1 / 2 / 3 rounds within 0.3 ms)
    A.grid_park_max = (s->variant & ptsel::kVarNoPark) ? 0u : (dev_knobs().park_max >= 0 ? std::min<uint32_t>((uint32_t)dev_knobs().park_max, kGridParkMax) : 6u);
    A.grid_park_after = dev_knobs().park_after >= 0 ? (uint32_t)dev_knobs().park_after : 2u;
    A.ready_min = dev_knobs().ready >= 0 ? (uint32_t)dev_knobs().ready : (uint32_t)((tree4) ? kShareMin : kReadyMin);   // (4-wide tree: lanes without work before subtrees change hands)
    A.drain_at = dev_knobs().drain >= 0 ? std::min<uint32_t>((uint32_t)dev_knobs().drain, (uint32_t)(kLeafQ - 4)) : (uint32_t)(kLeafQ - 4);
    A.wnodes = s->d_wnodes;
    A.n_nodes = s->tr.bin_nodes;
    A.bvh_root = tree4 ? (s->has_tree_items ? 0 : -1) : s->bin_root;   // (-1: every sphere is in bvh_large)
    A.bvh_stack_entries = c.bvh_stack_entries;
    A.nodes_in_lds = c.nodes_in_lds;
    A.stack_in_lds = c.stack_in_lds;
    A.lds_sphere_bytes = c.sph_bytes;
    A.pool_slots = c.pool_slots, A.pool_off = c.pool_off;
    A.tile_order = nullptr;
    A.tile_cost = nullptr;
    (void)bvh;

    const uint32_t blk = c.block, lds = c.lds_bytes;
    // ---- persistent grid: CUs x resident workgroups ----
    uint32_t bpc = c.bpc;
    if (!c.bpc_forced) bpc = std::min(bpc, blocks_per_cu_by_registers(reinterpret_cast<const void *>(kern), blk));
    uint32_t grid = (uint32_t)s->num_cus * bpc;
    // Fewer pixels than lanes (a shard of an 8-GPU frame: 120 000 pixels for 262 144 lanes): the waves that win the race for
    // work should be spread over ALL CUs -- two waves on a SIMD iterate faster than four -- so a one-workgroup-per-CU grid is
    // not cut down to the workgroups the pixels would fill (a wave that finds the queue empty leaves at once).
    const uint32_t need = (A.n_items + blk - 1) / blk;
    const bool wide = blk != (uint32_t)kBlock;
    // (with the cooperative mode on, the waves that find no work of their own finish pixels handed over by the others: the full grid stays)
    if (grid > need && !(wide && (c.coop || need * 4u >= grid) && !dev_knobs().clamp_grid)) grid = need;
    if (grid == 0) grid = 1;
    if (c.coop) {
        if (int rc = ensure_tail(s, grid * (blk / 64u), stream)) return rc;
        s->tail_gen = s->tail_gen >= 0x3fffffffu ? 1u : s->tail_gen + 1u;   // (a mailbox word keeps two bits below it)
        A.tail_box = s->d_tail_box, A.tail_cap = grid * (blk / 64u), A.tail_gen = s->tail_gen;
        A.tail_dry0 = A.n_items <= grid * blk ? 1u : 0u;
#ifdef PT_DEVKNOBS
        A.tail_dbg = getenv("PTGPU_COOP_DBG") ? (uint32_t)atoi(getenv("PTGPU_COOP_DBG")) : 0u;
#endif
        // (round 5, with the workers at the lower wave priority: a look every iteration, hand over from 8 live pixels down or as soon as ONE
        //  probe finds an idle worker -- config 3 +1.2 %, the halves / quarters / eighths of config 4 15.0 -> 14.2 / 12.1 -> 11.5 / 10.65 -> 10.3 ms
        //  against round 4's 4 / 2 / every fourth iteration; tools/coop_sweep.sh)
        A.tail_live_max = dev_knobs().coop_live >= 0 ? (uint32_t)dev_knobs().coop_live : 8u;
        A.tail_streak = dev_knobs().coop_streak >= 0 ? (uint32_t)dev_knobs().coop_streak : 1u;
        A.tail_period_mask = dev_knobs().coop_period >= 0 ? (uint32_t)dev_knobs().coop_period : 0u;
        A.tail_min_est = dev_knobs().coop_est >= 0 ? (float)dev_knobs().coop_est : 24.0f;
    }
    // (pixel pools of the 1024-thread frame kernels: a claim parks at most half of the wave's fair share of what the list still holds; below a
    //  share of 4 items claims take what is wanted and no more -- tools/pool_ab.sh)
    A.pool_tail = dev_knobs().pool_tail >= 0 ? (uint32_t)dev_knobs().pool_tail : 4u;
    A.pool_waves_magic = (uint32_t)(0x100000000ull / std::max<uint64_t>(2u, (uint64_t)grid * (blk / 64u)));
    // (pixel pools of the 1024-thread frame kernels: a claim parks at most half of the wave's fair share of what the list still holds; below a
    //  share of 4 items claims take what is wanted and no more -- tools/pool_ab.sh)
    A.pool_tail = dev_knobs().pool_tail >= 0 ? (uint32_t)dev_knobs().pool_tail : 4u;
    A.pool_waves_magic = (uint32_t)(0x100000000ull / std::max<uint64_t>(2u, (uint64_t)grid * (blk / 64u)));
    if (c.gstack) {
        if (int rc = ensure_gstack(s, (size_t)grid * params->max_depth * 3ull * blk)) return rc;
        A.gstack = s->d_gstack;
    }
    if (measure_kern)
        if (int rc = raise_lds_limit(s->device, reinterpret_cast<const void *>(measure_kern), lds)) return rc;
    if (int rc = raise_lds_limit(s->device, reinterpret_cast<const void *>(kern), lds)) return rc;
    HIP_TRY(hipEventRecord(s->ev_pass, stream));
    A.wave_end = s->d_wave_end;
    if (s->d_wave_end) (void)hipMemsetAsync(s->d_wave_end, 0, 65535 * 8, stream), (void)hipMemsetAsync(s->d_wave_end + 65535, 0xff, 8, stream), (void)hipMemsetAsync(s->d_debug + 91, 0, 8, stream);
    if (ordered) {
        // The measuring launch of the prefilter kernels refills ALL lanes of a wave at once: the wave then always holds one whole
        // 8x8 tile, and the rays of one sample differ little in length (measuring launch + order kernel on config 3: 0.44 ms at 16
        // waiting lanes, 0.35 at 48, 0.31 at 60, 0.28 at 64; tools/sweep_phase1.sh). Tree traversals differ a lot: 48 stays better there.
        const uint32_t measure_refill = c.family == ptsel::Family::Mfma ? 64u : 48u;
        if (int rc = order_work(s, A, params, cam, shard_index, shard_count, stream, measure_refill, [&](KArgs A1) {
                A1.wave_end = nullptr;
                hipLaunchKernelGGL(measure_kern, dim3(grid), dim3(blk), lds, stream, A1);
            }, c.family == ptsel::Family::Mfma && !(s->variant & ptsel::kVarMeasureAllTiles) && A.n_items / kTilePix >= 4u * A.tiles_x))
            return rc;
        // 16-wave workgroups: the waves' first fetches are handed out by age class -- the oldest wave of every SIMD takes the head of
        // the heavy-first list (pt_kernel.h first_static; +1 % on configs 3 / 4)
        if (blk == 1024u && (uint64_t)grid * 1024ull <= A.n_items) A.first_static = grid * 1024u;
    }
    if (before_frame) {
        bool zero = false;
        if (int rc = (*before_frame)(&zero)) return rc;
        A.prev_zero = zero ? 1u : 0u;
    }
    HIP_TRY(hipEventRecord(s->ev_start, stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(blk), lds, stream, A);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->ev_stop, stream));
    s->ev_valid = true;
    s->last_grid = grid, s->last_block = blk, s->last_lds = lds;
    report_dev_aids(s, A, grid, blk, stream);
    return PT_OK;
}

namespace {
// Development aids of -DPT_CULLSTATS / -DPT_SECTIONS / -DPT_WAVEDBG builds and PTGPU_TIMING (none of it in the shipped library).
void report_dev_aids(pt_scene *s, const KArgs &A, uint32_t grid, uint32_t blk, hipStream_t stream) {
    (void)s, (void)A, (void)grid, (void)blk, (void)stream;
#ifdef PT_CULLSTATS
    {   // how many tiles the culling leaves
        (void)hipStreamSynchronize(stream);
        unsigned long long c[48];
        (void)hipMemcpy(c, s->d_debug + 24, sizeof c, hipMemcpyDeviceToHost);
        (void)hipMemset(s->d_debug + 24, 0, sizeof c);
        if (c[0]) {
            fprintf(stderr, "[ptgpu cull] wave-iterations %llu, tiles run per iteration %.2f of %u; lanes asked for %.2f tiles each\n  run histogram:", c[0],
                    (double)c[1] / (double)c[0], A.n_tiles, (double)c[3] / (double)(c[2] ? c[2] : 1));
            for (int i = 0; i < 18; ++i) fprintf(stderr, " %d:%.1f%%", i, 100.0 * (double)c[4 + i] / (double)c[0]);
            fprintf(stderr, "\n  lane histogram:");
            for (int i = 0; i < 18; ++i) fprintf(stderr, " %d:%.1f%%", i, 100.0 * (double)c[24 + i] / (double)(c[2] ? c[2] : 1));
            fprintf(stderr, "\n");
        }
        unsigned long long l[32];
        (void)hipMemcpy(l, s->d_debug + 96, sizeof l, hipMemcpyDeviceToHost);
        (void)hipMemset(s->d_debug + 96, 0, sizeof l);
        if (l[0]) {
            const double n = (double)l[0];
            fprintf(stderr, "[ptgpu lanes] trips %llu: lanes with a ray %.2f, waiting for the batched refill %.2f, out of work %.2f; tile-quarters asked per trip %.2f (of 4 x tiles), tile-halves %.2f\n  live lanes per trip (eighths):",
                    l[0], l[1] / n, l[2] / n, l[3] / n, l[4] / n, l[5] / n);
            for (int i = 0; i < 9; ++i) fprintf(stderr, " %d+:%.1f%%", 8 * i, 100.0 * (double)l[8 + i] / n);
            static const char *pop[8] = {"1", "2", "3", "4", "5-8", "9-16", "17-32", "33-64"};
            fprintf(stderr, "\n  tiles run per trip by the number of lanes that asked for them:");
            for (int i = 0; i < 8; ++i) fprintf(stderr, " %s:%.2f", pop[i], (double)l[17 + i] / n);
            fprintf(stderr, "\n");
        }
    }
#endif
#ifdef PT_GRID_ROUNDS
    if (A.grid_cells && (A.verify & 1u)) {   // record visits per ray, wave-level rounds per call of grid_trace
        (void)hipStreamSynchronize(stream);
        unsigned long long c[64];
        (void)hipMemcpy(c, s->d_debug + 24, sizeof c, hipMemcpyDeviceToHost);
        (void)hipMemset(s->d_debug + 24, 0, sizeof c);
        double nr = 0, nc = 0;
        for (int i = 0; i < 32; ++i) nr += (double)c[i], nc += (double)c[32 + i];
        fprintf(stderr, "[ptgpu grid] visits per ray:");
        for (int i = 0; i < 32; ++i) fprintf(stderr, " %d:%.2f%%", i, 100.0 * (double)c[i] / (nr > 0 ? nr : 1));
        fprintf(stderr, "\n[ptgpu grid] rounds per call:");
        for (int i = 0; i < 32; ++i) fprintf(stderr, " %d:%.2f%%", i, 100.0 * (double)c[32 + i] / (nc > 0 ? nc : 1));
        fprintf(stderr, "\n");
        unsigned long long e[48] = {0};
        (void)hipMemcpy(e + 12, s->d_debug + 92, 36 * sizeof(unsigned long long), hipMemcpyDeviceToHost);   // e[12..15] drains, e[16..31] lanes per round, e[32..47] calls per round
        (void)hipMemset(s->d_debug + 92, 0, 36 * sizeof(unsigned long long));
        e[0] = e[12], e[1] = e[13], e[2] = e[14], e[3] = e[15];
        fprintf(stderr, "[ptgpu grid] calls %.3g; drains with pairs per call %.2f (%.1f %% forced by a full queue), %.1f pairs per drain, %.1f lanes walk on after a drain\n  lanes walking in round r (share of calls that reach it):",
                nc, (double)e[0] / (nc > 0 ? nc : 1), 100.0 * (double)e[1] / (double)(e[0] ? e[0] : 1), (double)e[2] / (double)(e[0] ? e[0] : 1), (double)e[3] / (double)(e[0] ? e[0] : 1));
        for (int r = 0; r < 16; ++r) fprintf(stderr, " %d:%.1f(%.0f%%)", r, (double)e[16 + r] / (double)(e[32 + r] ? e[32 + r] : 1), 100.0 * (double)e[32 + r] / (nc > 0 ? nc : 1));
        fprintf(stderr, "\n");
    }
#endif
#ifdef PT_COOPSEC
    {   // the cooperative workers' cycles (pt_coop.h)
        (void)hipStreamSynchronize(stream);
        unsigned long long c[8];
        (void)hipMemcpy(c, s->d_debug + 80, sizeof c, hipMemcpyDeviceToHost);
        (void)hipMemset(s->d_debug + 80, 0, sizeof c);
        if (c[7]) {
            const double rays = (double)(c[6] ? c[6] : 1);
            fprintf(stderr, "[ptgpu coop] %llu pixels handed over, %llu rays traced by workers; cycles per ray: camera %.0f scan %.0f reduce %.0f shade %.0f fold %.0f\n", c[7], c[6],
                    c[0] / rays, c[1] / rays, c[2] / rays, c[3] / rays, c[4] / rays);
        }
    }
#endif
#ifdef PT_SECTIONS
    {   // where the waves' cycles go
        (void)hipStreamSynchronize(stream);
        unsigned long long sec[8];
        (void)hipMemcpy(sec, s->d_debug + 16, sizeof sec, hipMemcpyDeviceToHost);
        (void)hipMemset(s->d_debug + 16, 0, sizeof sec);
        double tot = 0;
        for (int i = 0; i < 5; ++i) tot += (double)sec[i];
        static const char *names[8] = {"refill", "camera", "intersect", "shade+terminal", "epilogue", "  features", "  tiles", "  phase2"};
        fprintf(stderr, "[ptgpu sections]");
        for (int i = 0; i < 8; ++i) fprintf(stderr, " %s %.1f%%", names[i], 100.0 * (double)sec[i] / tot);
        // (4-wide tree kernels: features = node visits, tiles = candidate drains, and the last slot COUNTS wave-trips)
        fprintf(stderr, "  | raw: intersect %.3g cycles, slot7 %.3g, total %.3g\n", (double)sec[2], (double)sec[7], tot);
    }
#endif
    if (s->d_wave_end) {  // distribution of wave finish times
        (void)hipStreamSynchronize(stream);
        const uint32_t nw = grid * (blk / 64);
        std::vector<unsigned long long> t(nw);
        (void)hipMemcpy(t.data(), s->d_wave_end, std::min<size_t>(nw, 65536) * 8, hipMemcpyDeviceToHost);
        std::sort(t.begin(), t.end());
        const double tick_ns = 10.0;  // wall_clock64: 100 MHz
        fprintf(stderr, "[ptgpu timing] waves %u: finish spread (ms after first finisher) p10 %.3f p50 %.3f p90 %.3f p99 %.3f last %.3f\n", nw,
                (t[nw / 10] - t[0]) * tick_ns * 1e-6, (t[nw / 2] - t[0]) * tick_ns * 1e-6, (t[nw * 9 / 10] - t[0]) * tick_ns * 1e-6,
                (t[nw * 99 / 100] - t[0]) * tick_ns * 1e-6, (t[nw - 1] - t[0]) * tick_ns * 1e-6);
        // the cooperative workers' pixels (pt_coop.h): when each was received and finished, on the launch's own timeline
        unsigned long long n_log = 0, t0 = 0;
        (void)hipMemcpy(&n_log, s->d_debug + 91, 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&t0, s->d_wave_end + 65535, 8, hipMemcpyDeviceToHost);
        n_log = std::min<unsigned long long>(n_log, 130000ull);
        if (n_log) {
            std::vector<unsigned long long> e(4 * n_log);
            (void)hipMemcpy(e.data(), s->d_wave_end + 65536, e.size() * 8, hipMemcpyDeviceToHost);
            std::vector<size_t> idx(n_log);
            for (size_t i = 0; i < n_log; ++i) idx[i] = i;
            auto ms = [&](unsigned long long x) { return (double)(x - t0) * tick_ns * 1e-6; };
            std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return e[4 * a] < e[4 * b]; });
            fprintf(stderr, "[ptgpu coop log] %llu pixels; main loops ended %.3f .. %.3f ms; received at p0 %.3f p10 %.3f p50 %.3f p90 %.3f last %.3f ms\n", n_log, ms(t[0]), ms(t[nw - 1]),
                    ms(e[4 * idx[0]]), ms(e[4 * idx[n_log / 10]]), ms(e[4 * idx[n_log / 2]]), ms(e[4 * idx[n_log * 9 / 10]]), ms(e[4 * idx[n_log - 1]]));
            std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return e[4 * a + 1] > e[4 * b + 1]; });
            for (size_t i = 0; i < std::min<size_t>(n_log, 12); ++i) {
                const unsigned long long *q = &e[4 * idx[i]];
                const double dur = (double)(q[1] - q[0]) * tick_ns * 1e-3;
                fprintf(stderr, "   finished %.3f ms: received %.3f ms at sample %u with %u rays behind it, then %u rays in %.0f us (%.2f us per ray), pixel (%u, %u)\n", ms(q[1]), ms(q[0]),
                        (unsigned)(q[2] >> 32), (unsigned)(q[3] >> 32), (unsigned)q[2], dur, dur / std::max(1u, (unsigned)q[2]), (unsigned)(q[3] & 0xffff), (unsigned)((q[3] >> 16) & 0xffff));
            }
        }
    }
}
}  // namespace

}  // namespace pthostside
