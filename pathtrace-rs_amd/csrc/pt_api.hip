// pt_api.hip -- the small entry points of the C ABI: errors, identification, tuning, timing and debug getters, the device
// self-test probes, and pt_debug_select (kernel selection for a description, no GPU needed).
#include <dlfcn.h>

#include "pt_device.h"
#include "pt_host.h"
#include "pt_world.h"   // logf_ref (the probe of constant_medium.rs:60's ln)

using namespace pthostside;

namespace pthostside {

namespace {
thread_local char g_err[512] = "";
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
const char *last_error_message() { return g_err; }

const DevKnobs &dev_knobs() {
    static const DevKnobs k = [] {
        DevKnobs d;
#ifdef PT_DEVKNOBS
        if (const char *e = getenv("PTGPU_REFILL")) d.refill = atoi(e);
        if (const char *e = getenv("PTGPU_READY")) d.ready = atoi(e);
        if (const char *e = getenv("PTGPU_DRAIN")) d.drain = atoi(e);
        if (const char *e = getenv("PTGPU_PHASE1_REFILL")) d.phase1_refill = std::max(1, atoi(e));
        if (const char *e = getenv("PTGPU_CULL_AXIS")) d.cull_axis = atoi(e);
        if (const char *e = getenv("PTGPU_CULL_STRIPS")) d.cull_strips = atoi(e);
        if (const char *e = getenv("PTGPU_COOP_LIVE")) d.coop_live = atoi(e);
        if (const char *e = getenv("PTGPU_COOP_STREAK")) d.coop_streak = atoi(e);
        if (const char *e = getenv("PTGPU_COOP_PERIOD")) d.coop_period = atoi(e);
        if (const char *e = getenv("PTGPU_COOP_EST")) d.coop_est = atoi(e);
        if (const char *e = getenv("PTGPU_PARK_MAX")) d.park_max = std::max(0, atoi(e));
        if (const char *e = getenv("PTGPU_PARK_AFTER")) d.park_after = std::max(0, atoi(e));
        if (const char *e = getenv("PTGPU_POOL")) d.pool = std::max(0, atoi(e));
        if (const char *e = getenv("PTGPU_POOL_TAIL")) d.pool_tail = std::max(0, atoi(e));
        if (const char *e = getenv("PTGPU_HOST_THREADS")) d.host_threads = std::max(0, atoi(e));
        if (const char *e = getenv("PTGPU_BLOCKS_PER_CU")) d.blocks_per_cu = (uint32_t)atoi(e);
        if (const char *e = getenv("PTGPU_VARIANT")) d.variant = (uint32_t)atoi(e);
        d.world_occ3 = getenv("PTGPU_WORLD_OCC3") != nullptr;
        d.world_occ4 = getenv("PTGPU_WORLD_OCC4") != nullptr;
        d.debug = getenv("PTGPU_DEBUG") != nullptr;
        d.clamp_grid = getenv("PTGPU_CLAMP_GRID") != nullptr;
        d.timing = getenv("PTGPU_TIMING") != nullptr;
#endif
        return d;
    }();
    return k;
}

}  // namespace pthostside

extern "C" const char *pt_last_error(void) { return last_error_message(); }
// "ptgpu <version> gfx950 src <hash>[ defs <DEFS>]": the hash covers csrc/*.h, csrc/*.hip, include/ptgpu.h, the Makefile and its DEFS
// (pathtrace-rs_amd/Makefile SRC_HASH) -- what bench.py and the committed profiles use to tell which build a number belongs to.
#ifndef PT_SOURCE_HASH
#define PT_SOURCE_HASH "unknown"
#endif
#ifndef PT_BUILD_DEFS
#define PT_BUILD_DEFS ""
#endif
extern "C" const char *pt_version(void) { return sizeof(PT_BUILD_DEFS) > 1 ? "ptgpu 0.4 gfx950 src " PT_SOURCE_HASH " defs " PT_BUILD_DEFS : "ptgpu 0.4 gfx950 src " PT_SOURCE_HASH; }

extern "C" int pt_device_count(int *count_out) {
    if (!count_out) return fail(PT_ERR_INVALID_ARG, "count_out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count_out = 0;
        return fail(PT_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count_out = n;
    return PT_OK;
}

extern "C" uint32_t pt_shard_rows(uint32_t height, uint32_t shard_index, uint32_t shard_count) {
    if (shard_count == 0 || shard_index >= shard_count || height <= shard_index) return 0;
    return (height - shard_index + shard_count - 1) / shard_count;
}

extern "C" int pt_scene_set_seed_base(pt_scene *s, uint64_t seed_base) {
    if (!s) return fail(PT_ERR_INVALID_ARG, "scene is NULL");
    s->seed_base = seed_base;
    return PT_OK;
}

extern "C" int pt_scene_set_tuning(pt_scene *s, uint32_t blocks_per_cu, uint32_t variant) {
    if (!s) return fail(PT_ERR_INVALID_ARG, "scene is NULL");
    s->blocks_per_cu = blocks_per_cu;
    s->variant = variant;
    return PT_OK;
}

extern "C" int pt_last_kernel_ms(pt_scene *s, float *ms_out) {
    if (!s || !ms_out) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    if (!s->ev_valid) return fail(PT_ERR_INVALID_ARG, "no render has been launched on this scene");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipEventSynchronize(s->ev_stop));
    HIP_TRY(hipEventElapsedTime(ms_out, s->ev_start, s->ev_stop));
    return PT_OK;
}

extern "C" int pt_last_pass_ms(pt_scene *s, float *ms_out) {
    if (!s || !ms_out) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    if (!s->ev_valid) return fail(PT_ERR_INVALID_ARG, "no render has been launched on this scene");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipEventSynchronize(s->ev_stop));
    HIP_TRY(hipEventElapsedTime(ms_out, s->ev_pass, s->ev_stop));
    return PT_OK;
}

extern "C" int pt_last_host_ms(pt_scene *s, float out4[4]) {
    if (!s || !out4) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    memcpy(out4, s->host_ms, sizeof s->host_ms);
    return PT_OK;
}

extern "C" int pt_scene_build_info(pt_scene *s, float *build_ms_out, uint32_t *n_nodes_out, uint32_t *depth_out, uint32_t *on_device_out) {
    if (!s) return fail(PT_ERR_INVALID_ARG, "scene is NULL");
    if (build_ms_out) *build_ms_out = s->tree_build_ms;
    if (n_nodes_out) *n_nodes_out = s->tr.n_nodes4;
    if (depth_out) *depth_out = s->tr.depth4;
    if (on_device_out) *on_device_out = s->tree_on_device ? 1u : 0u;
    return PT_OK;
}

extern "C" int pt_scene_debug_tree(pt_scene *s, void *nodes_out, size_t capacity_bytes) {
    if (!s || !nodes_out) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    const size_t bytes = (size_t)s->tr.n_nodes4 * sizeof(DNode4);
    if (capacity_bytes < bytes) return fail(PT_ERR_INVALID_ARG, "buffer holds %zu bytes, the tree has %zu", capacity_bytes, bytes);
    HIP_TRY(hipSetDevice(s->device));
    if (bytes) HIP_TRY(hipMemcpy(nodes_out, s->d_nodes4, bytes, hipMemcpyDeviceToHost));
    return PT_OK;
}

extern "C" int pt_scene_debug_tree_packed(pt_scene *s, void *nodes_out, size_t capacity_bytes, uint32_t *usable_out) {
    if (!s || !nodes_out) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    const size_t bytes = s->d_nodes4q ? (size_t)s->tr.n_nodes4 * sizeof(DNode4Q) : 0;
    if (capacity_bytes < bytes) return fail(PT_ERR_INVALID_ARG, "buffer holds %zu bytes, the packed tree has %zu", capacity_bytes, bytes);
    HIP_TRY(hipSetDevice(s->device));
    if (bytes) HIP_TRY(hipMemcpy(nodes_out, s->d_nodes4q, bytes, hipMemcpyDeviceToHost));
    if (usable_out) *usable_out = s->tr.tree4_packed ? 1u : 0u;
    return PT_OK;
}

extern "C" int pt_last_launch_info(pt_scene *s, uint32_t *grid_out, uint32_t *block_out, uint32_t *lds_bytes_out) {
    if (!s) return fail(PT_ERR_INVALID_ARG, "scene is NULL");
    if (grid_out) *grid_out = s->last_grid;
    if (block_out) *block_out = s->last_block;
    if (lds_bytes_out) *lds_bytes_out = s->last_lds;
    return PT_OK;
}

namespace {
thread_local ptsel::KernelChoice g_debug_choice;
thread_local bool g_debug_choice_valid = false;
void fill_choice(const ptsel::KernelChoice &c, pt_kernel_choice *out) {
    memset(out, 0, sizeof *out);
    out->family = (uint32_t)c.family;
    out->block = c.block, out->lds_bytes = c.lds_bytes, out->blocks_per_cu = c.bpc;
    out->moving = c.moving, out->gate = c.gate, out->verify = c.verify, out->ref_bvh = c.ref_bvh;
    out->ordered = c.order == ptsel::Order::Measured;
    out->stack_in_lds = c.stack_in_lds, out->global_stack = c.gstack, out->n_tiles = c.n_tiles;
    out->world_hit_lds = c.world_hit_lds, out->world_occ = c.world_occ, out->world_media = c.world_media;
    out->refill_min = c.refill_min;
    out->coop = c.coop ? 1u : 0u;
    out->world_lazy = c.world_lazy ? 1u : 0u;
    out->world_graph = c.world_graph ? 1u : 0u;
    out->pool_slots = c.pool_slots;
    kernel_name(c, out->name, sizeof out->name);
}
}  // namespace

extern "C" int pt_last_kernel_choice(pt_scene *s, pt_kernel_choice *out) {
    if (!s || !out) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    fill_choice(s->last_choice, out);
    return PT_OK;
}

// Kernel selection for a DESCRIPTION, without a device: the host half of pt_scene_create* (pt_prep.hip; the 4-wide tree's
// size comes from the host restatement of the device build, which tests prove identical) followed by pt_select.h.
extern "C" int pt_debug_select(const pt_scene_desc *sphere_desc, const pt_world_desc *world_given, const pt_params *params, const pt_camera *camera,
                               uint32_t shard_count, uint32_t blocks_per_cu, uint32_t variant, pt_kernel_choice *out) {
    if ((!sphere_desc) == (!world_given) || !params || !camera || !out) return fail(PT_ERR_INVALID_ARG, "exactly one description, params, camera and out are required");
    FlatWorld flat;
    const pt_world_desc *world_desc = world_given;
    if (int rc = flatten_world_graph(world_given, flat, &world_desc)) return rc;
    if (shard_count == 0) shard_count = 1;
    ptsel::SceneTraits tr;
    WorldAsSpheres W;
    const pt_scene_desc *sd = sphere_desc;
    const MotionIn *motion = nullptr;
    if (world_desc) {
        if (int rc = analyze_world(world_desc, W)) return rc;
        if (W.sphere_like) sd = &W.desc, motion = W.all_spheres ? nullptr : W.motion.data();
    }
    bool planned = false;
    double plan_t_lo = 0.0, plan_t_hi = 0.0;
    if (sd && sd->n_spheres) {
        SpherePlan P;
        const int rc = plan_sphere_scene(sd, motion, P);
        if (rc == PT_OK) {
            tr = P.tr, plan_t_lo = P.t_lo, plan_t_hi = P.t_hi;
            if (!P.titems.empty()) {
                const Tree4Host t4 = tree4_build_host(P.titems);
                tr.n_nodes4 = (uint32_t)t4.nodes.size(), tr.depth4 = t4.depth;
                tr.tree4_packed = true;
                for (const DNode4 &w : t4.nodes) {
                    DNode4Q q;
                    if (!tree_pack_node(w, q)) tr.tree4_packed = false;
                }
            }
            if (world_desc) tr.n_hitables = world_desc->n_hitables, tr.n_world_xf = world_desc->n_transforms, tr.ref_bvh_depth = W.ref_depth;
            planned = true;
        } else if (!(world_desc && rc == PT_ERR_UNSUPPORTED)) {
            return rc;
        }
    }
    if (!planned) {
        if (!world_desc) {   // an empty sphere scene is an empty world
            tr = ptsel::SceneTraits{};
            tr.is_world = true;
            tr.has_caller_bvh = sphere_desc->n_bvh_nodes != 0;
        } else {
            world_traits(world_desc, W, tr);
        }
    }
    if (params->use_bvh && !tr.has_caller_bvh) return fail(PT_ERR_UNSUPPORTED, "use_bvh requested but the description has no BVH nodes");
    ptsel::Knobs knobs;
    knobs.variant = variant, knobs.blocks_per_cu = blocks_per_cu;
    const uint32_t local_rows = pt_shard_rows(params->height, 0, shard_count);
    ptsel::KernelChoice c;
    ptsel::select_kernel(tr, *params, camera->time0, camera->time1, local_rows, knobs, 4u, c);
    if (c.needs_binary_tree && tr.bin_depth == 0 && sd) {
        const AccelBuild acc = build_accel(sd, motion, plan_t_lo, plan_t_hi);
        tr.bin_depth = acc.depth, tr.bin_nodes = (uint32_t)acc.nodes.size();
        ptsel::select_kernel(tr, *params, camera->time0, camera->time1, local_rows, knobs, 4u, c);
    }
    fill_choice(c, out);
    g_debug_choice = c, g_debug_choice_valid = true;
    return PT_OK;
}

// The cell grid a sphere scene would get (include/ptgpu.h): plan_sphere_scene's GridPlan, copied out for the tests.
extern "C" int pt_debug_cell_grid(const pt_scene_desc *sphere_desc, uint32_t info16[16], uint32_t *records5x4, size_t capacity_records, uint32_t *large16) {
    if (!sphere_desc || !info16) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    SpherePlan P;
    if (int rc = plan_sphere_scene(sphere_desc, nullptr, P)) return rc;
    const GridPlan &g = P.grid;
    if (!g.ok) return fail(PT_ERR_UNSUPPORTED, "this scene gets no cell grid (fewer than 1 024 similar spheres, or not an even dense field): it walks the tree");
    union { float f; uint32_t u; } q;
    memset(info16, 0, 16 * sizeof(uint32_t));
    for (int k = 0; k < 3; ++k) {
        info16[k] = g.n[k];
        q.f = g.gmin[k], info16[8 + k] = q.u;
        q.f = g.ha[k], info16[11 + k] = q.u;
    }
    info16[3] = g.n_records, info16[4] = (uint32_t)g.large.size();
    q.f = (float)g.items_per_cell, info16[5] = q.u;
    q.f = (float)g.occupied, info16[6] = q.u;
    q.f = g.d_build, info16[14] = q.u;
    q.f = g.half_diag, info16[15] = q.u;
    if (records5x4) memcpy(records5x4, g.cells.data(), std::min<size_t>(capacity_records, g.n_records) * 5 * sizeof(uint4));
    if (large16) memcpy(large16, g.large.data(), std::min<size_t>(16, g.large.size()) * sizeof(uint32_t));
    return PT_OK;
}

// Which instantiations the choice of this thread's last successful pt_debug_select launches: the symbols of the host-side launch stubs of
// the frame kernel and (sphere kernels whose work is ordered by a measuring launch) of the measuring kernel, "" when there is none.
// tests/test_host_cpu.py holds the union over many descriptions against the stubs the shared object defines: an instantiation
// nothing selects is dead weight in a library whose build time and size are its kernels.
extern "C" int pt_debug_last_kernel_symbols(char *frame_out, char *measure_out, size_t capacity) {
    if (!frame_out || !measure_out || capacity == 0) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    if (!g_debug_choice_valid) return fail(PT_ERR_INVALID_ARG, "no successful pt_debug_select on this thread yet");
    const ptsel::KernelChoice &c = g_debug_choice;
    const void *frame = nullptr, *measure = nullptr;
    if (c.family == ptsel::Family::World) {
        frame = reinterpret_cast<const void *>(world_kernel_for(c));
    } else {
        SphereKernel f = nullptr, m = nullptr;
        sphere_kernels_for(c, &f, &m);
        frame = reinterpret_cast<const void *>(f);
        if (c.order == ptsel::Order::Measured) measure = reinterpret_cast<const void *>(m);
    }
    const auto name_of = [&](const void *fn, char *out) {
        Dl_info info{};
        out[0] = 0;
        if (fn && dladdr(fn, &info) && info.dli_sname) snprintf(out, capacity, "%s", info.dli_sname);
    };
    name_of(frame, frame_out), name_of(measure, measure_out);
    return PT_OK;
}

// ---- device self-test probes ---------------------------------------------------
namespace {
__global__ void probe_kernel(uint32_t probe, const float *in, float *out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = in[i];
    float r = 0.f, s, c;
    switch (probe) {
    case PT_PROBE_POW5: r = pow5_ref(x); break;
    case PT_PROBE_SIN: sinf_cosf_ref(x, s, c); r = s; break;
    case PT_PROBE_COS: sinf_cosf_ref(x, s, c); r = c; break;
    case PT_PROBE_LN: r = logf_ref(x); break;
    default: {
        Rng rng;
        rng_seed_from_u64(rng, (uint64_t)__float_as_uint(x));
        for (size_t k = 0; k <= (i & 15); ++k) r = rng_f32(rng);
    }
    }
    out[i] = r;
}
// Exhaustive checks of the device's own shortened math against the library lowering it replaces: every one of the 2^32 f32
// bit patterns goes through both; out[0] = mismatches (saturating at 2^24), out[1] = bit pattern of one mismatching input.
__global__ void sweep_kernel(uint32_t probe, uint32_t *result) {
    uint32_t bad = 0, where = 0;
    for (uint32_t k = 0; k < 16; ++k) {
        const uint32_t bits = (blockIdx.x * 16u + k) * 256u + threadIdx.x;
        const float x = __uint_as_float(bits);
        uint32_t got = 0, want = 0;
        if (probe == PT_PROBE_SWEEP_SQRT) got = __float_as_uint(sqrt_exact(x)), want = __float_as_uint(__builtin_sqrtf(x));
        if (probe == PT_PROBE_SWEEP_INVLEN) got = __float_as_uint(inv_sqrt_exact(x)), want = __float_as_uint(1.0f / __builtin_sqrtf(x));
        if (probe == PT_PROBE_SWEEP_DIV) {
            // 2^32 seeded (n, r) pairs: r from the divisor classes a radius can have (any sign, 2^-20 .. 2^20), n = r * a factor
            // spread over 2^-95 .. 2^4 (a point on the sphere has |n| <= ~|r|; far-origin hits overshoot), every 16th n a special
            // (0, -0, tiny, huge, inf, NaN). Lanes whose pair fails the guard take the full division on both sides.
            uint64_t z = (uint64_t)bits * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull, z = (z ^ (z >> 27)) * 0x94D049BB133111EBull, z ^= z >> 31;
            const uint32_t a = (uint32_t)z, b = (uint32_t)(z >> 32);
            const float r = __uint_as_float((a & 0x807fffffu) | ((107u + (a >> 23) % 41u) << 23));
            float f = __uint_as_float((b & 0x807fffffu) | ((32u + (b >> 23) % 100u) << 23));
            float n = r * f;
            if ((bits & 15u) == 0u) {
                const uint32_t sel = (bits >> 4) & 7u;
                n = sel == 0 ? 0.0f : sel == 1 ? -0.0f : sel == 2 ? 1.0e-40f : sel == 3 ? -3.0e38f : sel == 4 ? __uint_as_float(0x7f800000u)
                  : sel == 5 ? __uint_as_float(0x7fc00000u) : sel == 6 ? 0x1p-91f : 0x1p100f;
            }
            const volatile float yv = 1.0f / r;
            const f3 v = f3{n, n * 0.75f, f};
            const f3 g = divs3_known(v, r, yv);
            got = (__float_as_uint(g.x) ^ __float_as_uint(v.x / r)) | (__float_as_uint(g.y) ^ __float_as_uint(v.y / r)) | (__float_as_uint(g.z) ^ __float_as_uint(v.z / r));
            const bool nan_both = (g.x != g.x) == ((v.x / r) != (v.x / r)) && (g.y != g.y) == ((v.y / r) != (v.y / r)) && (g.z != g.z) == ((v.z / r) != (v.z / r));
            if ((g.x != g.x || g.y != g.y || g.z != g.z) && nan_both) {
                // NaN payloads may differ: compare the non-NaN components only
                got = 0;
                if (g.x == g.x) got |= __float_as_uint(g.x) ^ __float_as_uint(v.x / r);
                if (g.y == g.y) got |= __float_as_uint(g.y) ^ __float_as_uint(v.y / r);
                if (g.z == g.z) got |= __float_as_uint(g.z) ^ __float_as_uint(v.z / r);
            }
        }
        if (probe == PT_PROBE_SWEEP_DIVA) {
            // the sphere test's quotients n / a with a = d.d (pt_device.h DivA). Every thread checks (1) the reciprocal of one of the
            // 2^24 + 1 values of [0.5, 2] against 1.0f / a, and (2) one seeded pair: a within 64 ulps of 1 (half of the pairs) or
            // anywhere in [0.5, 2], n ANY bit pattern. Quotients of numerators below 2^-99 are only required to stay below 2^-97 (the
            // kernels reject them against t_min), numerators above 2^100 never occur (a finite discriminant keeps them below 2^66).
            const float ar = (bits & 0x1ffffffu) == 0x1000000u ? 2.0f : __uint_as_float(0x3f000000u + (bits & 0xffffffu));
            got = __float_as_uint(recip_unit_range(ar)) ^ __float_as_uint(1.0f / ar);
            uint64_t z = (uint64_t)bits * 0x9E3779B97F4A7C15ull + 0x2545F4914F6CDD1Dull;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull, z = (z ^ (z >> 27)) * 0x94D049BB133111EBull, z ^= z >> 31;
            const uint32_t ha = (uint32_t)z, hn = (uint32_t)(z >> 32);
            const float a = (ha & 1u) ? __uint_as_float(0x3f800000u - 64u + ((ha >> 1) & 127u))
                                      : __uint_as_float(0x3f000000u + ((ha >> 1) & 0xffffffu));
            float n = __uint_as_float(hn);
            if ((bits & 31u) == 0u) {
                const uint32_t sel = (bits >> 5) & 7u;
                n = sel == 0 ? 0.0f : sel == 1 ? -0.0f : sel == 2 ? 1.0e-40f : sel == 3 ? -0x1p-99f : sel == 4 ? __uint_as_float(0x7f800000u)
                  : sel == 5 ? __uint_as_float(0x7fc00000u) : sel == 6 ? 0x1p-100f : 0x1p66f;
            }
            const float q = div_by_unit_range(n, a, recip_unit_range(a)), w = n / a;
            const float an = __builtin_fabsf(n);
            if (an < 0x1p-99f && an > 0.0f) got |= (__builtin_fabsf(q) < 0x1p-97f) ? 0u : 1u;   // (tiny: magnitude only)
            else if (an >= 0x1p100f && an < __uint_as_float(0x7f800000u)) got |= 0u;                    // (never a numerator)
            else if (!(q != q && w != w)) got |= __float_as_uint(q) ^ __float_as_uint(w);
        }
        if (probe == PT_PROBE_SWEEP_RECIP) got = __float_as_uint(recip_exact(x)), want = __float_as_uint(1.0f / x);
        if (probe == PT_PROBE_SWEEP_DRAWS) {
            // every draw k * 2^-24 (low 24 bits) beside a pixel coordinate n (high 8 bits, spread over 0..8160): the fused forms
            // of pt_device.h against the reference's expressions
            const float k = (float)(bits & 0xffffffu), draw = (1.0f / 16777216.0f) * k, n = (float)((bits >> 24) * 32u);
            const uint32_t a = __float_as_uint(__builtin_fmaf(k, 1.0f / 8388608.0f, -1.0f)) ^ __float_as_uint(draw * 2.0f - 1.0f);
            const uint32_t b = __float_as_uint(k * (kPi * (1.0f / 8388608.0f))) ^ __float_as_uint(draw * 2.0f * kPi);
            const uint32_t c = __float_as_uint(__builtin_fmaf(k, 1.0f / 16777216.0f, n)) ^ __float_as_uint(n + draw);
            got = a | b | c;
        }
        if (got != want && !(got << 1 > 0xff000000u && want << 1 > 0xff000000u)) bad += 1, where = bits;   // (any NaN matches any NaN)
    }
    if (bad) {
        atomicAdd(&result[0], bad);
        result[1] = where;
    }
}
}  // namespace

extern "C" int pt_selftest_probe(int device, uint32_t probe, const float *in, float *out, size_t n) {
    if (!in || !out) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    if (probe > PT_PROBE_SWEEP_RECIP) return fail(PT_ERR_INVALID_ARG, "unknown probe %u", probe);
    if (n == 0) return PT_OK;
    if (probe >= PT_PROBE_SWEEP_SQRT && n < 2) return fail(PT_ERR_INVALID_ARG, "a sweep probe reports into out[0..1]");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PT_ERR_NO_DEVICE, "no HIP device available");
    HIP_TRY(hipSetDevice(device));
    if (probe >= PT_PROBE_SWEEP_SQRT) {
        uint32_t *d_res = nullptr, res[2] = {0, 0};
        HIP_TRY(hipMalloc((void **)&d_res, sizeof res));
        hipError_t e = hipMemset(d_res, 0, sizeof res);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(sweep_kernel, dim3(1u << 20), dim3(256), 0, 0, probe, d_res);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpy(res, d_res, sizeof res, hipMemcpyDeviceToHost);
        (void)hipFree(d_res);
        if (e != hipSuccess) return fail(PT_ERR_HIP, "sweep probe failed: %s", hipGetErrorString(e));
        out[0] = (float)(res[0] < (1u << 24) ? res[0] : (1u << 24));
        memcpy(&out[1], &res[1], 4);
        return PT_OK;
    }
    float *d_in = nullptr, *d_out = nullptr;
    HIP_TRY(hipMalloc((void **)&d_in, n * sizeof(float)));
    if (hipMalloc((void **)&d_out, n * sizeof(float)) != hipSuccess) {
        (void)hipFree(d_in);
        return fail(PT_ERR_HIP, "hipMalloc failed");
    }
    hipError_t e = hipMemcpy(d_in, in, n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, probe, d_in, d_out, n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, d_out, n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(PT_ERR_HIP, "probe failed: %s", hipGetErrorString(e));
    return PT_OK;
}

extern "C" int pt_scene_traversal_counters(pt_scene *s, uint64_t out2[2], int reset) {
    if (!s || !out2) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out2, s->d_debug + 8, 16, hipMemcpyDeviceToHost));
    if (reset) {
        HIP_TRY(hipMemset(s->d_debug + 8, 0, 16));
        HIP_TRY(hipStreamSynchronize(nullptr));
    }
    return PT_OK;
}

extern "C" int pt_scene_coop_counters(pt_scene *s, uint64_t out2[2], int reset) {
    if (!s || !out2) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out2, s->d_debug + 88, 16, hipMemcpyDeviceToHost));
    if (reset) {
        HIP_TRY(hipMemset(s->d_debug + 88, 0, 16));
        HIP_TRY(hipStreamSynchronize(nullptr));   // (later launches may use non-blocking streams, which do not wait for the NULL stream)
    }
    return PT_OK;
}

extern "C" int pt_scene_debug_tile_rays(pt_scene *s, uint32_t *rays_out, uint32_t capacity, uint32_t *n_tiles_out, uint32_t *tiles_x_out) {
    if (!s || !n_tiles_out || !tiles_x_out) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    const auto &ti = s->tile_rays_info;
    if (ti.n_tiles == 0 || !ti.frame_counted || !s->d_tile_buf)
        return fail(PT_ERR_UNSUPPORTED, "the handle's last frame did not count rays per work tile (a frame in one launch, or PT_TUNE_MEASURE_EVERY_FRAME)");
    *n_tiles_out = ti.n_tiles, *tiles_x_out = ti.tiles_x;
    if (!rays_out) return PT_OK;
    if (capacity < ti.n_tiles) return fail(PT_ERR_INVALID_ARG, "capacity %u below the frame's %u work tiles", capacity, ti.n_tiles);
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());
    // d_tile_buf = [8 scratch words | costs of the measuring launch | order | rays the frame kernel counted]
    std::vector<uint32_t> cost(ti.n_tiles), counted(ti.n_tiles);
    HIP_TRY(hipMemcpy(cost.data(), s->d_tile_buf + 8, (size_t)ti.n_tiles * 4u, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(counted.data(), s->d_tile_buf + 8 + 2 * s->d_tile_cap, (size_t)ti.n_tiles * 4u, hipMemcpyDeviceToHost));
    for (uint32_t t = 0; t < ti.n_tiles; ++t) {
        const uint32_t ty = t / ti.tiles_x, tx = t - ty * ti.tiles_x;
        // (a checkerboard's odd tiles start at their first sample in the frame kernel: their cost word holds their neighbours' mean, not rays)
        const bool first_here = ti.first_sample_in_cost && (!ti.checker || ((tx + ty) & 1u) == 0u);
        rays_out[t] = counted[t] + (first_here ? cost[t] : 0u);
    }
    return PT_OK;
}

extern "C" int pt_scene_debug_counters(pt_scene *s, uint64_t out4[4], int reset) {
    if (!s || !out4) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out4, s->d_debug, 32, hipMemcpyDeviceToHost));
    if (reset) {
        HIP_TRY(hipMemset(s->d_debug, 0, 1024));
        HIP_TRY(hipStreamSynchronize(nullptr));
    }
    return PT_OK;
}
