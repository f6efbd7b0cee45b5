// ptgpu.hip -- C-ABI implementation (include/ptgpu.h) over the gfx950 kernels.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared
// (-ffp-contract=off is REQUIRED: the reference never fuses mul+add, and every
// control-affecting value must round exactly like the CPU path.)
#include "ptgpu.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include <array>

#include "pt_build.h"
#include "pt_kernel.h"
#include "pt_world.h"

using namespace ptdev;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

// Development knobs (environment), read ONCE per process: the launch path does not call getenv.
struct DevKnobs {
    int refill = -1, ready = -1, drain = -1, pilot_div = -1, phase1 = -1, phase1_refill = -1;
    bool world_occ3 = false, debug = false, clamp_grid = false;
};
const DevKnobs &dev_knobs() {
    static const DevKnobs k = [] {
        DevKnobs d;
        if (const char *e = getenv("PTGPU_REFILL")) d.refill = atoi(e);
        if (const char *e = getenv("PTGPU_READY")) d.ready = atoi(e);
        if (const char *e = getenv("PTGPU_DRAIN")) d.drain = atoi(e);
        if (const char *e = getenv("PTGPU_PILOT_DIV")) d.pilot_div = std::max(1, atoi(e));
        if (const char *e = getenv("PTGPU_PHASE1")) d.phase1 = std::max(0, atoi(e));
        if (const char *e = getenv("PTGPU_PHASE1_REFILL")) d.phase1_refill = std::max(1, atoi(e));
        d.world_occ3 = getenv("PTGPU_WORLD_OCC3") != nullptr;
        d.debug = getenv("PTGPU_DEBUG") != nullptr;
        d.clamp_grid = getenv("PTGPU_CLAMP_GRID") != nullptr;
        return d;
    }();
    return k;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(PT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

constexpr uint32_t kLdsBudget = 160u * 1024u;       // LDS per CU on gfx950
constexpr uint32_t kTwoLaunchMinSamples = 12u;     // frames of a new view with at least this many samples measure their tiles with their own first sample
                                                   // (random_spheres 1200x800: -11 % at 8 spp, +2 % at 12, +8 % at 16, +5 % at 64 against no order / the pilot pass)
constexpr uint32_t kPhase1Samples = 1u;            // two-launch frames: samples of the first, measuring launch
constexpr uint32_t kPilotMinSamples = 12u;          // heavy-first tile ordering pays from here on (below: natural order, no measuring launch)
constexpr uint32_t kWideBlock = 768u;              // MFMA list kernels: one workgroup of 12 waves per CU (see launch())
constexpr uint32_t kLdsPerBlockMax = 96u * 1024u;   // leave room for >= 1 co-resident block's statics

}  // namespace

namespace {
// Motion of a MovingSphere entry (moving_sphere.rs:8-14) next to the pt_sphere holding centre_start / radius.
struct MotionIn {
    float delta[3];
    float time_start, inv_time_delta;
    uint32_t moving;
};
}  // namespace

struct pt_scene {
    int device = 0;
    int num_cus = 0;
    uint32_t n_spheres = 0, n_materials = 0, n_textures = 0, n_nodes = 0;
    int32_t bvh_root = -1;
    uint32_t bvh_depth = 0;
    uint32_t has_sky = 0;
    float sky[3] = {0, 0, 0};
    uint32_t has_noise = 0;
    bool palette_ok = false;   // wide MFMA kernels keep palette codes on the attenuation stack (pt_kernel.h PAL)
    bool word_ok = false;      // 4-wide tree kernels: every attenuation fits one stack word (pt_kernel.h WST)
    // device memory
    float4 *d_spheres = nullptr, *d_spheres_r2 = nullptr, *d_shade = nullptr;
    uint32_t *d_sphere_mat = nullptr;
    DMat *d_mats = nullptr;
    DTex *d_texs = nullptr;
    float4 *d_perlin_vec = nullptr;
    uint32_t *d_perlin_perm = nullptr;
    float4 *d_gate = nullptr, *d_gate_chain = nullptr;
    uint32_t *d_bvh_large = nullptr;
    uint32_t n_bvh_large = 0;
    DWideNode *d_wnodes = nullptr;
    DNode4 *d_nodes4 = nullptr;               // 4-wide internal tree (default of the tree kernels), built on the device
    uint32_t n_nodes4 = 0, depth4 = 0;
    DNode4Q *d_nodes4q = nullptr;             // ... as the packed 64-byte nodes the kernels read (pt_tree4.h), and the leaves' slot records
    float4 *d_slotrec = nullptr;
    bool tree4_packed = false;                // false: some node could not be packed (the binary tree is walked instead)
    float tree_build_ms = 0.f;                // device time of that build (HIP events)
    bool tree_on_device = false;
    // the binary tree (variant bit 2048, scenes beyond the 4-wide tree's 65535 nodes) is built on the host when first needed
    std::vector<pt_sphere> h_spheres;
    std::vector<MotionIn> h_motion;
    int32_t bin_root = -1;
    bool has_tree_items = false;              // some sphere is inside the internal tree (else every one is in d_bvh_large)
    double h_t_lo = 0.0, h_t_hi = 0.0;
    bool binary_built = false;
    uint32_t *d_leaf_rank = nullptr, *d_rank_sphere = nullptr;
    float4 *d_leafrec = nullptr, *d_shade_rank = nullptr;              // BVH worlds: sphere + gate + rank per sphere (pt_kernel.h KArgs::leafrec)
    float root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};
    // MFMA prefilter data (n_tiles == 0: prefilter not applicable to this scene)
    uint4 *d_afrag = nullptr;
    uint16_t *d_tile_sphere = nullptr;
    uint32_t *d_large = nullptr;
    uint32_t n_tiles = 0, n_large = 0;
    float c0[3] = {0, 0, 0};
    float rs2 = 0.f, m0 = 0.f, gamma = 0.f;
    uint32_t *d_cull_tab = nullptr;           // tile-culling tables (cull_axis == 3: off)
    uint32_t cull_axis = 3, cull_always = 0;
    float cull_u0 = 0.f, cull_inv_cell = 0.f, cull_rmin = 0.f, cull_rmax = 0.f, cull_cell = 0.f, rs_small = 0.f, clip_min[3] = {0, 0, 0}, clip_max[3] = {0, 0, 0};
    unsigned long long *d_debug = nullptr;    // 4 u64 counters (verify mode)
    uint32_t *d_tile_buf = nullptr;           // [8 scratch words | n tile costs | n tile order | n tile costs measured by the last frame]
    // Work order of the NEXT frame of the same view: the rays each tile really took in the last frame (same scene, camera,
    // size, samples, depth, shard). Only the order of the work depends on it, never a pixel.
    struct ViewKey {
        pt_params params;
        pt_camera cam;
        uint32_t shard_index, shard_count, variant, n_tiles;
    } hint_key{};
    bool hint_valid = false;
    uint32_t hint_scale = 1;                  // bucket width of the measured costs (samples the measuring launch traced x (depth + 1))
    uint4 *d_px_state = nullptr;              // two-launch frames: parked (xoshiro state, colour sum) per pixel, 48 B each
    size_t d_px_state_pixels = 0;
    size_t d_tile_cap = 0;
    uint32_t *d_work_counter = nullptr;       // 1 u32
    unsigned long long *d_ray_count = nullptr; // internal counter for pt_render
    float *d_frame = nullptr;                 // internal frame buffer for pt_render (host-buffer entry point)
    float *h_stage = nullptr;                 // pinned staging copy of the caller's (pageable) buffer
    size_t d_frame_floats = 0;
    float *d_gstack = nullptr;
    size_t d_gstack_floats = 0;
    uint64_t seed_base = 0x243f6a8885a308d3ull;
    // tuning
    uint32_t blocks_per_cu = 0, variant = 0;
    // worlds of Sphere + MovingSphere entries run on the sphere kernels' MOVING instantiations while the camera's
    // shutter interval stays inside [time_lo, time_hi] (the sweep the prefilter / tree were built for); the
    // general-world data below is the fallback
    bool has_motion = false;
    float4 *d_motion = nullptr;
    float time_lo = 0.f, time_hi = 0.f;
    // general world (pt_scene_create_world with non-sphere hitables): traced by pt_world_kernel
    bool is_world = false;
    uint32_t n_hitables = 0, n_world_xf = 0;
    pt_hitable *d_hitables = nullptr;
    pt_affine *d_transforms = nullptr;
    pt_bvh_node *d_ref_nodes = nullptr;   // the caller's tree as given (BVHNode::ray_hit is followed literally)
    uint4 *d_image_table = nullptr;       // Texture::Image sources: (byte offset, width, height, 0)
    uint8_t *d_image_bytes = nullptr;
    uint32_t has_image = 0;
    bool has_media = false;               // some hitable is a ConstantMedium
    uint32_t ref_bvh_depth = 0;
    // last launch
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;   // around the frame kernel alone (what rocprofv3 reports for it)
    hipEvent_t ev_pass = nullptr;                        // before the pilot pass: ev_pass..ev_stop = the whole Scene::update
    bool ev_valid = false;
    const void *attr_kern[2] = {nullptr, nullptr};       // kernels whose dynamic-LDS limit is already set to attr_lds
    uint32_t attr_lds[2] = {0, 0};
    uint32_t last_grid = 0, last_block = 0, last_lds = 0;
    // PTGPU_TIMING=1 (read once in pt_scene_create): per-wave finish times of the last launch, on this scene's device
    bool timing = false;
    unsigned long long *d_wave_end = nullptr;
};

extern "C" const char *pt_last_error(void) { return g_err; }
extern "C" const char *pt_version(void) { return "ptgpu 0.1 gfx950"; }

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

namespace {

// Depth of the supplied BVH (also validates child indices and acyclicity by
// bounding the walk); returns 0 on a malformed tree.
uint32_t bvh_depth_checked(const pt_bvh_node *nodes, uint32_t n_nodes, uint32_t n_spheres, int32_t root) {
    struct Item { int32_t node; uint32_t depth; };
    std::vector<Item> st;
    st.push_back({root, 1});
    uint32_t maxd = 0;
    size_t visited = 0;
    while (!st.empty()) {
        Item it = st.back();
        st.pop_back();
        if (it.node < 0 || (uint32_t)it.node >= n_nodes) return 0;
        if (++visited > (size_t)n_nodes * 2 + 2) return 0;  // a DAG/cycle would blow past this
        if (it.depth > maxd) maxd = it.depth;
        const int32_t ch[2] = {nodes[it.node].lhs, nodes[it.node].rhs};
        for (int c = 0; c < 2; ++c) {
            if (ch[c] >= 0) {
                st.push_back({ch[c], it.depth + 1});
            } else if ((uint32_t)(~ch[c]) >= n_spheres) {
                return 0;
            }
        }
    }
    return maxd;
}

// Texture / material tables shared by both scene constructors.
int validate_tables(uint32_t n_materials, const pt_material *materials, uint32_t n_textures, const pt_texture *textures,
                    const pt_perlin *perlin, bool allow_isotropic, bool *has_noise_out, uint32_t n_images = 0,
                    const pt_image *images = nullptr) {
    bool has_noise = false;
    for (uint32_t i = 0; i < n_images; ++i)
        if (!images || !images[i].rgb || images[i].width == 0 || images[i].height == 0 ||
            (uint64_t)images[i].width * images[i].height > (1ull << 28))
            return fail(PT_ERR_INVALID_ARG, "image %u: empty, NULL or larger than 2^28 pixels", i);
    for (uint32_t i = 0; i < n_textures; ++i) {
        const pt_texture &t = textures[i];
        if (t.kind > PT_TEX_IMAGE) return fail(PT_ERR_INVALID_ARG, "texture %u: unknown kind %u", i, t.kind);
        if (t.kind == PT_TEX_IMAGE && (t.odd < 0 || (uint32_t)t.odd >= n_images))
            return fail(PT_ERR_INVALID_ARG, "texture %u: image index %d out of range (images belong to pt_world_desc)", i, t.odd);
        if (t.kind == PT_TEX_CHECKER) {
            // arena order (storage.rs:45-48): sub-textures are allocated before the checker that
            // references them; requiring odd/even < i also guarantees termination on device.
            if (t.odd < 0 || t.even < 0 || (uint32_t)t.odd >= i || (uint32_t)t.even >= i)
                return fail(PT_ERR_INVALID_ARG, "texture %u: checker children must be earlier textures", i);
        }
        if (t.kind == PT_TEX_NOISE) has_noise = true;
    }
    if (has_noise && !perlin) return fail(PT_ERR_INVALID_ARG, "noise texture without perlin tables");
    for (uint32_t i = 0; i < n_materials; ++i) {
        const pt_material &m = materials[i];
        if (m.kind > (allow_isotropic ? (uint32_t)PT_MAT_ISOTROPIC : (uint32_t)PT_MAT_DIFFUSE_LIGHT))
            return fail(PT_ERR_INVALID_ARG, "material %u: unknown kind %u", i, m.kind);
        if (m.kind == PT_MAT_LAMBERTIAN || m.kind == PT_MAT_DIFFUSE_LIGHT || m.kind == PT_MAT_ISOTROPIC) {
            if (m.texture < 0 || (uint32_t)m.texture >= n_textures)
                return fail(PT_ERR_INVALID_ARG, "material %u: texture index %d out of range", i, m.texture);
        }
    }
    if (perlin)
        for (int i = 0; i < 256; ++i)
            if (perlin->perm_x[i] > 255 || perlin->perm_y[i] > 255 || perlin->perm_z[i] > 255)
                return fail(PT_ERR_INVALID_ARG, "perlin permutation entry > 255");
    *has_noise_out = has_noise;
    return PT_OK;
}

template <typename T>
int upload(T **dst, const void *src, size_t count) {
    HIP_TRY(hipMalloc((void **)dst, count * sizeof(T) > 0 ? count * sizeof(T) : sizeof(T)));
    if (count) HIP_TRY(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    return PT_OK;
}

}  // namespace


namespace {


// Conservative bound of sphere i over every ray time in [t_lo, t_hi]: centre of the swept segment and the
// half-length to add to |radius| (zero for plain spheres).
struct Sweep {
    double c[3];
    double half;
};
Sweep sweep_of(const pt_sphere &p, const MotionIn *m, double t_lo, double t_hi) {
    Sweep w{{p.cx, p.cy, p.cz}, 0.0};
    if (!m || !m->moving) return w;
    double s0 = (t_lo - (double)m->time_start) * (double)m->inv_time_delta, s1 = (t_hi - (double)m->time_start) * (double)m->inv_time_delta;
    if (s0 > s1) std::swap(s0, s1);
    const double padp = 1e-4 * (1.0 + std::fabs(s0) + std::fabs(s1));  // f32 rounding of time and of (time - t0) * inv
    s0 -= padp, s1 += padp;
    const double mid = 0.5 * (s0 + s1), len = std::sqrt((double)m->delta[0] * m->delta[0] + (double)m->delta[1] * m->delta[1] + (double)m->delta[2] * m->delta[2]);
    for (int k = 0; k < 3; ++k) w.c[k] += mid * (double)m->delta[k];
    w.half = 0.5 * (s1 - s0) * len * (1.0 + 1e-6) + 1e-6 * (std::fabs(w.c[0]) + std::fabs(w.c[1]) + std::fabs(w.c[2]));
    return w;
}

// ---- MFMA prefilter preparation (DESIGN.md "MFMA prefilter") ------------------------------------
// Spheres whose centre/radius stay within the f16 feature range relative to the set's centroid are
// packed 32 per tile into A fragments of v_mfma_f32_32x32x16_f16; the rest ("large", e.g. the
// r = 1000 ground sphere) are tested exactly for every ray. Features are computed in binary64 from
// the exact f32 inputs and split into hi/lo f16.
constexpr double kFeatRange = 48.0;   // |c - c0| + |r| bound for prefiltered spheres (features <= 2304 < 65504)
constexpr double kRadiusMax = 8.0;
constexpr uint32_t kMaxLarge = 8;

struct MfmaPrep {
    std::vector<uint16_t> afrag;  // f16 bit patterns, [tile][chunk][lane][8]
    std::vector<uint16_t> tile_sphere;
    std::vector<uint32_t> large;
    float c0[3] = {0, 0, 0};
    double rs = 0.0;
    double sweep_ratio = 0.0;  // max over prefiltered spheres of (swept half-length / |radius|)
    uint32_t n_tiles = 0;
    // tile culling: sort axis (3 = off), tiles that are always run, lookup tables (kCullCells cells), padded box of the sorted spheres
    uint32_t cull_axis = 3, cull_always = 0;
    std::vector<uint32_t> cull_tab;
    float cull_u0 = 0.f, cull_inv_cell = 0.f, cull_rmin = 0.f, cull_rmax = 0.f, cull_cell = 0.f, clip_min[3] = {0, 0, 0}, clip_max[3] = {0, 0, 0};
};

uint16_t f16_bits(_Float16 h) {
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}

bool prepare_mfma(const pt_scene_desc *desc, const MotionIn *motion, double t_lo, double t_hi, MfmaPrep &out) {
    const uint32_t n = desc->n_spheres;
    if (n > 0xfff0u) return false;
    // centroid of the moderate-radius spheres, rounded to f32 (c0 must be exactly what the device subtracts)
    double cx = 0, cy = 0, cz = 0;
    uint32_t m = 0;
    std::vector<Sweep> sw(n);
    for (uint32_t i = 0; i < n; ++i) sw[i] = sweep_of(desc->spheres[i], motion ? &motion[i] : nullptr, t_lo, t_hi);
    for (uint32_t i = 0; i < n; ++i) {
        const pt_sphere &p = desc->spheres[i];
        if (std::fabs((double)p.radius) + sw[i].half <= kRadiusMax && std::isfinite(sw[i].c[0] + sw[i].c[1] + sw[i].c[2] + p.radius + sw[i].half)) {
            cx += sw[i].c[0], cy += sw[i].c[1], cz += sw[i].c[2], ++m;
        }
    }
    if (m == 0) return false;
    out.c0[0] = (float)(cx / m), out.c0[1] = (float)(cy / m), out.c0[2] = (float)(cz / m);
    std::vector<uint32_t> small;
    for (uint32_t i = 0; i < n; ++i) {
        const pt_sphere &p = desc->spheres[i];
        const double dx = sw[i].c[0] - out.c0[0], dy = sw[i].c[1] - out.c0[1], dz = sw[i].c[2] - out.c0[2];
        const double rad = std::fabs((double)p.radius) + sw[i].half;
        const double reach = std::sqrt(dx * dx + dy * dy + dz * dz) + rad;
        if (std::isfinite(reach) && rad <= kRadiusMax && reach <= kFeatRange && (sw[i].half == 0.0 || std::fabs((double)p.radius) > 0.0)) {
            small.push_back(i);
            if (reach > out.rs) out.rs = reach;
            if (sw[i].half > 0.0) out.sweep_ratio = std::max(out.sweep_ratio, sw[i].half / std::fabs((double)p.radius));
        } else {
            out.large.push_back(i);
        }
    }
    if (out.large.size() > kMaxLarge || small.size() < 32) return false;
    out.n_tiles = (uint32_t)((small.size() + 31) / 32);
    // ---- tile culling: give the tiles a spatial meaning -------------------------------------------------------
    // Spheres of ordinary size are sorted along one axis, so a tile of 32 consecutive ones covers a short interval of
    // that axis and a wave can skip the tiles no ray of it comes near (pt_kernel.h lane_tile_mask). Oversized spheres
    // go last; a tile holding any of them is always run. The axis is the one on which the tiles come out narrowest.
    // The order of the prefiltered spheres never affects the image (closest hit by (t, index)).
    if (out.n_tiles >= 4 && out.n_tiles <= 32) {
        auto radius_of = [&](uint32_t i) { return std::fabs((double)desc->spheres[i].radius) + sw[i].half; };
        std::vector<double> rr;
        for (uint32_t i : small) rr.push_back(radius_of(i));
        std::nth_element(rr.begin(), rr.begin() + rr.size() / 2, rr.end());
        const double r_med = rr[rr.size() / 2];
        std::vector<uint32_t> regular, big;
        for (uint32_t i : small) (radius_of(i) <= 3.0 * r_med ? regular : big).push_back(i);
        const size_t full_tiles = regular.size() / 32;   // tiles made of sorted spheres only
        int forced = -1;
        if (const char *e = getenv("PTGPU_CULL_AXIS")) forced = atoi(e);
        double best_score = 1e300;
        int best_axis = 3;
        for (int ax = 0; ax < 3 && full_tiles >= 3; ++ax) {
            std::vector<uint32_t> ord = regular;
            std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return sw[a].c[ax] < sw[b].c[ax]; });
            double lo_all = 1e300, hi_all = -1e300, sum = 0;
            for (size_t T = 0; T < full_tiles; ++T) {
                double lo = 1e300, hi = -1e300;
                for (size_t j = T * 32; j < T * 32 + 32; ++j)
                    lo = std::min(lo, sw[ord[j]].c[ax] - radius_of(ord[j])), hi = std::max(hi, sw[ord[j]].c[ax] + radius_of(ord[j]));
                sum += hi - lo, lo_all = std::min(lo_all, lo), hi_all = std::max(hi_all, hi);
            }
            const double score = sum / ((double)full_tiles * std::max(hi_all - lo_all, 1e-30));   // mean tile extent / set extent
            if ((forced < 0 && score < best_score) || forced == ax) best_score = score, best_axis = ax;
        }
        if (forced == 3) best_axis = 3;
        if (best_axis < 3 && (best_score < 0.5 || forced >= 0)) {
            const int ax = best_axis;
            std::stable_sort(regular.begin(), regular.end(), [&](uint32_t a, uint32_t b) { return sw[a].c[ax] < sw[b].c[ax]; });
            small = regular;
            small.insert(small.end(), big.begin(), big.end());
            std::vector<double> lo(out.n_tiles, 0.0), hi(out.n_tiles, 0.0);
            double bmin[3] = {1e300, 1e300, 1e300}, bmax[3] = {-1e300, -1e300, -1e300};
            for (uint32_t T = 0; T < out.n_tiles; ++T) {
                if (T >= full_tiles) {
                    out.cull_always |= 1u << T;
                    continue;
                }
                lo[T] = 1e300, hi[T] = -1e300;
                for (size_t j = (size_t)T * 32; j < (size_t)T * 32 + 32; ++j) {
                    const uint32_t i = small[j];
                    const double r = radius_of(i);
                    lo[T] = std::min(lo[T], sw[i].c[ax] - r), hi[T] = std::max(hi[T], sw[i].c[ax] + r);
                    for (int k = 0; k < 3; ++k) bmin[k] = std::min(bmin[k], sw[i].c[k] - r), bmax[k] = std::max(bmax[k], sw[i].c[k] + r);
                }
            }
            // the box and the tile intervals are padded by 2e-3 + 1e-5 of their magnitude: the reference's f32 hit test
            // sees a sphere inflated by ~1e-6 relative, and a moving sphere's sweep bound already carries its own slack
            for (int k = 0; k < 3; ++k) {
                const double pad = 2e-3 + 1e-5 * std::max(std::fabs(bmin[k]), std::fabs(bmax[k]));
                out.clip_min[k] = std::nextafter((float)(bmin[k] - pad), -3.0e38f);
                out.clip_max[k] = std::nextafter((float)(bmax[k] + pad), 3.0e38f);
            }
            out.cull_axis = (uint32_t)ax;
            out.cull_u0 = out.clip_min[ax];
            const double cell = std::max(((double)out.clip_max[ax] - (double)out.clip_min[ax]) / (double)kCullCells, 1e-30);
            double rmin = 1e300, rmax = 0.0;
            for (size_t j = 0; j < full_tiles * 32; ++j) {
                const double r = std::fabs((double)desc->spheres[small[j]].radius);
                rmin = std::min(rmin, r), rmax = std::max(rmax, r);
            }
            out.cull_rmin = (float)rmin, out.cull_rmax = (float)rmax, out.cull_cell = (float)cell;
            out.cull_inv_cell = (float)(1.0 / cell);
            out.cull_tab.assign(2 * kCullCells, 0u);
            for (int c = 0; c < kCullCells; ++c) {
                // cell c as the DEVICE sees it: a coordinate u lands in cell clamp(int((u - u0) * inv_cell)); one extra cell
                // of slack on each side covers the f32 rounding of that expression
                const double c_lo = (c == 0) ? -1e300 : (double)out.cull_u0 + (c - 1) * cell;
                const double c_hi = (c == kCullCells - 1) ? 1e300 : (double)out.cull_u0 + (c + 2) * cell;
                for (uint32_t T = 0; T < (uint32_t)full_tiles; ++T) {
                    const double pad = 2e-3 + 1e-5 * std::max(std::fabs(lo[T]), std::fabs(hi[T]));
                    if (hi[T] + pad >= c_lo) out.cull_tab[c] |= 1u << T;        // tiles reaching cell c or beyond
                    if (lo[T] - pad <= c_hi) out.cull_tab[kCullCells + c] |= 1u << T;   // tiles starting at cell c or before
                }
            }
        }
    }
    out.tile_sphere.assign((size_t)out.n_tiles * 32, 0xffffu);
    out.afrag.assign((size_t)out.n_tiles * 2 * 64 * 8, 0);
    for (uint32_t T = 0; T < out.n_tiles; ++T) {
        for (uint32_t row = 0; row < 32; ++row) {
            const size_t j = (size_t)T * 32 + row;
            double S[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 60000.0};  // padding: S.R = -a * 60000 < thr, never a candidate
            if (j < small.size()) {
                const pt_sphere &p = desc->spheres[small[j]];
                out.tile_sphere[j] = (uint16_t)small[j];
                const Sweep &w = sw[small[j]];
                const double x = w.c[0] - out.c0[0], y = w.c[1] - out.c0[1], z = w.c[2] - out.c0[2];
                const volatile float r2s = p.radius * p.radius;  // sphere.rs:36 (the reference squares in f32)
                // a moving sphere enters the prefilter as the sphere bounding its sweep: a line that meets the
                // sphere at any covered time passes within |r| + half of the sweep's midpoint
                const double rb = std::fabs((double)p.radius) + w.half;
                const double r2f = w.half > 0.0 ? rb * rb : (double)r2s;
                S[0] = x * x, S[1] = y * y, S[2] = z * z, S[3] = x * y, S[4] = x * z, S[5] = y * z;
                S[6] = x, S[7] = y, S[8] = z, S[9] = x * x + y * y + z * z - (double)r2f;
            }
            _Float16 slot[32];
            for (int f = 0; f < 10; ++f) {  // fragments hold -S: the GEMM yields thr - S.R (negative = candidate)
                const _Float16 h = (_Float16)(-S[f]);
                const _Float16 l = (_Float16)(-S[f] - (double)h);
                slot[f] = h;        // x Rh
                slot[10 + f] = h;   // x Rl
                slot[20 + f] = l;   // x Rh
            }
            slot[30] = (_Float16)1.0, slot[31] = (_Float16)1.0;  // x thr_hi, x thr_lo
            for (int c = 0; c < 2; ++c)
                for (int half = 0; half < 2; ++half) {
                    const uint32_t lane = row + 32 * half;
                    for (int e = 0; e < 8; ++e)
                        out.afrag[(((size_t)T * 2 + c) * 64 + lane) * 8 + e] = f16_bits(slot[c * 16 + half * 8 + e]);
                }
        }
    }
    return true;
}

}  // namespace


namespace {

// ---- internal BVH for BVH mode ------------------------------------------------------------------
// The caller's tree (bvh.rs:64-94: random split axis, median split) defines the RESULT of BVHNode::ray_hit
// but can be arbitrarily bad for traversal (random_spheres / perlin_spheres: every sphere has the same y, so
// a third of the levels do not separate anything: ~1800 node visits per ray in the 10k-sphere scene).
// pt_scene_create therefore builds its own tree (longest-axis median split over sphere centres; spheres
// with a huge radius are kept out and tested for every ray) and keeps, per sphere, the AABB of its parent in
// the CALLER's tree, which is all that is needed to reproduce the reference's accept/reject decision.
struct AccelBuild {
    std::vector<DWideNode> nodes;
    std::vector<uint32_t> large;
    int32_t root = -1;
    uint32_t depth = 0;
};

struct AccelItem {
    uint32_t sphere;
    float c[3], mn[3], mx[3], r, signed_r;
    float c_start[3];  // the sphere as stored (centre_start for a MovingSphere): what the leaf slot carries
};

struct AccelRef {
    int32_t ref;
    float mn[3], mx[3], rmin;
    uint32_t depth;
    float sph[4];  // leaves: the sphere as given (centre, signed radius)
};

AccelRef accel_build(std::vector<AccelItem> &items, size_t lo, size_t hi, std::vector<DWideNode> &nodes) {
    if (hi - lo == 1) {
        const AccelItem &it = items[lo];
        AccelRef r{~(int32_t)it.sphere, {it.mn[0], it.mn[1], it.mn[2]}, {it.mx[0], it.mx[1], it.mx[2]}, it.r, 0,
                   {it.c_start[0], it.c_start[1], it.c_start[2], it.signed_r}};
        return r;
    }
    float cmin[3] = {3e38f, 3e38f, 3e38f}, cmax[3] = {-3e38f, -3e38f, -3e38f};
    for (size_t i = lo; i < hi; ++i)
        for (int k = 0; k < 3; ++k) cmin[k] = std::min(cmin[k], items[i].c[k]), cmax[k] = std::max(cmax[k], items[i].c[k]);
    int axis = 0;
    for (int k = 1; k < 3; ++k)
        if (cmax[k] - cmin[k] > cmax[axis] - cmin[axis]) axis = k;
    const size_t mid = lo + (hi - lo) / 2;
    std::nth_element(items.begin() + lo, items.begin() + mid, items.begin() + hi,
                     [axis](const AccelItem &a, const AccelItem &b) { return a.c[axis] < b.c[axis] || (a.c[axis] == b.c[axis] && a.sphere < b.sphere); });
    const AccelRef l = accel_build(items, lo, mid, nodes), r = accel_build(items, mid, hi, nodes);
    DWideNode w;
    memset(&w, 0, sizeof w);
    // inner children are stored as box centre / half extent (the slab test then needs no midpoint arithmetic);
    // the half extent is rounded up until centre -+ half covers the box in exact arithmetic
    auto centre_half = [](const float mn[3], const float mx[3], float c[3], float h[3]) {
        for (int k = 0; k < 3; ++k) {
            c[k] = (float)(0.5 * ((double)mn[k] + (double)mx[k]));
            h[k] = (float)(0.5 * ((double)mx[k] - (double)mn[k]));
            while ((double)c[k] - (double)h[k] > (double)mn[k] || (double)c[k] + (double)h[k] < (double)mx[k]) h[k] = std::nextafter(h[k], 3.0e38f);
        }
    };
    centre_half(l.mn, l.mx, w.lmin, w.lmax);
    centre_half(r.mn, r.mx, w.rmin, w.rmax);
    w.lhs = l.ref, w.rhs = r.ref;
    // a leaf child needs no box: its slot carries the sphere (centre, signed radius) so a leaf test costs no fetch
    if (l.ref < 0) memcpy(w.lmin, l.sph, 12), w.lmax[0] = l.sph[3];
    if (r.ref < 0) memcpy(w.rmin, r.sph, 12), w.rmax[0] = r.sph[3];
    const float inv_l = 1.0f / l.rmin, inv_r = 1.0f / r.rmin;  // the pad of the conservative box test divides by the smallest radius below
    memcpy(&w.pad0, &inv_l, 4), memcpy(&w.pad1, &inv_r, 4);
    nodes.push_back(w);
    AccelRef out{};
    out.ref = (int32_t)nodes.size() - 1;
    for (int k = 0; k < 3; ++k) out.mn[k] = std::min(l.mn[k], r.mn[k]), out.mx[k] = std::max(l.mx[k], r.mx[k]);
    out.rmin = std::min(l.rmin, r.rmin);
    out.depth = 1 + std::max(l.depth, r.depth);
    return out;
}

// Spheres that go into the internal tree (with the box of their whole sweep when they move); the rest -- huge,
// degenerate or non-finite ones -- are returned in `large` and tested for every ray.
std::vector<AccelItem> accel_items(const pt_scene_desc *desc, const MotionIn *motion, double t_lo, double t_hi, std::vector<uint32_t> &large) {
    std::vector<float> radii;
    for (uint32_t i = 0; i < desc->n_spheres; ++i) radii.push_back(std::fabs(desc->spheres[i].radius));
    std::vector<float> sorted = radii;
    std::nth_element(sorted.begin(), sorted.begin() + sorted.size() / 2, sorted.end());
    const float median = sorted[sorted.size() / 2];
    std::vector<AccelItem> items;
    for (uint32_t i = 0; i < desc->n_spheres; ++i) {
        const pt_sphere &p = desc->spheres[i];
        const float r = radii[i];
        const bool finite = std::isfinite(p.cx) && std::isfinite(p.cy) && std::isfinite(p.cz) && std::isfinite(r);
        // (radii below 1e-5 stay out of the tree as well: its packed nodes hold the pad constant 6e-6 / r_min as a power of two <= 1)
        if (!finite || r > 16.0f * median || !(r > 1.0e-5f)) {
            large.push_back(i);
            continue;
        }
        AccelItem it{i, {p.cx, p.cy, p.cz}, {p.cx - r, p.cy - r, p.cz - r}, {p.cx + r, p.cy + r, p.cz + r}, r, p.radius, {p.cx, p.cy, p.cz}};
        if (motion && motion[i].moving) {  // box the whole sweep; the leaf slot keeps centre_start (sphere_at moves it)
            const Sweep w = sweep_of(p, &motion[i], t_lo, t_hi);
            const double len = std::sqrt((double)motion[i].delta[0] * motion[i].delta[0] + (double)motion[i].delta[1] * motion[i].delta[1] +
                                         (double)motion[i].delta[2] * motion[i].delta[2]);
            for (int k = 0; k < 3; ++k) {
                const double ext = len > 0.0 ? w.half * std::fabs((double)motion[i].delta[k]) / len : 0.0;
                it.mn[k] = (float)(w.c[k] - ext - r - 1e-5 * (1.0 + std::fabs(w.c[k])));
                it.mx[k] = (float)(w.c[k] + ext + r + 1e-5 * (1.0 + std::fabs(w.c[k])));
                it.c[k] = (float)w.c[k];
            }
        }
        items.push_back(it);
    }
    if (items.size() < 2) {  // degenerate: everything is tested directly
        for (const AccelItem &it : items) large.push_back(it.sphere);
        items.clear();
    }
    return items;
}

AccelBuild build_accel(const pt_scene_desc *desc, const MotionIn *motion, double t_lo, double t_hi) {
    AccelBuild out;
    std::vector<AccelItem> items = accel_items(desc, motion, t_lo, t_hi, out.large);
    if (items.empty()) return out;
    const AccelRef root = accel_build(items, 0, items.size(), out.nodes);
    out.root = root.ref;
    out.depth = root.depth;
    return out;
}

// ---- 4-wide internal tree: host restatement of the DEVICE build (pt_build.hip; rules in pt_tree4.h) ----------------
// Used as the reference the device build is tested against (PTGPU_HOST_BUILD=1 selects it) and when the device build
// cannot run. Level by level like the device: order every segment along its longest centroid axis (stable, by the
// orderable coordinate), cut, order the halves that are cut again, emit the children; boxes bottom-up at the end.
struct Tree4Host {
    std::vector<DNode4> nodes;
    uint32_t depth = 0;
};

void tree4_order_range(std::vector<TreeItem> &items, size_t lo, size_t hi) {
    float cmin[3] = {0, 0, 0}, cmax[3] = {0, 0, 0};
    uint32_t umin[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, umax[3] = {0, 0, 0};
    for (size_t i = lo; i < hi; ++i)
        for (int k = 0; k < 3; ++k) {
            const uint32_t u = tree_orderable(items[i].c[k]);
            if (u < umin[k]) umin[k] = u, cmin[k] = items[i].c[k];
            if (u > umax[k]) umax[k] = u, cmax[k] = items[i].c[k];
        }
    const volatile float ex = cmax[0] - cmin[0], ey = cmax[1] - cmin[1], ez = cmax[2] - cmin[2];
    const int axis = tree_axis_of_extents(ex, ey, ez);
    std::stable_sort(items.begin() + lo, items.begin() + hi,
                     [axis](const TreeItem &a, const TreeItem &b) { return tree_orderable(a.c[axis]) < tree_orderable(b.c[axis]); });
}

Tree4Host tree4_build_host(std::vector<TreeItem> items) {
    Tree4Host out;
    struct Seg { uint32_t lo, hi, node; };
    std::vector<Seg> segs{{0u, (uint32_t)items.size(), 0u}};
    std::vector<std::array<uint32_t, 4>> leaf_item(1);
    std::vector<std::pair<uint32_t, uint32_t>> levels;   // (first node, count)
    out.nodes.resize(1);
    while (!segs.empty()) {
        for (const Seg &sg : segs) tree4_order_range(items, sg.lo, sg.hi);
        for (const Seg &sg : segs) {
            const TreePlan pl = tree_plan(sg.hi - sg.lo);
            if (pl.half == 2u) tree4_order_range(items, sg.lo, sg.lo + pl.cut[2]);
            if (pl.c - pl.half == 2u) tree4_order_range(items, sg.lo + pl.cut[pl.half], sg.hi);
        }
        levels.push_back({(uint32_t)out.nodes.size() - (uint32_t)segs.size(), (uint32_t)segs.size()});
        std::vector<Seg> next;
        for (const Seg &sg : segs) {
            const TreePlan pl = tree_plan(sg.hi - sg.lo);
            uint32_t slot = 0;
            int32_t child[4] = {kNoChild4, kNoChild4, kNoChild4, kNoChild4};
            std::array<uint32_t, 4> li{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
            for (int pass = 0; pass < 2; ++pass)
                for (uint32_t j = 0; j < pl.c; ++j) {
                    const uint32_t a = sg.lo + pl.cut[j], b = sg.lo + pl.cut[j + 1];
                    if ((b - a > 1u) != (pass == 0)) continue;
                    if (pass == 0) {
                        const uint32_t node = (uint32_t)out.nodes.size();
                        out.nodes.emplace_back();
                        leaf_item.emplace_back();
                        child[slot] = (int32_t)node;
                        next.push_back({a, b, node});
                    } else {
                        child[slot] = ~(int32_t)items[a].sphere;
                        li[slot] = a;
                    }
                    ++slot;
                }
            memcpy(out.nodes[sg.node].child, child, sizeof child);
            leaf_item[sg.node] = li;
        }
        segs.swap(next);
    }
    std::vector<TreeBox> box(out.nodes.size());
    for (size_t l = levels.size(); l-- > 0;)
        for (uint32_t node = levels[l].first; node < levels[l].first + levels[l].second; ++node) {
            DNode4 w = out.nodes[node];
            TreeBox ch[4];
            uint32_t c = 0;
            for (uint32_t j = 0; j < 4u && w.child[j] != kNoChild4; ++j, ++c) {
                if (w.child[j] >= 0) {
                    ch[j] = box[w.child[j]];
                } else {
                    const TreeItem &it = items[leaf_item[node][j]];
                    for (int k = 0; k < 3; ++k) ch[j].mn[k] = it.mn[k], ch[j].mx[k] = it.mx[k];
                    ch[j].rmin = it.r;
                }
            }
            box[node] = tree_finish_node(w, ch, c);
            out.nodes[node] = w;
        }
    out.depth = (uint32_t)levels.size();
    return out;
}

}  // namespace

namespace {
int create_sphere_scene(const pt_scene_desc *desc, const MotionIn *motion, int device, pt_scene **scene_out);
}

extern "C" int pt_scene_create(const pt_scene_desc *desc, int device, pt_scene **scene_out) {
    if (desc && scene_out && desc->n_spheres == 0) {  // an empty HitableList is a valid world: every ray sees the sky
        pt_world_desc w{};
        w.n_materials = desc->n_materials, w.materials = desc->materials;
        w.n_textures = desc->n_textures, w.textures = desc->textures, w.perlin = desc->perlin;
        w.n_bvh_nodes = desc->n_bvh_nodes, w.bvh_nodes = desc->bvh_nodes, w.bvh_root = desc->bvh_root;
        w.has_sky = desc->has_sky;
        memcpy(w.sky, desc->sky, sizeof w.sky);
        return pt_scene_create_world(&w, device, scene_out);
    }
    return create_sphere_scene(desc, nullptr, device, scene_out);
}

namespace {
// `motion` (optional, n_spheres entries): MovingSphere parameters of the entries that move; desc->spheres then
// holds centre_start / radius for them.
int create_sphere_scene(const pt_scene_desc *desc, const MotionIn *motion, int device, pt_scene **scene_out) {
    if (!desc || !scene_out) return fail(PT_ERR_INVALID_ARG, "desc/scene_out is NULL");
    *scene_out = nullptr;
    if (desc->n_spheres == 0 || !desc->spheres || !desc->sphere_material)
        return fail(PT_ERR_INVALID_ARG, "scene has no spheres");
    if (desc->n_materials == 0 || !desc->materials) return fail(PT_ERR_INVALID_ARG, "scene has no materials");
    if (desc->n_textures && !desc->textures) return fail(PT_ERR_INVALID_ARG, "textures is NULL");
    if (desc->n_spheres > 0x7fffffffu) return fail(PT_ERR_INVALID_ARG, "too many spheres");
    bool has_noise = false;
    if (int rc = validate_tables(desc->n_materials, desc->materials, desc->n_textures, desc->textures, desc->perlin, motion != nullptr, &has_noise)) return rc;
    for (uint32_t i = 0; i < desc->n_spheres; ++i)
        if (desc->sphere_material[i] >= desc->n_materials)
            return fail(PT_ERR_INVALID_ARG, "sphere %u: material index out of range", i);
    uint32_t bvh_depth = 0;
    if (desc->n_bvh_nodes) {
        if (!desc->bvh_nodes) return fail(PT_ERR_INVALID_ARG, "bvh_nodes is NULL");
        bvh_depth = bvh_depth_checked(desc->bvh_nodes, desc->n_bvh_nodes, desc->n_spheres, desc->bvh_root);
        if (bvh_depth == 0) return fail(PT_ERR_INVALID_ARG, "malformed BVH (bad child index or cycle)");
        // (the caller's depth is irrelevant: traversal runs over the internal tree built below)
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PT_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(PT_ERR_INVALID_ARG, "device %d out of range (%d devices)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));

    pt_scene *s = new (std::nothrow) pt_scene();
    if (!s) return fail(PT_ERR_INVALID_ARG, "out of host memory");
    s->device = device;
    s->num_cus = prop.multiProcessorCount;
    s->n_spheres = desc->n_spheres;
    s->n_materials = desc->n_materials;
    s->n_textures = desc->n_textures;
    s->n_nodes = desc->n_bvh_nodes;
    s->bvh_root = desc->n_bvh_nodes ? desc->bvh_root : -1;
    s->bvh_depth = bvh_depth;
    s->has_sky = desc->has_sky ? 1u : 0u;
    memcpy(s->sky, desc->sky, sizeof s->sky);
    s->has_noise = has_noise ? 1u : 0u;
    // the time interval the moving entries are defined over: the sweeps are bounded for ray times inside it
    double t_lo = 0.0, t_hi = 0.0;
    if (motion) {
        bool first = true;
        for (uint32_t i = 0; i < desc->n_spheres; ++i) {
            if (!motion[i].moving) continue;
            const double a0 = motion[i].time_start, a1 = a0 + 1.0 / (double)motion[i].inv_time_delta;
            if (!std::isfinite(a0) || !std::isfinite(a1)) {
                delete s;
                return fail(PT_ERR_UNSUPPORTED, "moving sphere %u has a degenerate time interval", i);
            }
            t_lo = first ? std::min(a0, a1) : std::min(t_lo, std::min(a0, a1));
            t_hi = first ? std::max(a0, a1) : std::max(t_hi, std::max(a0, a1));
            first = false;
        }
        s->has_motion = !first;
        if (first) motion = nullptr;
        s->time_lo = (float)t_lo, s->time_hi = (float)t_hi;
    }

    // flatten to the device layouts
    const uint32_t n_pad = (desc->n_spheres + kScanUnroll - 1) / kScanUnroll * kScanUnroll;
    std::vector<float4> sph(desc->n_spheres), sph_r2(n_pad, make_float4(3.0e38f, 3.0e38f, 3.0e38f, 0.0f));
    for (uint32_t i = 0; i < desc->n_spheres; ++i) {
        const pt_sphere &p = desc->spheres[i];
        sph[i] = make_float4(p.cx, p.cy, p.cz, p.radius);
        const volatile float r2 = p.radius * p.radius;  // sphere.rs:36, one f32 rounding
        sph_r2[i] = make_float4(p.cx, p.cy, p.cz, r2);
    }
    // per-sphere shading records (one 64-byte fetch per hit)
    std::vector<float4> shade(4 * (size_t)desc->n_spheres);
    bool palette_ok = true;   // every scattering material's attenuation is a per-sphere constant or one of two checker colours
    for (uint32_t i = 0; i < desc->n_spheres; ++i) {
        const pt_sphere &p = desc->spheres[i];
        const pt_material &m = desc->materials[desc->sphere_material[i]];
        uint32_t flags = 0;
        float4 qa = make_float4(m.albedo[0], m.albedo[1], m.albedo[2], 0.f), qb = make_float4(0, 0, 0, 0);
        if (m.kind == PT_MAT_LAMBERTIAN || m.kind == PT_MAT_DIFFUSE_LIGHT) {
            const pt_texture &t = desc->textures[m.texture];
            if (t.kind == PT_TEX_CONSTANT) {
                flags = kShadeConst;
                qa = make_float4(t.color[0], t.color[1], t.color[2], 0.f);
            } else if (t.kind == PT_TEX_CHECKER && desc->textures[t.odd].kind == PT_TEX_CONSTANT &&
                       desc->textures[t.even].kind == PT_TEX_CONSTANT) {
                flags = kShadeChecker2;
                const pt_texture &o = desc->textures[t.odd], &e = desc->textures[t.even];
                qa = make_float4(o.color[0], o.color[1], o.color[2], 0.f);
                qb = make_float4(e.color[0], e.color[1], e.color[2], 0.f);
            } else if (t.kind == PT_TEX_NOISE) {
                flags = kShadeNoise;
                qa = make_float4(t.scale, 0.f, 0.f, 0.f);
            }
        }
        if (m.kind == PT_MAT_LAMBERTIAN && (flags & (kShadeConst | kShadeChecker2)) == 0) palette_ok = false;
        if (m.kind > PT_MAT_DIFFUSE_LIGHT) palette_ok = false;
        union { uint32_t u; float f; } k{m.kind}, fl{flags}, tx{(uint32_t)m.texture};
        shade[4 * i] = make_float4(p.cx, p.cy, p.cz, p.radius);
        shade[4 * i + 1] = make_float4(k.f, fl.f, tx.f, m.param);
        shade[4 * i + 2] = qa;
        shade[4 * i + 3] = qb;
    }
    s->palette_ok = palette_ok;
    {   // can every attenuation be one stack word (pt_kernel.h WST)? Noise -> its grey value; Constant / Checker2 / metal / glass -> a code
        bool ok = true;
        for (uint32_t i = 0; i < desc->n_spheres && ok; ++i) {
            const pt_material &m = desc->materials[desc->sphere_material[i]];
            if (m.kind == PT_MAT_LAMBERTIAN) {
                const pt_texture &t = desc->textures[m.texture];
                ok = t.kind == PT_TEX_NOISE || t.kind == PT_TEX_CONSTANT ||
                     (t.kind == PT_TEX_CHECKER && desc->textures[t.odd].kind == PT_TEX_CONSTANT && desc->textures[t.even].kind == PT_TEX_CONSTANT);
            } else if (m.kind > PT_MAT_DIFFUSE_LIGHT) {
                ok = false;
            }
        }
        s->word_ok = ok && desc->n_spheres < 0xFFFFFu;
    }
    std::vector<DMat> mats(desc->n_materials);
    for (uint32_t i = 0; i < desc->n_materials; ++i) {
        const pt_material &m = desc->materials[i];
        mats[i] = DMat{m.kind, m.albedo[0], m.albedo[1], m.albedo[2], m.param, m.texture, 0.f, 0.f};
    }
    std::vector<DTex> texs(desc->n_textures ? desc->n_textures : 1);
    for (uint32_t i = 0; i < desc->n_textures; ++i) {
        const pt_texture &t = desc->textures[i];
        texs[i] = DTex{t.kind, t.color[0], t.color[1], t.color[2], t.odd, t.even, t.scale, 0.f};
    }
    // BVH mode: per-sphere parent AABB + DFS rank from the CALLER's tree (they define the result), and the
    // internal traversal tree.
    std::vector<uint32_t> leaf_rank(desc->n_spheres, 0), bvh_large, rank_sphere;
    std::vector<float4> gate(2 * (size_t)desc->n_spheres, make_float4(0, 0, 0, 0)), gate_chain;
    if (desc->n_bvh_nodes) {
        union FU { uint32_t u; float f; };
        // a sphere that is not a leaf of the caller's tree can never be hit: chain count 0xffffffff = "never"
        const FU never{0xffffffffu};
        for (uint32_t i = 0; i < desc->n_spheres; ++i) gate[2 * i] = make_float4(0, 0, 0, never.f);
        // The slab test (aabb.rs:46-58) takes min/max of the two plane distances, so a box acts as the interval
        // [min(mn, mx), max(mn, mx)] per axis; when an ancestor's interval contains its child's on every axis,
        // passing the child implies passing the ancestor (the arithmetic is monotone). Boxes built by
        // AABB::add (aabb.rs:61-66) nest like that, EXCEPT above inverted boxes (a negative radius gives
        // min > max, sphere.rs:69-75): there an ancestor can reject a ray its descendant accepts. Each leaf
        // therefore gets its parent's box plus every ancestor that is not implied by the one below it.
        auto implied_by = [&](const pt_bvh_node &up, const pt_bvh_node &low) {
            for (int a = 0; a < 3; ++a) {
                const float ul = std::min(up.min[a], up.max[a]), uh = std::max(up.min[a], up.max[a]);
                const float ll = std::min(low.min[a], low.max[a]), lh = std::max(low.min[a], low.max[a]);
                if (!(ul <= ll && uh >= lh)) return false;
            }
            return true;
        };
        // lhs-before-rhs DFS; a sphere referenced by several leaves keeps its LAST rank (bvh.rs:73-79 lhs == rhs)
        struct Item { int32_t ref; int32_t parent; uint32_t depth; };
        std::vector<Item> st{{desc->bvh_root, -1, 0}};
        std::vector<int32_t> path;   // ancestors of the item being visited, root first
        uint32_t rank = 0;
        while (!st.empty()) {
            const Item it = st.back();
            st.pop_back();
            path.resize(it.depth);
            if (it.ref < 0) {
                const uint32_t k = (uint32_t)~it.ref;
                rank_sphere.push_back(k);
                leaf_rank[k] = rank++;
                const pt_bvh_node &pn = desc->bvh_nodes[it.parent];
                FU cnt{0}, off{(uint32_t)(gate_chain.size() / 2)};
                for (size_t j = path.size() - 1; j-- > 0;) {   // grandparent upwards
                    const pt_bvh_node &up = desc->bvh_nodes[path[j]], &low = desc->bvh_nodes[path[j + 1]];
                    if (!implied_by(up, low)) {
                        gate_chain.push_back(make_float4(up.min[0], up.min[1], up.min[2], 0.f));
                        gate_chain.push_back(make_float4(up.max[0], up.max[1], up.max[2], 0.f));
                        ++cnt.u;
                    }
                }
                gate[2 * k] = make_float4(pn.min[0], pn.min[1], pn.min[2], cnt.f);
                gate[2 * k + 1] = make_float4(pn.max[0], pn.max[1], pn.max[2], off.f);
            } else {
                path.push_back(it.ref);
                st.push_back({desc->bvh_nodes[it.ref].rhs, it.ref, it.depth + 1});
                st.push_back({desc->bvh_nodes[it.ref].lhs, it.ref, it.depth + 1});
            }
        }
    }
    DNode4 *d_nodes4_built = nullptr;
    {   // the internal tree is built for every scene: BVH mode always uses it, list mode uses it for scenes too
        // large for the brute-force scan (there it needs no gate: closest t, ties to the lower list index)
        const std::vector<AccelItem> items = accel_items(desc, motion, t_lo, t_hi, bvh_large);
        s->n_bvh_large = (uint32_t)bvh_large.size();
        s->has_tree_items = !items.empty();
        s->h_spheres.assign(desc->spheres, desc->spheres + desc->n_spheres);
        if (motion) s->h_motion.assign(motion, motion + desc->n_spheres);
        s->h_t_lo = t_lo, s->h_t_hi = t_hi;
        if (!items.empty()) {
            std::vector<TreeItem> titems(items.size());
            for (size_t i = 0; i < items.size(); ++i) {
                titems[i].sphere = items[i].sphere, titems[i].r = items[i].r;
                memcpy(titems[i].c, items[i].c, 12), memcpy(titems[i].mn, items[i].mn, 12), memcpy(titems[i].mx, items[i].mx, 12);
            }
            // 4-wide tree: built ON THE DEVICE (pt_build.hip); PTGPU_HOST_BUILD=1 runs the host restatement instead (tests)
            int brc = getenv("PTGPU_HOST_BUILD") ? -1 : tree4_build_device(titems.data(), (uint32_t)titems.size(), nullptr, &d_nodes4_built, &s->n_nodes4, &s->depth4,
                                                                         &s->tree_build_ms);
            if (brc == 0) {
                s->tree_on_device = true;
            } else {
                const Tree4Host t4 = tree4_build_host(titems);
                s->n_nodes4 = (uint32_t)t4.nodes.size(), s->depth4 = t4.depth;
                if (upload(&d_nodes4_built, t4.nodes.data(), t4.nodes.size()) != PT_OK) {
                    pt_scene_destroy(s);
                    return fail(PT_ERR_HIP, "uploading the internal tree failed");
                }
            }
        }
        s->d_nodes4 = d_nodes4_built;
    }
    std::vector<float4> pvec(256, make_float4(0, 0, 0, 0));
    std::vector<uint32_t> pperm(768, 0);
    if (desc->perlin) {
        for (int i = 0; i < 256; ++i) {
            pvec[i] = make_float4(desc->perlin->randvec[i][0], desc->perlin->randvec[i][1], desc->perlin->randvec[i][2], 0.f);
            pperm[i] = desc->perlin->perm_x[i];
            pperm[256 + i] = desc->perlin->perm_y[i];
            pperm[512 + i] = desc->perlin->perm_z[i];
        }
    }
    std::vector<float4> leafrec;
    if (desc->n_bvh_nodes) {
        leafrec.resize(4 * (size_t)desc->n_spheres);
        for (uint32_t i = 0; i < desc->n_spheres; ++i) {
            union { uint32_t u; float f; } rk{leaf_rank[i]};
            leafrec[4 * i] = sph[i], leafrec[4 * i + 1] = gate[2 * i], leafrec[4 * i + 2] = gate[2 * i + 1];
            leafrec[4 * i + 3] = make_float4(rk.f, 0.f, 0.f, 0.f);
        }
    }
    std::vector<float4> shade_rank(4 * rank_sphere.size());
    for (size_t r = 0; r < rank_sphere.size(); ++r)
        for (int q = 0; q < 4; ++q) shade_rank[4 * r + q] = shade[4 * (size_t)rank_sphere[r] + q];
    int rc = PT_OK;
    if ((rc = upload(&s->d_shade_rank, shade_rank.data(), shade_rank.size())) || (rc = upload(&s->d_leafrec, leafrec.data(), leafrec.size())) || (rc = upload(&s->d_spheres, sph.data(), sph.size())) || (rc = upload(&s->d_spheres_r2, sph_r2.data(), sph_r2.size())) || (rc = upload(&s->d_shade, shade.data(), shade.size())) ||
        (rc = upload(&s->d_sphere_mat, desc->sphere_material, desc->n_spheres)) ||
        (rc = upload(&s->d_mats, mats.data(), mats.size())) || (rc = upload(&s->d_texs, texs.data(), texs.size())) ||
        (rc = upload(&s->d_perlin_vec, pvec.data(), pvec.size())) || (rc = upload(&s->d_perlin_perm, pperm.data(), pperm.size())) ||
        (rc = upload(&s->d_leaf_rank, leaf_rank.data(), leaf_rank.size())) ||
        (rc = upload(&s->d_gate, gate.data(), gate.size())) || (rc = upload(&s->d_gate_chain, gate_chain.data(), gate_chain.size())) ||
        (rc = upload(&s->d_bvh_large, bvh_large.data(), bvh_large.size())) || (rc = upload(&s->d_rank_sphere, rank_sphere.data(), rank_sphere.size()))) {
        pt_scene_destroy(s);
        return rc;
    }
    if (s->d_nodes4) {   // the nodes the kernels read + the leaves' slot records, from what is on the device now
        const int prc = tree4_pack_device(s->d_nodes4, s->n_nodes4, s->d_spheres, desc->n_bvh_nodes ? s->d_leafrec : nullptr, nullptr, &s->d_nodes4q,
                                          &s->d_slotrec, &s->tree4_packed);
        if (prc != 0) {
            pt_scene_destroy(s);
            return fail(PT_ERR_HIP, "packing the internal tree failed (hipError %d)", prc);
        }
    }
    if (motion) {
        std::vector<float4> mot(2 * (size_t)desc->n_spheres, make_float4(0, 0, 0, 0));
        for (uint32_t i = 0; i < desc->n_spheres; ++i) {
            if (!motion[i].moving) continue;
            mot[2 * i] = make_float4(motion[i].delta[0], motion[i].delta[1], motion[i].delta[2], motion[i].inv_time_delta);
            mot[2 * i + 1] = make_float4(motion[i].time_start, 1.0f, 0.f, 0.f);
        }
        if ((rc = upload(&s->d_motion, mot.data(), mot.size()))) {
            pt_scene_destroy(s);
            return rc;
        }
    }
    {
        MfmaPrep prep;
        if (prepare_mfma(desc, motion, t_lo, t_hi, prep)) {
            if ((rc = upload(&s->d_afrag, prep.afrag.data(), prep.afrag.size() / 8)) ||
                (rc = upload(&s->d_tile_sphere, prep.tile_sphere.data(), prep.tile_sphere.size())) ||
                (rc = upload(&s->d_large, prep.large.data(), prep.large.size()))) {
                pt_scene_destroy(s);
                return rc;
            }
            if (prep.cull_axis < 3u) {
                if ((rc = upload(&s->d_cull_tab, prep.cull_tab.data(), prep.cull_tab.size()))) {
                    pt_scene_destroy(s);
                    return rc;
                }
                s->cull_axis = prep.cull_axis, s->cull_always = prep.cull_always;
                s->cull_u0 = prep.cull_u0, s->cull_inv_cell = prep.cull_inv_cell;
                s->cull_rmin = prep.cull_rmin, s->cull_rmax = prep.cull_rmax, s->cull_cell = prep.cull_cell, s->rs_small = (float)prep.rs;
                memcpy(s->clip_min, prep.clip_min, 12), memcpy(s->clip_max, prep.clip_max, 12);
            }
            s->n_tiles = prep.n_tiles;
            s->n_large = (uint32_t)prep.large.size();
            memcpy(s->c0, prep.c0, sizeof s->c0);
            s->rs2 = (float)(prep.rs * prep.rs * 1.0001);
            // margin = a * (m0 + gamma * (|o - c0|^2 + Rs^2)); see DESIGN.md for the derivation
            // a swept bound of half-length h moves the reference's rounding slack from radius r to r + h: scale by (1 + h/r)
            const double widen = 1.0 + 1.5 * prep.sweep_ratio;
            s->m0 = (float)((1.0e-5 * prep.rs * prep.rs + 1.0e-4) * widen);
            s->gamma = (float)(8.0e-6 * widen);
        }
    }
    if (hipMalloc((void **)&s->d_debug, 1024) != hipSuccess || hipMemset(s->d_debug, 0, 1024) != hipSuccess) {
        pt_scene_destroy(s);
        return fail(PT_ERR_HIP, "allocating debug counters failed");
    }
    if (hipMalloc((void **)&s->d_work_counter, 64) != hipSuccess || hipMalloc((void **)&s->d_ray_count, 64) != hipSuccess ||
        hipEventCreate(&s->ev_start) != hipSuccess || hipEventCreate(&s->ev_stop) != hipSuccess || hipEventCreate(&s->ev_pass) != hipSuccess) {
        pt_scene_destroy(s);
        return fail(PT_ERR_HIP, "allocating work counters / events failed");
    }
    if (const char *e = getenv("PTGPU_BLOCKS_PER_CU")) s->blocks_per_cu = (uint32_t)atoi(e);
    if (const char *e = getenv("PTGPU_VARIANT")) s->variant = (uint32_t)atoi(e);
    if (getenv("PTGPU_TIMING") != nullptr && hipMalloc((void **)&s->d_wave_end, 65536 * 8) == hipSuccess) s->timing = true;
    *scene_out = s;
    return PT_OK;
}
}  // namespace

// ---- general worlds --------------------------------------------------------------------------------
extern "C" int pt_scene_create_world(const pt_world_desc *desc, int device, pt_scene **scene_out) {
    if (!desc || !scene_out) return fail(PT_ERR_INVALID_ARG, "desc/scene_out is NULL");
    *scene_out = nullptr;
    // an EMPTY list is a valid world (HitableList::ray_hit returns None for every ray: the `final` preset)
    if (desc->n_hitables && !desc->hitables) return fail(PT_ERR_INVALID_ARG, "hitables is NULL");
    if (desc->n_hitables > 0x3fffffffu) return fail(PT_ERR_INVALID_ARG, "too many hitables");
    if (desc->n_hitables && (desc->n_materials == 0 || !desc->materials)) return fail(PT_ERR_INVALID_ARG, "world has no materials");
    if (desc->n_materials && !desc->materials) return fail(PT_ERR_INVALID_ARG, "materials is NULL");
    if (desc->n_hitables == 0 && desc->n_bvh_nodes) return fail(PT_ERR_INVALID_ARG, "BVH nodes over an empty list");
    if (desc->n_textures && !desc->textures) return fail(PT_ERR_INVALID_ARG, "textures is NULL");
    if (desc->n_transforms && !desc->transforms) return fail(PT_ERR_INVALID_ARG, "transforms is NULL");
    bool has_noise = false;
    if (int rc = validate_tables(desc->n_materials, desc->materials, desc->n_textures, desc->textures, desc->perlin, true, &has_noise,
                                 desc->n_images, desc->images)) return rc;
    bool has_image = false;
    for (uint32_t i = 0; i < desc->n_textures; ++i) has_image = has_image || desc->textures[i].kind == PT_TEX_IMAGE;
    bool all_spheres = true, sphere_like = true, has_media = false;
    for (uint32_t i = 0; i < desc->n_hitables; ++i) {
        const pt_hitable &h = desc->hitables[i];
        if (h.kind > PT_HIT_CUBOID) return fail(PT_ERR_INVALID_ARG, "hitable %u: unknown kind %u", i, h.kind);
        if (h.material >= desc->n_materials) return fail(PT_ERR_INVALID_ARG, "hitable %u: material index out of range", i);
        if (desc->materials[h.material].kind == PT_MAT_ISOTROPIC)
            return fail(PT_ERR_INVALID_ARG, "hitable %u: Isotropic is only valid as a medium's phase function", i);
        if (h.transform >= 0 && (uint32_t)h.transform >= desc->n_transforms)
            return fail(PT_ERR_INVALID_ARG, "hitable %u: transform index out of range", i);
        if (h.medium_material >= 0) {
            if ((uint32_t)h.medium_material >= desc->n_materials || desc->materials[h.medium_material].kind != PT_MAT_ISOTROPIC)
                return fail(PT_ERR_INVALID_ARG, "hitable %u: medium_material must index an Isotropic material", i);
        }
        has_media = has_media || h.medium_material >= 0;
        if (h.kind != PT_HIT_SPHERE || h.transform >= 0 || h.medium_material >= 0) all_spheres = false;
        if (h.kind > PT_HIT_MOVING_SPHERE || h.transform >= 0 || h.medium_material >= 0) sphere_like = false;
    }
    if (desc->n_bvh_nodes) {
        if (!desc->bvh_nodes) return fail(PT_ERR_INVALID_ARG, "bvh_nodes is NULL");
        if (bvh_depth_checked(desc->bvh_nodes, desc->n_bvh_nodes, desc->n_hitables, desc->bvh_root) == 0)
            return fail(PT_ERR_INVALID_ARG, "malformed BVH (bad child index or cycle)");
    }
    uint32_t ref_depth = 0;
    if (desc->n_bvh_nodes) ref_depth = bvh_depth_checked(desc->bvh_nodes, desc->n_bvh_nodes, desc->n_hitables, desc->bvh_root);
    if (sphere_like && desc->n_hitables) {
        // Sphere / MovingSphere entries only: the specialised kernels apply (MFMA prefilter, internal tree); with
        // moving entries their MOVING instantiations, and the general-world data rides along as the fallback
        std::vector<pt_sphere> sph(desc->n_hitables);
        std::vector<uint32_t> mat(desc->n_hitables);
        std::vector<MotionIn> motion(desc->n_hitables);
        for (uint32_t i = 0; i < desc->n_hitables; ++i) {
            const pt_hitable &h = desc->hitables[i];
            mat[i] = h.material;
            if (h.kind == PT_HIT_SPHERE) {
                sph[i] = pt_sphere{h.p[0], h.p[1], h.p[2], h.p[3]};
                motion[i] = MotionIn{{0, 0, 0}, 0.f, 0.f, 0u};
            } else {
                sph[i] = pt_sphere{h.p[0], h.p[1], h.p[2], h.p[6]};
                motion[i] = MotionIn{{h.p[3], h.p[4], h.p[5]}, h.p[7], h.p[8], 1u};
            }
        }
        // Sphere hits have u = v = 0 (sphere.rs:47-48), so an Image texture is one texel for them: i = 0,
        // j = ((1 - 0) * height - 0.001) as i32 = height - 1 (texture.rs:28-33). Fold it into a Constant.
        std::vector<pt_texture> folded(desc->textures, desc->textures + desc->n_textures);
        for (pt_texture &t : folded) {
            if (t.kind != PT_TEX_IMAGE) continue;
            const pt_image &im = desc->images[t.odd];
            const volatile float fj = (1.0f - 0.0f) * (float)im.height - 0.001f;
            int64_t j = (int64_t)fj;
            j = std::max<int64_t>(0, std::min<int64_t>(j, (int64_t)im.height - 1));
            const uint8_t *px = im.rgb + 3ull * im.width * (uint64_t)j;
            const volatile float k255 = 255.0f;
            t.kind = PT_TEX_CONSTANT;
            t.color[0] = (float)px[0] / k255, t.color[1] = (float)px[1] / k255, t.color[2] = (float)px[2] / k255;
            t.odd = t.even = -1;
        }
        pt_scene_desc d{};
        d.n_spheres = desc->n_hitables, d.spheres = sph.data(), d.sphere_material = mat.data();
        d.n_materials = desc->n_materials, d.materials = desc->materials;
        d.n_textures = desc->n_textures, d.textures = folded.data(), d.perlin = desc->perlin;
        d.n_bvh_nodes = desc->n_bvh_nodes, d.bvh_nodes = desc->bvh_nodes, d.bvh_root = desc->bvh_root;
        d.has_sky = desc->has_sky;
        memcpy(d.sky, desc->sky, sizeof d.sky);
        if (all_spheres) return create_sphere_scene(&d, nullptr, device, scene_out);
        int rc = create_sphere_scene(&d, motion.data(), device, scene_out);
        if (rc == PT_OK) {
            pt_scene *s = *scene_out;
            s->n_hitables = desc->n_hitables;
            s->ref_bvh_depth = ref_depth;
            if (ref_depth + 2 > 64u || (rc = upload(&s->d_hitables, desc->hitables, desc->n_hitables)) ||
                (rc = upload(&s->d_transforms, desc->transforms, desc->n_transforms)) ||
                (rc = upload(&s->d_ref_nodes, desc->bvh_nodes, desc->n_bvh_nodes))) {
                pt_scene_destroy(s);
                *scene_out = nullptr;
                return rc ? rc : fail(PT_ERR_UNSUPPORTED, "BVH depth %u exceeds the traversal stack", ref_depth);
            }
            return PT_OK;
        }
        if (rc != PT_ERR_UNSUPPORTED) return rc;
        // (unsupported by the specialised path, e.g. a degenerate time interval: trace it as a general world)
    }
    if (ref_depth + 2 > 64u) return fail(PT_ERR_UNSUPPORTED, "BVH depth %u exceeds the traversal stack", ref_depth);

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PT_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(PT_ERR_INVALID_ARG, "device %d out of range (%d devices)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    pt_scene *s = new (std::nothrow) pt_scene();
    if (!s) return fail(PT_ERR_INVALID_ARG, "out of host memory");
    s->device = device;
    s->num_cus = prop.multiProcessorCount;
    s->is_world = true;
    s->has_media = has_media;
    s->n_hitables = desc->n_hitables;
    s->n_world_xf = desc->n_transforms;
    s->n_materials = desc->n_materials;
    s->n_textures = desc->n_textures;
    s->bvh_root = desc->n_bvh_nodes ? desc->bvh_root : -1;
    s->ref_bvh_depth = ref_depth;
    s->has_sky = desc->has_sky ? 1u : 0u;
    memcpy(s->sky, desc->sky, sizeof s->sky);
    s->has_noise = has_noise ? 1u : 0u;
    std::vector<DMat> mats(desc->n_materials);
    for (uint32_t i = 0; i < desc->n_materials; ++i) {
        const pt_material &m = desc->materials[i];
        mats[i] = DMat{m.kind, m.albedo[0], m.albedo[1], m.albedo[2], m.param, m.texture, 0.f, 0.f};
        // a Constant texture is resolved here: the general kernel then needs no texture call for it (pad0 = 1)
        if ((m.kind == PT_MAT_LAMBERTIAN || m.kind == PT_MAT_DIFFUSE_LIGHT || m.kind == PT_MAT_ISOTROPIC) &&
            desc->textures[m.texture].kind == PT_TEX_CONSTANT) {
            const pt_texture &t = desc->textures[m.texture];
            mats[i].a0 = t.color[0], mats[i].a1 = t.color[1], mats[i].a2 = t.color[2], mats[i].pad0 = 1.0f;
        }
    }
    std::vector<DTex> texs(desc->n_textures ? desc->n_textures : 1);
    for (uint32_t i = 0; i < desc->n_textures; ++i) {
        const pt_texture &t = desc->textures[i];
        texs[i] = DTex{t.kind, t.color[0], t.color[1], t.color[2], t.odd, t.even, t.scale, 0.f};
    }
    std::vector<float4> pvec(256, make_float4(0, 0, 0, 0));
    std::vector<uint32_t> pperm(768, 0);
    if (desc->perlin) {
        for (int i = 0; i < 256; ++i) {
            pvec[i] = make_float4(desc->perlin->randvec[i][0], desc->perlin->randvec[i][1], desc->perlin->randvec[i][2], 0.f);
            pperm[i] = desc->perlin->perm_x[i];
            pperm[256 + i] = desc->perlin->perm_y[i];
            pperm[512 + i] = desc->perlin->perm_z[i];
        }
    }
    int rc = PT_OK;
    if (has_image) {
        std::vector<uint4> table(desc->n_images);
        std::vector<uint8_t> blob;
        for (uint32_t i = 0; i < desc->n_images; ++i) {
            const pt_image &im = desc->images[i];
            table[i] = make_uint4((uint32_t)blob.size(), im.width, im.height, 0u);
            blob.insert(blob.end(), im.rgb, im.rgb + 3ull * im.width * im.height);
        }
        if (blob.size() > 0xf0000000ull) {
            pt_scene_destroy(s);
            return fail(PT_ERR_UNSUPPORTED, "image textures exceed 3.75 GB");
        }
        if ((rc = upload(&s->d_image_table, table.data(), table.size())) || (rc = upload(&s->d_image_bytes, blob.data(), blob.size()))) {
            pt_scene_destroy(s);
            return rc;
        }
        s->has_image = 1u;
    }
    if ((rc = upload(&s->d_hitables, desc->hitables, desc->n_hitables)) ||
        (rc = upload(&s->d_transforms, desc->transforms, desc->n_transforms)) ||
        (rc = upload(&s->d_ref_nodes, desc->bvh_nodes, desc->n_bvh_nodes)) ||
        (rc = upload(&s->d_mats, mats.data(), mats.size())) || (rc = upload(&s->d_texs, texs.data(), texs.size())) ||
        (rc = upload(&s->d_perlin_vec, pvec.data(), pvec.size())) || (rc = upload(&s->d_perlin_perm, pperm.data(), pperm.size()))) {
        pt_scene_destroy(s);
        return rc;
    }
    if (hipMalloc((void **)&s->d_debug, 1024) != hipSuccess || hipMemset(s->d_debug, 0, 1024) != hipSuccess ||
        hipMalloc((void **)&s->d_work_counter, 64) != hipSuccess || hipMalloc((void **)&s->d_ray_count, 64) != hipSuccess ||
        hipEventCreate(&s->ev_start) != hipSuccess || hipEventCreate(&s->ev_stop) != hipSuccess || hipEventCreate(&s->ev_pass) != hipSuccess) {
        pt_scene_destroy(s);
        return fail(PT_ERR_HIP, "allocating counters / events failed");
    }
    if (const char *e = getenv("PTGPU_BLOCKS_PER_CU")) s->blocks_per_cu = (uint32_t)atoi(e);
    *scene_out = s;
    return PT_OK;
}

extern "C" void pt_scene_destroy(pt_scene *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    (void)hipFree(s->d_hitables);
    (void)hipFree(s->d_motion);
    (void)hipFree(s->d_transforms);
    (void)hipFree(s->d_ref_nodes);
    (void)hipFree(s->d_image_table);
    (void)hipFree(s->d_image_bytes);
    (void)hipFree(s->d_spheres);
    (void)hipFree(s->d_spheres_r2);
    (void)hipFree(s->d_shade);
    (void)hipFree(s->d_sphere_mat);
    (void)hipFree(s->d_mats);
    (void)hipFree(s->d_texs);
    (void)hipFree(s->d_perlin_vec);
    (void)hipFree(s->d_perlin_perm);
    (void)hipFree(s->d_gate);
    (void)hipFree(s->d_gate_chain);
    (void)hipFree(s->d_bvh_large);
    (void)hipFree(s->d_wnodes);
    (void)hipFree(s->d_nodes4);
    (void)hipFree(s->d_nodes4q);
    (void)hipFree(s->d_slotrec);
    (void)hipFree(s->d_rank_sphere);
    (void)hipFree(s->d_leafrec);
    (void)hipFree(s->d_shade_rank);
    (void)hipFree(s->d_leaf_rank);
    (void)hipFree(s->d_afrag);
    (void)hipFree(s->d_tile_sphere);
    (void)hipFree(s->d_cull_tab);
    (void)hipFree(s->d_large);
    (void)hipFree(s->d_debug);
    (void)hipFree(s->d_tile_buf);
    (void)hipFree(s->d_px_state);
    (void)hipFree(s->d_work_counter);
    (void)hipFree(s->d_ray_count);
    (void)hipFree(s->d_frame);
    (void)hipHostFree(s->h_stage);
    (void)hipFree(s->d_gstack);
    (void)hipFree(s->d_wave_end);
    if (s->ev_start) (void)hipEventDestroy(s->ev_start);
    if (s->ev_stop) (void)hipEventDestroy(s->ev_stop);
    if (s->ev_pass) (void)hipEventDestroy(s->ev_pass);
    delete s;
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

namespace {

f3 to3(const float *p) { return f3{p[0], p[1], p[2]}; }

// hipFuncAttributeMaxDynamicSharedMemorySize is sticky per kernel: set it when the kernel or its LDS size changes, not
// on every frame
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

int set_lds_limit(pt_scene *s, int slot, const void *kern, uint32_t lds) {
    if (s->attr_kern[slot] == kern && s->attr_lds[slot] == lds) return PT_OK;
    HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    s->attr_kern[slot] = kern, s->attr_lds[slot] = lds;
    return PT_OK;
}

// The per-frame arguments both kernels share (KArgs and WArgs use the same member names).
template <typename Args>
void fill_frame_args(Args &X, const pt_scene *s, const pt_params *params, const pt_camera *cam, uint32_t frame_num,
                     uint32_t shard_index, uint32_t shard_count, float *d_rgb, uint64_t *d_ray_count) {
    X.has_sky = s->has_sky;
    X.sky = to3(s->sky);
    X.has_noise = s->has_noise;
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
    // refills are batched: measured best at 4 waiting lanes for long pixels, 8 when pixels are short (< 32 spp)
    X.refill_min = params->samples < 32u ? 8u : 4u;
    if (dev_knobs().refill >= 0) X.refill_min = (uint32_t)dev_knobs().refill;   // (development knob PTGPU_REFILL)
    X.seed_base = s->seed_base;
    X.shard_index = shard_index;
    X.shard_count = shard_count;
    X.local_rows = pt_shard_rows(params->height, shard_index, shard_count);
    X.tiles_x = (params->width + kTileSide - 1u) / kTileSide;
    X.tiles_x_magic = X.tiles_x > 1u ? (uint32_t)(0x100000000ull / X.tiles_x) : 0xffffffffu;   // (tiles_x == 1: umulhi gives tile - 1 for tile > 0, corrected by the kernel's one step)
    X.n_items = X.tiles_x * ((X.local_rows + kTileSide - 1u) / kTileSide) * kTilePix;
    X.rgb = d_rgb;
    X.ray_count = reinterpret_cast<unsigned long long *>(d_ray_count);
    X.work_counter = s->d_work_counter;
}

// The binary internal tree (variant bit 2048 for A/B runs, and scenes whose 4-wide tree would exceed 65535 nodes) is
// built on the host the first time a launch needs it.
int ensure_binary_tree(pt_scene *s) {
    if (s->binary_built) return PT_OK;
    pt_scene_desc d{};
    d.n_spheres = (uint32_t)s->h_spheres.size();
    d.spheres = s->h_spheres.data();
    AccelBuild acc = build_accel(&d, s->h_motion.empty() ? nullptr : s->h_motion.data(), s->h_t_lo, s->h_t_hi);
    if (acc.depth + 2 > (uint32_t)kBvhStack) return fail(PT_ERR_UNSUPPORTED, "internal BVH depth %u exceeds the traversal stack", acc.depth);
    if (int rc = upload(&s->d_wnodes, acc.nodes.data(), acc.nodes.size())) return rc;
    s->bin_root = acc.root;
    s->bvh_depth = acc.depth;
    s->n_nodes = (uint32_t)acc.nodes.size();
    s->binary_built = true;
    return PT_OK;
}

int launch(pt_scene *s, const pt_params *params, const pt_camera *cam, uint32_t frame_num, uint32_t shard_index,
           uint32_t shard_count, float *d_rgb, uint64_t *d_ray_count, hipStream_t stream) {
    if (!s || !params || !cam || !d_rgb || !d_ray_count) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    if (params->width == 0 || params->height == 0 || params->samples == 0)
        return fail(PT_ERR_INVALID_ARG, "width/height/samples must be non-zero");
    if ((uint64_t)params->width * params->height > 0x3fffffffull) return fail(PT_ERR_INVALID_ARG, "frame too large");
    // every kernel packs a lane's pixel into one register; the sphere kernels its (depth, sample) counters as well
    if (params->width > 0xffffu || params->height > 0xffffu) return fail(PT_ERR_UNSUPPORTED, "width and height must be below 65536");
    if (!s->is_world && (params->max_depth > 0xfffu || params->samples > 0xfffffu))
        return fail(PT_ERR_UNSUPPORTED, "sphere kernels take max_depth < 4096, samples < 2^20");
    if (shard_count == 0 || shard_index >= shard_count) return fail(PT_ERR_INVALID_ARG, "bad shard %u/%u", shard_index, shard_count);
    const bool ref_bvh = params->use_bvh != 0;   // BVHNode::ray_hit semantics (needs the caller's tree for the gates)
    if (ref_bvh && s->bvh_root < 0) return fail(PT_ERR_UNSUPPORTED, "use_bvh requested but the scene was created without BVH nodes");
    // list mode walks the internal tree instead of scanning when the scan would be the slower option:
    // more than kListTreeMin spheres, or a scene the MFMA prefilter cannot take (variant bit 64 forces the scan)
    constexpr uint32_t kListTreeMin = 768;   // = 24 MFMA tiles: beyond that the fragments no longer leave room for 2 workgroups per CU
    const bool list_tree = !ref_bvh && (s->variant & (4u | 64u)) == 0 && (s->n_spheres > kListTreeMin || s->n_spheres > 0xfff0u);
    // A BVH WORLD is a list world plus two rules applied when a hit is accepted (ancestor-AABB gate, DFS-rank ties),
    // so scenes the MFMA prefilter can take run on it in BVH mode as well (the tree kernel: variant bit 256, or 4)
    const uint32_t n_pad0 = (s->n_spheres + kScanUnroll - 1) / kScanUnroll * kScanUnroll;
    const bool mfma_fits = !s->is_world && (s->variant & (1u | 4u)) == 0 && n_pad0 * 16u <= 64u * 1024u && s->n_tiles > 0 && s->n_tiles <= 24u;
    const bool bvh = (ref_bvh && !(mfma_fits && (s->variant & 256u) == 0)) || list_tree;   // kernel flavour: tree traversal
    HIP_TRY(hipSetDevice(s->device));

    // Sphere + MovingSphere worlds: the MOVING instantiations exist for the MFMA list kernel and the tree kernel,
    // and their swept bounds cover ray times in [time_lo, time_hi] only. Anything else (exact-scan variants, a
    // camera shutter outside that interval, variant bit 128) is traced by the general kernel.
    bool moving = false;
    if (s->has_motion) {
        const float lo = std::min(cam->time0, cam->time1), hi = std::max(cam->time0, cam->time1);
        const bool time_ok = std::isfinite(lo) && std::isfinite(hi) && lo >= s->time_lo && hi <= s->time_hi;
        moving = time_ok && (bvh || mfma_fits) && (s->variant & 128u) == 0;
    }

    if (s->is_world || (s->has_motion && !moving)) {
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
        W.has_image = s->has_image;
        W.n_hit = s->n_hitables;
        W.bvh_root = ref_bvh ? s->bvh_root : -1;
        W.bvh_stack_entries = s->ref_bvh_depth + 2u;
        fill_frame_args(W, s, params, cam, frame_num, shard_index, shard_count, d_rgb, d_ray_count);
        HIP_TRY(hipMemsetAsync(s->d_work_counter, 0, sizeof(uint32_t), stream));
        HIP_TRY(hipMemsetAsync(d_ray_count, 0, sizeof(uint64_t), stream));
        if (W.n_items == 0) return PT_OK;
        uint32_t lds = s->has_noise ? (4096u + 768u) : 0u;
        if (ref_bvh) lds += W.bvh_stack_entries * kBlock * 4u;
        W.n_xf = s->n_world_xf;
        const bool hit_lds = s->n_hitables * 64u + s->n_world_xf * 96u <= 40960u;   // records + transforms staged in LDS
        if (hit_lds) lds += s->n_hitables * 64u + s->n_world_xf * 96u;
        lds += 8u * kBlock * 4u;                                                    // the running closest-hit record
        const uint64_t path_bytes = (uint64_t)params->max_depth * 3ull * kBlock * 4ull;
        W.stack_in_lds = (lds + path_bytes <= 60u * 1024u) ? 1u : 0u;
        if (W.stack_in_lds) lds += (uint32_t)path_bytes;
        const bool occ4 = !s->has_noise && !s->has_image && s->blocks_per_cu == 0 && 4u * lds <= kLdsBudget && !dev_knobs().world_occ3;   // see pt_world_kernel's OCC
        // one instantiation per (traversal, records in LDS, waves per SIMD, world has media); worlds whose records do not
        // fit LDS (more than ~600 hitables) share the MEDIA = true code
        const bool media = s->has_media || !hit_lds || s->has_motion;
        void (*wk)(const WArgs) = nullptr;
        if (occ4)
            wk = ref_bvh ? (hit_lds ? (media ? pt_world_kernel<true, true, 4, true> : pt_world_kernel<true, true, 4, false>) : pt_world_kernel<true, false, 4, true>)
                         : (hit_lds ? (media ? pt_world_kernel<false, true, 4, true> : pt_world_kernel<false, true, 4, false>) : pt_world_kernel<false, false, 4, true>);
        else
            wk = ref_bvh ? (hit_lds ? (media ? pt_world_kernel<true, true, 3, true> : pt_world_kernel<true, true, 3, false>) : pt_world_kernel<true, false, 3, true>)
                         : (hit_lds ? (media ? pt_world_kernel<false, true, 3, true> : pt_world_kernel<false, true, 3, false>) : pt_world_kernel<false, false, 3, true>);
        uint32_t bpc = s->blocks_per_cu ? s->blocks_per_cu : std::min(occ4 ? 4u : 3u, blocks_per_cu_by_registers(reinterpret_cast<const void *>(wk), kBlock));
        const uint32_t lds_limit = lds ? (kLdsBudget / lds) : 8u;
        if (bpc > lds_limit) bpc = lds_limit ? lds_limit : 1u;
        if (bpc > 8u) bpc = 8u;
        uint32_t grid = (uint32_t)s->num_cus * bpc;
        const uint32_t need = (W.n_items + kBlock - 1) / kBlock;
        if (grid > need) grid = need;
        if (!W.stack_in_lds) {
            const size_t need_floats = (size_t)grid * params->max_depth * 3ull * kBlock;
            if (need_floats > s->d_gstack_floats) {
                (void)hipFree(s->d_gstack);
                s->d_gstack = nullptr;
                s->d_gstack_floats = 0;
                HIP_TRY(hipMalloc((void **)&s->d_gstack, need_floats * sizeof(float)));
                s->d_gstack_floats = need_floats;
            }
            W.gstack = s->d_gstack;
        }
        if (int rc = set_lds_limit(s, 0, reinterpret_cast<const void *>(wk), lds)) return rc;
        HIP_TRY(hipEventRecord(s->ev_pass, stream));
        // heavy-first work order, as for the sphere kernels below: a repeated view is ordered by the
        // rays its last frame measured per tile; a new view runs as two launches of this kernel, the first tracing the first
        // sample of every pixel while it counts
        const uint32_t n_world_tiles = W.n_items / kTilePix;
        if (n_world_tiles >= 256u && params->samples >= kPilotMinSamples && (s->variant & (32u | 16384u)) == 0) {
            if (n_world_tiles > s->d_tile_cap) {
                (void)hipFree(s->d_tile_buf);
                s->d_tile_buf = nullptr, s->d_tile_cap = 0, s->hint_valid = false;
                HIP_TRY(hipMalloc((void **)&s->d_tile_buf, (8 + 3 * (size_t)n_world_tiles) * sizeof(uint32_t)));
                s->d_tile_cap = n_world_tiles;
            }
            uint32_t *cost = s->d_tile_buf + 8, *order = cost + s->d_tile_cap, *measured = order + s->d_tile_cap;
            pt_scene::ViewKey key{};
            key.params = *params, key.cam = *cam, key.shard_index = shard_index, key.shard_count = shard_count, key.variant = s->variant, key.n_tiles = n_world_tiles;
            const bool reuse = s->hint_valid && (s->variant & 8192u) == 0 && memcmp(&key, &s->hint_key, sizeof key) == 0;
            uint32_t measured_scale = params->samples * (params->max_depth + 1u);
            if (reuse) {
                hipLaunchKernelGGL(pt_tile_order_kernel, dim3(1), dim3(1024), 0, stream, n_world_tiles, measured, s->hint_scale, order);
            } else {
                const size_t pixels = (size_t)W.width * W.local_rows;
                if (pixels > s->d_px_state_pixels) {
                    (void)hipFree(s->d_px_state);
                    s->d_px_state = nullptr, s->d_px_state_pixels = 0;
                    HIP_TRY(hipMalloc((void **)&s->d_px_state, pixels * 48u));
                    s->d_px_state_pixels = pixels;
                }
                HIP_TRY(hipMemsetAsync(cost, 0, (size_t)n_world_tiles * sizeof(uint32_t), stream));
                WArgs W1 = W;
                W1.samples = 1, W1.phase = 1, W1.px_state = s->d_px_state, W1.tile_cost = cost, W1.refill_min = 48u;
                hipLaunchKernelGGL(wk, dim3(grid), dim3(kBlock), lds, stream, W1);
                hipLaunchKernelGGL(pt_tile_order_kernel, dim3(1), dim3(1024), 0, stream, n_world_tiles, cost, params->max_depth + 1u, order);
                W.samples = params->samples - 1u, W.phase = 2, W.px_state = s->d_px_state;
                measured_scale = W.samples * (params->max_depth + 1u);
            }
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemsetAsync(s->d_work_counter, 0, sizeof(uint32_t), stream));
            W.tile_order = order;
            if ((s->variant & 8192u) == 0) {
                HIP_TRY(hipMemsetAsync(measured, 0, (size_t)n_world_tiles * sizeof(uint32_t), stream));
                W.tile_cost = measured;
                s->hint_key = key, s->hint_valid = true, s->hint_scale = measured_scale;
            }
        }
        HIP_TRY(hipEventRecord(s->ev_start, stream));
        hipLaunchKernelGGL(wk, dim3(grid), dim3(kBlock), lds, stream, W);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(s->ev_stop, stream));
        s->ev_valid = true;
        s->last_grid = grid, s->last_block = kBlock, s->last_lds = lds;
        return PT_OK;
    }

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
    A.gate = ref_bvh ? s->d_gate : nullptr;   // list semantics: no ancestor-AABB gate, ties to the lower index
    A.gate_chain = s->d_gate_chain;
    A.bvh_large = s->d_bvh_large;
    A.n_bvh_large = s->n_bvh_large;
    A.nodes4 = s->d_nodes4q;
    A.slotrec = s->d_slotrec;
    A.rank_sphere = s->d_rank_sphere;
    A.shade_rank = s->d_shade_rank;
    A.leaf_rank = s->d_leaf_rank;
    memcpy(A.root_min, s->root_min, 12);
    memcpy(A.root_max, s->root_max, 12);
    A.n_spheres = s->n_spheres;
    A.n_spheres_pad = (s->n_spheres + kScanUnroll - 1) / kScanUnroll * kScanUnroll;
    fill_frame_args(A, s, params, cam, frame_num, shard_index, shard_count, d_rgb, d_ray_count);
    HIP_TRY(hipMemsetAsync(s->d_work_counter, 0, sizeof(uint32_t), stream));
    HIP_TRY(hipMemsetAsync(d_ray_count, 0, sizeof(uint64_t), stream));
    if (A.n_items == 0) return PT_OK;

    // ---- LDS carve -----------------------------------------------------------
    uint32_t sph_bytes = 0;
    bool sph_lds = false;
    if (!bvh) {
        sph_bytes = A.n_spheres_pad * 16u;
        sph_lds = (s->variant & 1u) == 0 && sph_bytes <= 64u * 1024u;
        if (!sph_lds) sph_bytes = 0;
    }
    const bool mfma = !bvh && sph_lds && s->n_tiles > 0 && s->n_tiles <= 24u && (s->variant & 4u) == 0;
    A.afrag = s->d_afrag;
    A.tile_sphere = s->d_tile_sphere;
    A.cull_tab = s->d_cull_tab;
    A.cull_axis = (s->variant & 1024u) ? 3u : s->cull_axis;   // variant bit 1024: run every tile
    A.cull_always = s->cull_always;
    A.cull_u0 = s->cull_u0, A.cull_inv_cell = s->cull_inv_cell;
    memcpy(A.clip_min, s->clip_min, 12), memcpy(A.clip_max, s->clip_max, 12);
    if (A.cull_axis < 3u) {
        // Per-RAY reach of the reference's f32 discriminant error (pt_kernel.h lane_tile_mask): the kernel pads the clip box and
        // the segment's extent along the sort axis by sqrt(r_min^2 + kappa (2 |o - c0|^2 + 2 Rs^2 + r_max^2)) - r_min for the
        // ray at hand, so nothing here depends on where the camera is. The constants are rounded up.
        const double kappa = 4.0 * 1.3e-6;
        const double k1 = kappa * (2.0 * (double)s->rs_small * s->rs_small + (double)s->cull_rmax * s->cull_rmax) + (double)s->cull_rmin * s->cull_rmin;
        A.cull_reach[0] = std::nextafter((float)(2.0 * kappa), 3.0e38f);
        A.cull_reach[1] = std::nextafter((float)k1, 3.0e38f);
        A.cull_reach[2] = std::nextafter(s->cull_rmin, 0.0f);
    }
    A.large = s->d_large;
    A.n_tiles = mfma ? s->n_tiles : 0u;
    A.n_large = s->n_large;
    memcpy(A.c0, s->c0, sizeof A.c0);
    A.rs2 = s->rs2;
    A.m0 = s->m0;
    A.gamma = s->gamma;
    A.verify = ((s->variant & 8u) ? 1u : 0u) | ((s->variant & 16u) ? 2u : 0u);  // bit 16: timing experiment, no stack
    A.debug = s->d_debug;
    uint32_t lds = sph_bytes + kLdsParamBytes;
    if (s->has_noise) lds += 4096u + 768u;
    A.ready_min = (uint32_t)kReadyMin, A.drain_at = (uint32_t)(kLeafQ - 4);
    if (dev_knobs().ready >= 0) A.ready_min = (uint32_t)dev_knobs().ready;          // (development knobs PTGPU_READY / PTGPU_DRAIN)
    if (dev_knobs().drain >= 0) A.drain_at = std::min<uint32_t>((uint32_t)dev_knobs().drain, (uint32_t)(kLeafQ - 4));
    // 4-wide tree (default; variant bit 2048: the binary tree): a visit pushes at most three siblings per level
    // (its stack entries are 16-bit node indices; a bigger tree -- more than ~190 000 spheres -- walks the binary one)
    const bool tree4 = bvh && (s->variant & 2048u) == 0 && s->n_nodes4 < 65536u && s->word_ok && s->tree4_packed;
    if (bvh && !tree4)
        if (int rc = ensure_binary_tree(s)) return rc;
    A.wnodes = s->d_wnodes;
    A.n_nodes = s->n_nodes;
    A.bvh_root = tree4 ? (s->has_tree_items ? 0 : -1) : s->bin_root;   // (-1: every sphere is in bvh_large)
    // a visit pushes at most three siblings, and only above the bottom level (3 (depth - 1) entries at most); a bottom node
    // still WRITES its three slots (uncounted), hence + 3. 512 bytes per entry keep every later LDS region 16-byte aligned.
    A.bvh_stack_entries = tree4 ? (3u * (s->depth4 ? s->depth4 - 1u : 0u) + 3u) : (s->bvh_depth + 2u);
    // tree nodes go to LDS only while FOUR workgroups still fit on the CU (with two levels of attenuation stack each):
    // the fourth wave per SIMD is worth more than LDS-resident nodes (random_spheres -B on the tree kernel: 7.4 vs 6.3
    // Grays/s), and the nodes stay L2-resident anyway
    if (bvh) lds += A.bvh_stack_entries * kBlock * (tree4 ? 2u : 4u);
    if (tree4) lds += tree4_queue_bytes(kBlock);
    A.nodes_in_lds = (bvh && !tree4 && (s->variant & 1u) == 0 && lds + s->n_nodes * 64u + 2u * 3u * kBlock * 4u <= kLdsBudget / 4u) ? 1u : 0u;
    if (A.nodes_in_lds) lds += s->n_nodes * 64u;
    // Workgroup size. The MFMA list kernels run ONE 768-thread workgroup per CU when everything fits: the sphere
    // fragments (identical in every workgroup) are staged once per CU, and the LDS that frees holds the per-lane
    // attenuation stacks (levels 1..max_depth-1; level 0 lives in registers), which otherwise stream through L2 to HBM
    // (1.3 GB per 1200x800x64 frame). Variant bit 2 keeps the three 256-thread workgroups with the stack in HBM.
    const uint32_t stack_levels = params->max_depth > 1u ? params->max_depth - 1u : 1u;
    uint32_t blk = kBlock;
    // wide kernels: 16-bit palette codes on the attenuation stack + the shading records in LDS (pt_kernel.h PAL)
    const auto wide_extra = [&](uint32_t b) {
        return ((uint64_t)s->n_spheres + 1ull) * 64ull + (((uint64_t)stack_levels * 2ull * b + 15ull) & ~15ull) +
               (ref_bvh ? (uint64_t)s->n_spheres * 32ull + (((uint64_t)s->n_spheres * 4ull + 15ull) & ~15ull) : 0ull) +   // BVH world: gates + ranks
               (moving ? (uint64_t)s->n_spheres * 32ull : 0ull);                                                          // MovingSphere records
    };
    if (mfma && s->palette_ok && (s->variant & 2u) == 0 && (A.verify & 1u) == 0 && s->blocks_per_cu == 0) {
        const auto wide_lds = [&](uint32_t b) {
            return (uint64_t)lds + mfma_queue_bytes(b) + s->n_tiles * 2048u + ((s->n_tiles * 64u + 15u) & ~15u) + 8u * kCullCells + wide_extra(b);
        };
        // 16 waves per CU (four per SIMD, 128 VGPRs) when the LDS allows, else 12 (variant bit 4096 keeps 12 for A/B runs)
        if (wide_lds(1024u) <= kLdsBudget && (s->variant & 4096u) == 0) blk = 1024u;
        else if (wide_lds(kWideBlock) <= kLdsBudget) blk = kWideBlock;
    }
    if (!bvh) lds += mfma ? mfma_queue_bytes(blk) : scan_queue_bytes(blk);
    if (mfma) lds += s->n_tiles * 2048u + ((s->n_tiles * 64u + 15u) & ~15u) + 8u * kCullCells;
    const uint32_t slots = tree4 ? 1u : 3u;   // attenuation-stack slots per level (4-wide tree kernels: one word, pt_kernel.h WST)
    const uint64_t path_bytes = (uint64_t)stack_levels * slots * blk * 4ull;
    // the 256-thread MFMA variant keeps the attenuation stack in HBM: its LDS goes to the A fragments, and 3 resident
    // workgroups per CU beat 1 with an LDS stack (measured 7.5 vs 3.0 Grays/s)
    // Stack slots (3 per level) kept in LDS. 768-thread kernels: all of them (that is what made them fit). Tree kernels
    // run four workgroups per CU (123 VGPRs): as many levels as fit next to four of them, deeper ones in HBM/L2.
    // Exact-scan list kernels: all or nothing.
    uint32_t lds_levels = 0;
    if (blk == kWideBlock || blk == 1024u) {
        lds_levels = 0;   // (the palette stack is accounted for below)
    } else if (bvh && (s->variant & 2u) == 0) {
        // workgroups per CU the registers allow (4 for the binary tree kernel's 123 VGPRs)
        const uint32_t wg_regs = tree4 ? blocks_per_cu_by_registers(moving ? reinterpret_cast<const void *>(pt_trace_kernel<true, true, false, false, false, true>)
                                                                           : reinterpret_cast<const void *>(pt_trace_kernel<true, true, false, false, false, false>), kBlock) : 4u;
        const uint32_t per_block = kLdsBudget / std::min(4u, std::max(1u, wg_regs));
        if (per_block > lds) lds_levels = std::min<uint32_t>(stack_levels, (per_block - lds) / (slots * blk * 4u));
    } else if (!bvh && !mfma && (s->variant & 2u) == 0 && lds + path_bytes <= kLdsPerBlockMax) {
        lds_levels = stack_levels;
    }
    A.stack_in_lds = lds_levels * slots;
    if (dev_knobs().debug) fprintf(stderr, "[ptgpu launch] bvh %d tree4 %d mfma %d blk %u slots %u stack_levels %u lds_levels %u lds %u\n", (int)bvh, (int)tree4, (int)mfma, blk, slots, stack_levels, lds_levels, lds);
    lds += lds_levels * slots * blk * 4u;
    if (blk == kWideBlock || blk == 1024u) lds += (uint32_t)wide_extra(blk);
    A.lds_sphere_bytes = sph_bytes;

    A.tile_order = nullptr;
    A.tile_cost = nullptr;

    // kernel flavour: pt_trace_kernel<BVH, SPH_LDS, MFMA, VERIFY, PILOT, MOVING, GATE, BLK>; the pilot pass is the PILOT twin
    typedef void (*Kern)(const KArgs);
    Kern kern = nullptr, pilot_kern = nullptr;
    const bool verify = (A.verify & 1u) != 0;
    if (bvh) {   // tree kernels (SPH_LDS = true: the 4-wide tree); verify: count node fetches / sphere tests, one launch, no pilot
        static const Kern tree[2][2][3] = {   // [tree4][moving][main, pilot, verify]
            {{pt_trace_kernel<true, false, false, false, false, false>, pt_trace_kernel<true, false, false, false, true, false>, pt_trace_kernel<true, false, false, true, false, false>},
             {pt_trace_kernel<true, false, false, false, false, true>, pt_trace_kernel<true, false, false, false, true, true>, pt_trace_kernel<true, false, false, true, false, true>}},
            {{pt_trace_kernel<true, true, false, false, false, false>, pt_trace_kernel<true, true, false, false, true, false>, pt_trace_kernel<true, true, false, true, false, false>},
             {pt_trace_kernel<true, true, false, false, false, true>, pt_trace_kernel<true, true, false, false, true, true>, pt_trace_kernel<true, true, false, true, false, true>}}};
        const Kern *t = tree[tree4 ? 1 : 0][moving ? 1 : 0];
        kern = verify ? t[2] : t[0], pilot_kern = verify ? nullptr : t[1];
    } else if (mfma) {   // MFMA list kernels; GATE = a BVH world's accept rules; wide = one 768- or 1024-thread workgroup per CU
        static const Kern list[2][2][7] = {   // [gate][moving][256 main, 256 pilot, verify, 768 main, 768 pilot, 1024 main, 1024 pilot]
            {{pt_trace_kernel<false, true, true, false, false, false, false>, pt_trace_kernel<false, true, true, false, true, false, false>,
              pt_trace_kernel<false, true, true, true, false, false, false>,
              pt_trace_kernel<false, true, true, false, false, false, false, 768>, pt_trace_kernel<false, true, true, false, true, false, false, 768>,
              pt_trace_kernel<false, true, true, false, false, false, false, 1024>, pt_trace_kernel<false, true, true, false, true, false, false, 1024>},
             {pt_trace_kernel<false, true, true, false, false, true, false>, pt_trace_kernel<false, true, true, false, true, true, false>,
              pt_trace_kernel<false, true, true, true, false, true, false>,
              pt_trace_kernel<false, true, true, false, false, true, false, 768>, pt_trace_kernel<false, true, true, false, true, true, false, 768>,
              pt_trace_kernel<false, true, true, false, false, true, false, 1024>, pt_trace_kernel<false, true, true, false, true, true, false, 1024>}},
            {{pt_trace_kernel<false, true, true, false, false, false, true>, pt_trace_kernel<false, true, true, false, true, false, true>,
              pt_trace_kernel<false, true, true, true, false, false, true>,
              pt_trace_kernel<false, true, true, false, false, false, true, 768>, pt_trace_kernel<false, true, true, false, true, false, true, 768>,
              pt_trace_kernel<false, true, true, false, false, false, true, 1024>, pt_trace_kernel<false, true, true, false, true, false, true, 1024>},
             {pt_trace_kernel<false, true, true, false, false, true, true>, pt_trace_kernel<false, true, true, false, true, true, true>,
              pt_trace_kernel<false, true, true, true, false, true, true>,
              pt_trace_kernel<false, true, true, false, false, true, true, 768>, pt_trace_kernel<false, true, true, false, true, true, true, 768>,
              pt_trace_kernel<false, true, true, false, false, true, true, 1024>, pt_trace_kernel<false, true, true, false, true, true, true, 1024>}}};
        const Kern *t = list[ref_bvh ? 1 : 0][moving ? 1 : 0];
        const int w = blk == 1024u ? 5 : (blk == kWideBlock ? 3 : 0);
        kern = verify ? t[2] : t[w], pilot_kern = verify ? nullptr : t[w + 1];
    } else if (sph_lds) {
        kern = pt_trace_kernel<false, true, false, false, false>, pilot_kern = pt_trace_kernel<false, true, false, false, true>;
    } else {
        kern = pt_trace_kernel<false, false, false, false, false>;
    }
    // ---- persistent grid: CUs x resident blocks --------------------------------
    uint32_t bpc = s->blocks_per_cu;
    // 16-wave workgroups batch their refills harder (measured on configs 3 / 4 and `random`: 12 waiting lanes +1.5 % over 4)
    if (blk == 1024u && params->samples >= 32u && dev_knobs().refill < 0) A.refill_min = 12u;
    if (bpc == 0) bpc = (blk == kWideBlock || blk == 1024u) ? 1u : (bvh ? 4u : 3u);
    const uint32_t lds_limit = lds ? (kLdsBudget / lds) : 8u;
    if (bpc > lds_limit) bpc = lds_limit ? lds_limit : 1u;
    if (bpc > 8u) bpc = 8u;
    if (s->blocks_per_cu == 0) bpc = std::min(bpc, blocks_per_cu_by_registers(reinterpret_cast<const void *>(kern), blk));
    uint32_t grid = (uint32_t)s->num_cus * bpc;
    // Fewer pixels than lanes (a shard of an 8-GPU frame: 120 000 pixels for 262 144 lanes): the waves that win the race for
    // work should be spread over ALL CUs -- two waves on a SIMD iterate faster than four -- so a one-workgroup-per-CU grid is
    // not cut down to the workgroups the pixels would fill (a wave that finds the queue empty leaves at once).
    const uint32_t need = (A.n_items + blk - 1) / blk;
    if (grid > need && !((blk == kWideBlock || blk == 1024u) && need * 4u >= grid && !dev_knobs().clamp_grid)) grid = need;
    if (grid == 0) grid = 1;

    if (blk == kBlock && lds_levels < stack_levels) {
        const size_t need_floats = (size_t)grid * params->max_depth * 3ull * blk;
        if (need_floats > s->d_gstack_floats) {
            (void)hipFree(s->d_gstack);
            s->d_gstack = nullptr;
            s->d_gstack_floats = 0;
            HIP_TRY(hipMalloc((void **)&s->d_gstack, need_floats * sizeof(float)));
            s->d_gstack_floats = need_floats;
        }
        A.gstack = s->d_gstack;
    }

    if (pilot_kern)
        if (int rc = set_lds_limit(s, 1, reinterpret_cast<const void *>(pilot_kern), lds)) return rc;
    if (int rc = set_lds_limit(s, 0, reinterpret_cast<const void *>(kern), lds)) return rc;
    HIP_TRY(hipEventRecord(s->ev_pass, stream));

    const bool timing = s->timing;
    unsigned long long *const d_wave_end = s->d_wave_end;
    A.wave_end = timing ? d_wave_end : nullptr;
    if (timing) (void)hipMemsetAsync(d_wave_end, 0, 65536 * 8, stream);
    // ---- heavy-first work order from a 1-spp pilot pass (variant bit 32 disables it) -------------------
    const uint32_t n_work_tiles = A.n_items / kTilePix;
    if (pilot_kern && n_work_tiles >= 256u && params->samples >= kPilotMinSamples && (s->variant & 32u) == 0) {  // the pilot costs ~0.3 ms
        if (n_work_tiles > s->d_tile_cap) {
            (void)hipFree(s->d_tile_buf);
            s->d_tile_buf = nullptr;
            s->d_tile_cap = 0;
            HIP_TRY(hipMalloc((void **)&s->d_tile_buf, (8 + 3 * (size_t)n_work_tiles) * sizeof(uint32_t)));
            s->hint_valid = false;
            s->d_tile_cap = n_work_tiles;
        }
        uint32_t *scratch = s->d_tile_buf, *cost = s->d_tile_buf + 8, *order = cost + s->d_tile_cap, *measured = order + s->d_tile_cap;
        pt_scene::ViewKey key{};
        key.params = *params, key.cam = *cam, key.shard_index = shard_index, key.shard_count = shard_count, key.variant = s->variant, key.n_tiles = n_work_tiles;
        const bool reuse = s->hint_valid && (s->variant & 8192u) == 0 && memcmp(&key, &s->hint_key, sizeof key) == 0;
        // A frame of a view not seen before: TWO launches instead of a throw-away pilot pass. The first (the pilot symbol, phase 1)
        // traces the first sample of every pixel in natural order and counts the rays per tile while doing so (real work, 64
        // pixels per tile instead of the pilot's 16), the second the remaining samples, ordered by those costs; a pixel's RNG
        // stream and colour sum wait in d_px_state in between. Variant bit 16384 keeps the pilot pass.
        uint32_t phase1 = kPhase1Samples;
        if (dev_knobs().phase1 >= 0) phase1 = (uint32_t)dev_knobs().phase1;   // (development knob PTGPU_PHASE1)
        const bool two_phase = !reuse && (s->variant & 16384u) == 0 && phase1 > 0 && params->samples >= kTwoLaunchMinSamples && params->samples > phase1 && (A.verify & 1u) == 0;
        uint32_t measured_scale = params->samples * (params->max_depth + 1u);
        if (reuse) {
            // the last frame of this view measured every tile: order by that (64 buckets over samples x (depth + 1) x 64 pixels)
            hipLaunchKernelGGL(pt_tile_order_kernel, dim3(1), dim3(1024), 0, stream, n_work_tiles, measured, s->hint_scale, order);
            HIP_TRY(hipGetLastError());
        } else if (two_phase) {
            const size_t pixels = (size_t)A.width * A.local_rows;
            if (pixels > s->d_px_state_pixels) {
                (void)hipFree(s->d_px_state);
                s->d_px_state = nullptr, s->d_px_state_pixels = 0;
                HIP_TRY(hipMalloc((void **)&s->d_px_state, pixels * 48u));
                s->d_px_state_pixels = pixels;
            }
            HIP_TRY(hipMemsetAsync(cost, 0, (size_t)n_work_tiles * sizeof(uint32_t), stream));
            KArgs A1 = A;
            A1.samples = phase1, A1.phase = 1, A1.px_state = s->d_px_state, A1.tile_cost = cost, A1.wave_end = nullptr;
            A1.refill_min = 48u;   // one sample per pixel: refills dominate, batch them hard (a plain 1-spp frame: 0.48 ms at 8, 0.34 at 32; frame: best at 48)
            if (dev_knobs().phase1_refill > 0) A1.refill_min = (uint32_t)dev_knobs().phase1_refill;   // (development knob PTGPU_PHASE1_REFILL)
            hipLaunchKernelGGL(pilot_kern, dim3(grid), dim3(blk), lds, stream, A1);
            hipLaunchKernelGGL(pt_tile_order_kernel, dim3(1), dim3(1024), 0, stream, n_work_tiles, cost, phase1 * (params->max_depth + 1u), order);
            HIP_TRY(hipGetLastError());
            A.samples = params->samples - phase1, A.phase = 2, A.px_state = s->d_px_state;
            measured_scale = A.samples * (params->max_depth + 1u);
        } else {
            HIP_TRY(hipMemsetAsync(scratch, 0, (8 + (size_t)s->d_tile_cap) * sizeof(uint32_t), stream));
            KArgs P = A;
            P.samples = 1;
            P.inv_ns = 1.0f;
            P.random_seed = 1;                       // throw-away seeds: the pilot must not look like frame data
            P.seed_base = 0x9e3779b97f4a7c15ull ^ frame_num;
            // (P.rgb stays the frame: PILOT kernels never write pixels)
            P.tile_cost = cost;
            P.ray_count = reinterpret_cast<unsigned long long *>(scratch);      // scratch[0..1]
            P.verify = 0;
            P.wave_end = nullptr;
            // a third of the frame's grid: the pilot has ~100x less work, and each workgroup stages the scene into LDS
            uint32_t pilot_div = 3u;
            if (dev_knobs().pilot_div > 0) pilot_div = (uint32_t)dev_knobs().pilot_div;   // (development knob PTGPU_PILOT_DIV)
            hipLaunchKernelGGL(pilot_kern, dim3((grid + pilot_div - 1u) / pilot_div), dim3(blk), lds, stream, P);
            hipLaunchKernelGGL(pt_tile_order_kernel, dim3(1), dim3(1024), 0, stream, n_work_tiles, cost, params->max_depth + 1u, order);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipMemsetAsync(s->d_work_counter, 0, sizeof(uint32_t), stream));
        // 16-wave workgroups: the waves' first fetches are handed out by age class (pt_kernel.h first_static; +1 % on configs 3 / 4);
        // variant bit 32768 leaves them to the race for the counter
        if (blk == 1024u && (uint64_t)grid * 1024ull <= A.n_items && (s->variant & 32768u) == 0) A.first_static = grid * 1024u;
        A.tile_order = order;
        if ((s->variant & 8192u) == 0) {   // this frame measures the tiles for the next one (after the order kernel has read the old values)
            HIP_TRY(hipMemsetAsync(measured, 0, (size_t)n_work_tiles * sizeof(uint32_t), stream));
            A.tile_cost = measured;
            s->hint_key = key, s->hint_valid = true, s->hint_scale = measured_scale;
        }
    }

    HIP_TRY(hipEventRecord(s->ev_start, stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(blk), lds, stream, A);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->ev_stop, stream));
    s->ev_valid = true;
    s->last_grid = grid;
    s->last_block = blk;
    s->last_lds = lds;
#ifdef PT_CULLSTATS
    {   // development aid (-DPT_CULLSTATS builds only): how many tiles the culling leaves
        (void)hipStreamSynchronize(stream);
        unsigned long long c[48];
        (void)hipMemcpy(c, s->d_debug + 24, sizeof c, hipMemcpyDeviceToHost);
        (void)hipMemset(s->d_debug + 24, 0, sizeof c);
        if (c[0]) {
            fprintf(stderr, "[ptgpu cull] wave-iterations %llu, tiles run per iteration %.2f of %u; lanes asked for %.2f tiles each\n  run histogram:",
                    c[0], (double)c[1] / (double)c[0], A.n_tiles, (double)c[3] / (double)(c[2] ? c[2] : 1));
            for (int i = 0; i < 18; ++i) fprintf(stderr, " %d:%.1f%%", i, 100.0 * (double)c[4 + i] / (double)c[0]);
            fprintf(stderr, "\n  lane histogram:");
            for (int i = 0; i < 18; ++i) fprintf(stderr, " %d:%.1f%%", i, 100.0 * (double)c[24 + i] / (double)(c[2] ? c[2] : 1));
            fprintf(stderr, "\n");
        }
    }
#endif
#ifdef PT_SECTIONS
    {   // development aid (-DPT_SECTIONS builds only): where the waves' cycles go
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
    if (timing) {  // development aid: distribution of wave finish times
        (void)hipStreamSynchronize(stream);
        const uint32_t nw = grid * (blk / 64);
        std::vector<unsigned long long> t(nw), all(13 * (size_t)nw);
        (void)hipMemcpy(all.data(), d_wave_end, std::min<size_t>(13 * (size_t)nw, 65536) * 8, hipMemcpyDeviceToHost);
        std::copy(all.begin(), all.begin() + nw, t.begin());
#if defined(PT_SECTIONS) || defined(PT_WAVEDBG)
        if (13u * nw <= 65536u) {   // the last finishers: iterations, when they last fetched pixels, when they started
            std::vector<uint32_t> idx(nw);
            for (uint32_t i = 0; i < nw; ++i) idx[i] = i;
            std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return all[a] > all[b]; });
            unsigned long long t_first = ~0ull;
            for (uint32_t i = 0; i < nw; ++i) t_first = std::min(t_first, all[3 * (size_t)nw + i]);
            double it_sum = 0;
            for (uint32_t i = 0; i < nw; ++i) it_sum += (double)(all[nw + i] & 0xffffffffull);
            fprintf(stderr, "[ptgpu timing] mean iterations per wave %.0f\n", it_sum / nw);
            for (uint32_t r : {0u, 1u, 2u, 5u, 10u, 20u, 40u, 100u, 400u, 2000u, nw - 1u}) {
                if (r >= nw) continue;
                const uint32_t w = idx[r];
                fprintf(stderr, "  lane 0: first pixel (%u, %u), last pixel (%u, %u);", (unsigned)(all[4 * (size_t)nw + w] & 0xffffu), (unsigned)((all[4 * (size_t)nw + w] >> 16) & 0xffffu),
                        (unsigned)((all[4 * (size_t)nw + w] >> 32) & 0xffffu), (unsigned)(all[4 * (size_t)nw + w] >> 48));
#ifdef PT_SECTIONS
                fprintf(stderr, " kcycles: refill %.0f camera %.0f features %.0f tiles %.0f phase2 %.0f shade %.0f;", all[5 * (size_t)nw + w] * 1e-3, all[6 * (size_t)nw + w] * 1e-3,
                        all[10 * (size_t)nw + w] * 1e-3, all[11 * (size_t)nw + w] * 1e-3, (all[7 * (size_t)nw + w] - all[10 * (size_t)nw + w] - all[11 * (size_t)nw + w]) * 1e-3, all[8 * (size_t)nw + w] * 1e-3);
#endif
#ifdef PT_WAVEDBG
                fprintf(stderr, " tiles %llu pairs %llu overflows %llu drains %llu;", all[5 * (size_t)nw + w], all[6 * (size_t)nw + w], all[7 * (size_t)nw + w], all[8 * (size_t)nw + w]);
#endif
                fprintf(stderr, "  rank %4u wave %4u: end %.3f ms; first empty-handed fetch (lane 0) at %.3f ms after %llu of %llu iterations\n", r, w,
                        (all[w] - t_first) * 1e-5, all[2 * (size_t)nw + w] ? (all[2 * (size_t)nw + w] - t_first) * 1e-5 : -1.0,
                        all[nw + w] >> 32, all[nw + w] & 0xffffffffull);
            }
        }
#endif
        std::sort(t.begin(), t.end());
        const double tick_ns = 10.0;  // wall_clock64: 100 MHz
        fprintf(stderr, "[ptgpu timing] waves %u: finish spread (ms after first finisher) p10 %.3f p50 %.3f p90 %.3f p99 %.3f last %.3f\n", nw,
                (t[nw / 10] - t[0]) * tick_ns * 1e-6, (t[nw / 2] - t[0]) * tick_ns * 1e-6, (t[nw * 9 / 10] - t[0]) * tick_ns * 1e-6,
                (t[nw * 99 / 100] - t[0]) * tick_ns * 1e-6, (t[nw - 1] - t[0]) * tick_ns * 1e-6);
    }
    return PT_OK;
}

}  // namespace

extern "C" int pt_render_device(pt_scene *s, const pt_params *params, const pt_camera *cam, uint32_t frame_num,
                                float *d_rgb_inout, uint64_t *d_ray_count, void *hip_stream) {
    return launch(s, params, cam, frame_num, 0, 1, d_rgb_inout, d_ray_count, reinterpret_cast<hipStream_t>(hip_stream));
}

extern "C" int pt_render_shard_device(pt_scene *s, const pt_params *params, const pt_camera *cam, uint32_t frame_num,
                                      uint32_t shard_index, uint32_t shard_count, float *d_rgb_shard_inout,
                                      uint64_t *d_ray_count, void *hip_stream) {
    return launch(s, params, cam, frame_num, shard_index, shard_count, d_rgb_shard_inout, d_ray_count,
                  reinterpret_cast<hipStream_t>(hip_stream));
}

namespace {
// frame buffer + pinned staging copy used by the host-buffer entry point
int ensure_frame_buffers(pt_scene *s, size_t floats) {
    if (floats <= s->d_frame_floats) return PT_OK;
    (void)hipFree(s->d_frame);
    (void)hipHostFree(s->h_stage);
    s->d_frame = nullptr;
    s->h_stage = nullptr;
    s->d_frame_floats = 0;
    HIP_TRY(hipMalloc((void **)&s->d_frame, floats * sizeof(float)));
    // pinned staging buffer: the caller's Vec is pageable, and a pageable hipMemcpy runs at a fraction of the
    // link rate; CPU memcpy into pinned memory + DMA is ~2x faster for the 11.5 MB frame (a failure is not fatal)
    if (hipHostMalloc((void **)&s->h_stage, floats * sizeof(float), hipHostMallocDefault) != hipSuccess) s->h_stage = nullptr;
    s->d_frame_floats = floats;
    return PT_OK;
}
}  // namespace

extern "C" int pt_scene_prepare(pt_scene *s, const pt_params *params) {
    if (!s || !params) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    if (params->width == 0 || params->height == 0 || params->samples == 0)
        return fail(PT_ERR_INVALID_ARG, "width/height/samples must be non-zero");
    if ((uint64_t)params->width * params->height > 0x3fffffffull) return fail(PT_ERR_INVALID_ARG, "frame too large");
    HIP_TRY(hipSetDevice(s->device));
    if (int rc = ensure_frame_buffers(s, (size_t)params->width * params->height * 3u)) return rc;
    if (params->use_bvh && s->bvh_root < 0) return PT_OK;   // (pt_render will report the missing tree)
    // one throw-away frame with the caller's geometry of launch (samples only scale the work, except that the
    // heavy-first pilot pass needs kPilotMinSamples of them to be scheduled at all, the two-launch frame kTwoLaunchMinSamples):
    // allocates every lazily sized buffer
    pt_params p = *params;
    p.samples = params->samples >= kTwoLaunchMinSamples ? kTwoLaunchMinSamples : (params->samples >= kPilotMinSamples ? kPilotMinSamples : 1u);
    p.max_depth = params->max_depth;
    pt_camera cam;
    memset(&cam, 0, sizeof cam);
    cam.lower_left_corner[0] = cam.lower_left_corner[1] = cam.lower_left_corner[2] = -1.0f;
    cam.horizontal[0] = 2.0f;
    cam.vertical[1] = 2.0f;
    HIP_TRY(hipMemsetAsync(s->d_frame, 0, (size_t)params->width * params->height * 3u * sizeof(float), nullptr));
    if (int rc = launch(s, &p, &cam, 0, 0, 1, s->d_frame, reinterpret_cast<uint64_t *>(s->d_ray_count), nullptr)) return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));
    s->ev_valid = false;
    return PT_OK;
}

// Host buffers the caller registered (pt_buffer_register): pt_render then renders straight into them over PCIe instead of
// staging through a device frame (H2D + D2H + two CPU copies of 11.5 MB at 1200x800).
namespace {
struct RegisteredBuffer {
    void *host;
    size_t bytes;
};
std::vector<RegisteredBuffer> g_registered;   // (registration is rare and process-wide; guarded by g_reg_mutex)
std::mutex g_reg_mutex;

bool registered_device_ptr(const void *host, size_t bytes, void **dev_out) {
    std::lock_guard<std::mutex> lock(g_reg_mutex);
    for (const RegisteredBuffer &r : g_registered) {
        const char *b = static_cast<const char *>(r.host), *p = static_cast<const char *>(host);
        if (p >= b && p + bytes <= b + r.bytes) {
            void *d = nullptr;
            if (hipHostGetDevicePointer(&d, r.host, 0) != hipSuccess || !d) return false;
            *dev_out = static_cast<char *>(d) + (p - b);
            return true;
        }
    }
    return false;
}
}  // namespace

extern "C" int pt_buffer_register(void *host_ptr, size_t bytes) {
    if (!host_ptr || bytes == 0) return fail(PT_ERR_INVALID_ARG, "NULL buffer / zero size");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PT_ERR_NO_DEVICE, "no HIP device available");
    HIP_TRY(hipHostRegister(host_ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable));
    std::lock_guard<std::mutex> lock(g_reg_mutex);
    g_registered.push_back(RegisteredBuffer{host_ptr, bytes});
    return PT_OK;
}

extern "C" int pt_buffer_unregister(void *host_ptr) {
    if (!host_ptr) return fail(PT_ERR_INVALID_ARG, "NULL buffer");
    {
        std::lock_guard<std::mutex> lock(g_reg_mutex);
        size_t i = 0;
        while (i < g_registered.size() && g_registered[i].host != host_ptr) ++i;
        if (i == g_registered.size()) return fail(PT_ERR_INVALID_ARG, "buffer was not registered with pt_buffer_register");
        g_registered.erase(g_registered.begin() + (long)i);
    }
    HIP_TRY(hipHostUnregister(host_ptr));
    return PT_OK;
}

extern "C" int pt_render(pt_scene *s, const pt_params *params, const pt_camera *cam, uint32_t frame_num, float *rgb_inout,
                         uint64_t *ray_count_out) {
    if (!s || !params || !cam || !rgb_inout || !ray_count_out) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    if (params->width == 0 || params->height == 0 || params->samples == 0)
        return fail(PT_ERR_INVALID_ARG, "width/height/samples must be non-zero");
    HIP_TRY(hipSetDevice(s->device));
    const size_t floats = (size_t)params->width * params->height * 3u;
    void *mapped = nullptr;
    if (registered_device_ptr(rgb_inout, floats * sizeof(float), &mapped)) {
        // registered (pinned + mapped) caller buffer: the kernel reads the previous frame and writes the new one in place,
        // pixel by pixel as lanes finish them -- the transfers ride under the render, nothing is staged or copied afterwards
        int rc = launch(s, params, cam, frame_num, 0, 1, static_cast<float *>(mapped), reinterpret_cast<uint64_t *>(s->d_ray_count), nullptr);
        if (rc != PT_OK) return rc;
        unsigned long long rc64 = 0;
        HIP_TRY(hipMemcpy(&rc64, s->d_ray_count, sizeof rc64, hipMemcpyDeviceToHost));   // (synchronises the null stream)
        *ray_count_out = rc64;
        return PT_OK;
    }
    if (int rc0 = ensure_frame_buffers(s, floats)) return rc0;
    // the buffer is read (frame blend, scene.rs:114-116) and written
    if (s->h_stage) {
        memcpy(s->h_stage, rgb_inout, floats * sizeof(float));
        HIP_TRY(hipMemcpyAsync(s->d_frame, s->h_stage, floats * sizeof(float), hipMemcpyHostToDevice, nullptr));
    } else {
        HIP_TRY(hipMemcpy(s->d_frame, rgb_inout, floats * sizeof(float), hipMemcpyHostToDevice));
    }
    int rc = launch(s, params, cam, frame_num, 0, 1, s->d_frame, reinterpret_cast<uint64_t *>(s->d_ray_count), nullptr);
    if (rc != PT_OK) return rc;
    if (s->h_stage) {
        HIP_TRY(hipMemcpyAsync(s->h_stage, s->d_frame, floats * sizeof(float), hipMemcpyDeviceToHost, nullptr));
        HIP_TRY(hipStreamSynchronize(nullptr));
        memcpy(rgb_inout, s->h_stage, floats * sizeof(float));
    } else {
        HIP_TRY(hipMemcpy(rgb_inout, s->d_frame, floats * sizeof(float), hipMemcpyDeviceToHost));
    }
    unsigned long long rc64 = 0;
    HIP_TRY(hipMemcpy(&rc64, s->d_ray_count, sizeof rc64, hipMemcpyDeviceToHost));
    *ray_count_out = rc64;
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

extern "C" int pt_scene_build_info(pt_scene *s, float *build_ms_out, uint32_t *n_nodes_out, uint32_t *depth_out, uint32_t *on_device_out) {
    if (!s) return fail(PT_ERR_INVALID_ARG, "scene is NULL");
    if (build_ms_out) *build_ms_out = s->tree_build_ms;
    if (n_nodes_out) *n_nodes_out = s->n_nodes4;
    if (depth_out) *depth_out = s->depth4;
    if (on_device_out) *on_device_out = s->tree_on_device ? 1u : 0u;
    return PT_OK;
}

extern "C" int pt_scene_debug_tree(pt_scene *s, void *nodes_out, size_t capacity_bytes) {
    if (!s || !nodes_out) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    const size_t bytes = (size_t)s->n_nodes4 * sizeof(DNode4);
    if (capacity_bytes < bytes) return fail(PT_ERR_INVALID_ARG, "buffer holds %zu bytes, the tree has %zu", capacity_bytes, bytes);
    HIP_TRY(hipSetDevice(s->device));
    if (bytes) HIP_TRY(hipMemcpy(nodes_out, s->d_nodes4, bytes, hipMemcpyDeviceToHost));
    return PT_OK;
}

extern "C" int pt_scene_debug_tree_packed(pt_scene *s, void *nodes_out, size_t capacity_bytes, uint32_t *usable_out) {
    if (!s || !nodes_out) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    const size_t bytes = s->d_nodes4q ? (size_t)s->n_nodes4 * sizeof(DNode4Q) : 0;
    if (capacity_bytes < bytes) return fail(PT_ERR_INVALID_ARG, "buffer holds %zu bytes, the packed tree has %zu", capacity_bytes, bytes);
    HIP_TRY(hipSetDevice(s->device));
    if (bytes) HIP_TRY(hipMemcpy(nodes_out, s->d_nodes4q, bytes, hipMemcpyDeviceToHost));
    if (usable_out) *usable_out = s->tree4_packed ? 1u : 0u;
    return PT_OK;
}

extern "C" int pt_last_launch_info(pt_scene *s, uint32_t *grid_out, uint32_t *block_out, uint32_t *lds_bytes_out) {
    if (!s) return fail(PT_ERR_INVALID_ARG, "scene is NULL");
    if (grid_out) *grid_out = s->last_grid;
    if (block_out) *block_out = s->last_block;
    if (lds_bytes_out) *lds_bytes_out = s->last_lds;
    return PT_OK;
}

// ---- multi-GPU frames: RCCL communicator behind the C ABI -------------------------
// One frame is split by rows (row y -> rank y % world, disjoint pixels as scene.rs:90-93); the only exchange is ONE
// ncclAllGather / ncclGather of the float3 shards plus an 8-byte ncclAllReduce of the ray count (scene.rs:118-120).
// xGMI is point-to-point and the message is small (11.5 MB at 1200x800), so one collective, no ring tuning.
struct pt_comm {
    ncclComm_t comm = nullptr;
    int device = 0;
    uint32_t rank = 0, world = 1;
    float *d_gather = nullptr;       // [world][ceil(H / world)][W][3]; this rank's shard is rendered in place in slot `rank`
    size_t gather_floats = 0;
};

#define NCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess) return fail(PT_ERR_HIP, "%s failed: %s", #expr, ncclGetErrorString(r_)); \
    } while (0)

namespace {

__global__ void shard_pack_kernel(const float *full, float *shard, uint32_t row_floats, uint32_t rows, uint32_t index, uint32_t count) {
    const size_t n = (size_t)rows * row_floats;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t j = i / row_floats, x = i - j * row_floats;
        shard[i] = full[(j * count + index) * row_floats + x];
    }
}

// row y of the frame = gathered[y % count][y / count]
__global__ void shard_unpack_kernel(const float *gathered, float *full, uint32_t row_floats, uint32_t height, uint32_t count, uint32_t prow) {
    const size_t n = (size_t)height * row_floats;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t y = i / row_floats, x = i - y * row_floats;
        full[i] = gathered[((y % count) * prow + y / count) * row_floats + x];
    }
}

uint32_t copy_grid(size_t n) { return (uint32_t)std::min<size_t>((n + 255) / 256, 4096); }

int comm_ensure(pt_comm *c, uint32_t width, uint32_t height) {
    const uint32_t prow = (height + c->world - 1) / c->world;
    const size_t need = (size_t)c->world * prow * width * 3u;
    if (need <= c->gather_floats) return PT_OK;
    (void)hipFree(c->d_gather);
    c->d_gather = nullptr;
    c->gather_floats = 0;
    HIP_TRY(hipMalloc((void **)&c->d_gather, need * sizeof(float)));
    HIP_TRY(hipMemset(c->d_gather, 0, need * sizeof(float)));   // ranks with one row less send a zero row
    c->gather_floats = need;
    return PT_OK;
}

// shard already sits in slot `rank` of c->d_gather
int comm_exchange(pt_comm *c, uint32_t width, uint32_t height, float *d_rgb_full, uint64_t *d_ray_count, int root, hipStream_t stream) {
    const uint32_t prow = (height + c->world - 1) / c->world;
    const size_t slot = (size_t)prow * width * 3u;
    const bool have_frame = root < 0 || (uint32_t)root == c->rank;
    if (have_frame && !d_rgb_full) return fail(PT_ERR_INVALID_ARG, "d_rgb_full is NULL on a rank that receives the frame");
    {   // (a one-rank communicator goes through the same calls: that is what a 1-GPU box can test)
        NCCL_TRY(ncclGroupStart());
        if (root < 0)
            NCCL_TRY(ncclAllGather(c->d_gather + (size_t)c->rank * slot, c->d_gather, slot, ncclFloat, c->comm, stream));
        else
            NCCL_TRY(ncclGather(c->d_gather + (size_t)c->rank * slot, c->d_gather, slot, ncclFloat, root, c->comm, stream));
        NCCL_TRY(ncclAllReduce(d_ray_count, d_ray_count, 1, ncclUint64, ncclSum, c->comm, stream));
        NCCL_TRY(ncclGroupEnd());
    }
    if (have_frame) {
        const size_t n = (size_t)height * width * 3u;
        hipLaunchKernelGGL(shard_unpack_kernel, dim3(copy_grid(n)), dim3(256), 0, stream, c->d_gather, d_rgb_full, width * 3u, height, c->world, prow);
        HIP_TRY(hipGetLastError());
    }
    return PT_OK;
}

}  // namespace

extern "C" int pt_comm_unique_id(uint8_t id_out[PT_COMM_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == PT_COMM_ID_BYTES, "PT_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
    if (!id_out) return fail(PT_ERR_INVALID_ARG, "id_out is NULL");
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return PT_OK;
}

extern "C" int pt_comm_create(const uint8_t id[PT_COMM_ID_BYTES], uint32_t rank, uint32_t world, int device, pt_comm **comm_out) {
    if (!id || !comm_out) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    *comm_out = nullptr;
    if (world == 0 || rank >= world) return fail(PT_ERR_INVALID_ARG, "bad rank %u of %u", rank, world);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PT_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(PT_ERR_INVALID_ARG, "device %d out of range (%d devices)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    pt_comm *c = new (std::nothrow) pt_comm();
    if (!c) return fail(PT_ERR_INVALID_ARG, "out of host memory");
    c->device = device, c->rank = rank, c->world = world;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    const ncclResult_t r = ncclCommInitRank(&c->comm, (int)world, uid, (int)rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(PT_ERR_HIP, "ncclCommInitRank failed: %s", ncclGetErrorString(r));
    }
    *comm_out = c;
    return PT_OK;
}

extern "C" int pt_comm_create_all(const int *devices, uint32_t n, pt_comm **comms_out) {
    if (!devices || !comms_out || n == 0) return fail(PT_ERR_INVALID_ARG, "NULL argument / no devices");
    for (uint32_t i = 0; i < n; ++i) comms_out[i] = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PT_ERR_NO_DEVICE, "no HIP device available");
    for (uint32_t i = 0; i < n; ++i)
        if (devices[i] < 0 || devices[i] >= ndev) return fail(PT_ERR_INVALID_ARG, "device %d out of range (%d devices)", devices[i], ndev);
    std::vector<ncclComm_t> cs(n, nullptr);
    NCCL_TRY(ncclCommInitAll(cs.data(), (int)n, devices));
    for (uint32_t i = 0; i < n; ++i) {
        pt_comm *c = new (std::nothrow) pt_comm();
        if (!c) return fail(PT_ERR_INVALID_ARG, "out of host memory");
        c->comm = cs[i], c->device = devices[i], c->rank = i, c->world = n;
        comms_out[i] = c;
    }
    return PT_OK;
}

extern "C" void pt_comm_destroy(pt_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    (void)hipFree(c->d_gather);
    delete c;
}

extern "C" int pt_comm_rank(const pt_comm *c, uint32_t *rank_out, uint32_t *world_out) {
    if (!c) return fail(PT_ERR_INVALID_ARG, "comm is NULL");
    if (rank_out) *rank_out = c->rank;
    if (world_out) *world_out = c->world;
    return PT_OK;
}

extern "C" int pt_shard_pack(const float *d_rgb_full, float *d_rgb_shard, uint32_t width, uint32_t height, uint32_t shard_index,
                             uint32_t shard_count, void *hip_stream) {
    if (!d_rgb_full || !d_rgb_shard || width == 0 || height == 0) return fail(PT_ERR_INVALID_ARG, "NULL argument / empty frame");
    if (shard_count == 0 || shard_index >= shard_count) return fail(PT_ERR_INVALID_ARG, "bad shard %u/%u", shard_index, shard_count);
    const uint32_t rows = pt_shard_rows(height, shard_index, shard_count);
    if (rows == 0) return PT_OK;
    const size_t n = (size_t)rows * width * 3u;
    hipLaunchKernelGGL(shard_pack_kernel, dim3(copy_grid(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream), d_rgb_full, d_rgb_shard,
                       width * 3u, rows, shard_index, shard_count);
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

extern "C" int pt_shard_unpack_all(const float *d_gathered, float *d_rgb_full, uint32_t width, uint32_t height, uint32_t shard_count,
                                   void *hip_stream) {
    if (!d_gathered || !d_rgb_full || width == 0 || height == 0 || shard_count == 0) return fail(PT_ERR_INVALID_ARG, "NULL argument / empty frame");
    const size_t n = (size_t)height * width * 3u;
    hipLaunchKernelGGL(shard_unpack_kernel, dim3(copy_grid(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream), d_gathered, d_rgb_full,
                       width * 3u, height, shard_count, (height + shard_count - 1) / shard_count);
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

extern "C" int pt_comm_gather_frame(pt_comm *c, uint32_t width, uint32_t height, const float *d_rgb_shard, float *d_rgb_full,
                                    uint64_t *d_ray_count, int root, void *hip_stream) {
    if (!c || !d_rgb_shard || !d_ray_count || width == 0 || height == 0) return fail(PT_ERR_INVALID_ARG, "NULL argument / empty frame");
    if (root >= (int)c->world) return fail(PT_ERR_INVALID_ARG, "root %d out of range (%u ranks)", root, c->world);
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = comm_ensure(c, width, height)) return rc;
    hipStream_t stream = reinterpret_cast<hipStream_t>(hip_stream);
    const uint32_t prow = (height + c->world - 1) / c->world, rows = pt_shard_rows(height, c->rank, c->world);
    if (rows)
        HIP_TRY(hipMemcpyAsync(c->d_gather + (size_t)c->rank * prow * width * 3u, d_rgb_shard, (size_t)rows * width * 3u * sizeof(float),
                               hipMemcpyDeviceToDevice, stream));
    return comm_exchange(c, width, height, d_rgb_full, d_ray_count, root, stream);
}

extern "C" int pt_render_sharded(pt_scene *s, pt_comm *c, const pt_params *params, const pt_camera *cam, uint32_t frame_num,
                                 float *d_rgb_full_inout, uint64_t *d_ray_count, int root, void *hip_stream) {
    if (!s || !c || !params || !cam || !d_rgb_full_inout || !d_ray_count) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    if (params->width == 0 || params->height == 0 || params->samples == 0)
        return fail(PT_ERR_INVALID_ARG, "width/height/samples must be non-zero");
    if (s->device != c->device) return fail(PT_ERR_INVALID_ARG, "scene lives on device %d, communicator on %d", s->device, c->device);
    if (root >= (int)c->world) return fail(PT_ERR_INVALID_ARG, "root %d out of range (%u ranks)", root, c->world);
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = comm_ensure(c, params->width, params->height)) return rc;
    hipStream_t stream = reinterpret_cast<hipStream_t>(hip_stream);
    const uint32_t prow = (params->height + c->world - 1) / c->world;
    float *slot = c->d_gather + (size_t)c->rank * prow * params->width * 3u;
    // the blend reads the previous frame (scene.rs:114-116): this rank's rows, compacted into its gather slot
    if (int rc = pt_shard_pack(d_rgb_full_inout, slot, params->width, params->height, c->rank, c->world, hip_stream)) return rc;
    if (int rc = launch(s, params, cam, frame_num, c->rank, c->world, slot, d_ray_count, stream)) return rc;
    return comm_exchange(c, params->width, params->height, d_rgb_full_inout, d_ray_count, root, stream);
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
}  // namespace

extern "C" int pt_selftest_probe(int device, uint32_t probe, const float *in, float *out, size_t n) {
    if (!in || !out) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    if (probe > PT_PROBE_LN) return fail(PT_ERR_INVALID_ARG, "unknown probe %u", probe);
    if (n == 0) return PT_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PT_ERR_NO_DEVICE, "no HIP device available");
    HIP_TRY(hipSetDevice(device));
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
    if (reset) HIP_TRY(hipMemset(s->d_debug + 8, 0, 16));
    return PT_OK;
}

extern "C" int pt_scene_debug_counters(pt_scene *s, uint64_t out4[4], int reset) {
    if (!s || !out4) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out4, s->d_debug, 32, hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(hipMemset(s->d_debug, 0, 1024));
    return PT_OK;
}
