// pt_prep.hip -- host-side analysis of a scene description: validation, the flattened device layouts, the MFMA prefilter's
// sphere fragments and tile-culling tables, and the host restatements of the internal traversal trees. No HIP call in this
// file: pt_scene.hip uploads what it produces, pt_debug_select (pt_api.hip) runs it on machines without a GPU.
#include "pt_host.h"
#include <limits>

#include <algorithm>
#include <array>
#include <cmath>

namespace pthostside {

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
                    const pt_perlin *perlin, bool allow_isotropic, bool *has_noise_out, uint32_t n_images,
                    const pt_image *images) {
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
    // that axis and a wave can skip the tiles no ray of it comes near (pt_prefilter.h lane_tile_mask). Oversized spheres
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
        forced = dev_knobs().cull_axis;   // (development knob: force the sort axis, 3 = culling off)
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
            // Tiles as BOXES in two axes: the spheres of the full tiles, sorted along `ax`, are cut into strips of whole tiles and
            // each strip is sorted along a second axis `bx` (the wider of the other two), so a tile covers a short interval on
            // both. The rays of a wave start close together and leave in all directions: the tiles they can reach are those near
            // a disc around the origins, and boxes cover a disc with fewer spheres than slabs across the whole scene do
            // (random_spheres: 7.1 of 16 tiles run per wave-iteration with slabs along x, 6.2 with 4 strips of 4 tiles; 3 to 8 strips measure alike).
            int bx = (ax + 1) % 3;
            {
                double ext[3];
                for (int k = 0; k < 3; ++k) {
                    double lo_k = 1e300, hi_k = -1e300;
                    for (size_t q = 0; q < full_tiles * 32; ++q) lo_k = std::min(lo_k, sw[regular[q]].c[k]), hi_k = std::max(hi_k, sw[regular[q]].c[k]);
                    ext[k] = hi_k - lo_k;
                }
                const int other = (ax + 2) % 3;
                if (ext[other] > ext[bx]) bx = other;
                uint32_t strips = 1;
                if (ext[bx] > 0.25 * ext[ax] && full_tiles >= 4) strips = (uint32_t)std::lround(std::sqrt((double)full_tiles));
                if (dev_knobs().cull_strips > 0) strips = (uint32_t)dev_knobs().cull_strips;
                strips = std::max(1u, std::min(strips, (uint32_t)full_tiles));
                const size_t tps = (full_tiles + strips - 1) / strips;   // tiles per strip
                for (size_t t0 = 0; t0 < full_tiles && strips > 1; t0 += tps) {   // (one strip: the tiles stay slabs of the sort axis)
                    const size_t q0 = t0 * 32, q1 = std::min(t0 + tps, full_tiles) * 32;
                    std::stable_sort(regular.begin() + q0, regular.begin() + q1, [&](uint32_t a, uint32_t b) { return sw[a].c[bx] < sw[b].c[bx]; });
                }
                out.cull_strips = strips;
            }
            small = regular;
            small.insert(small.end(), big.begin(), big.end());
            std::vector<double> lo(out.n_tiles, 0.0), hi(out.n_tiles, 0.0), lo2(out.n_tiles, 0.0), hi2(out.n_tiles, 0.0);
            double bmin[3] = {1e300, 1e300, 1e300}, bmax[3] = {-1e300, -1e300, -1e300};
            for (uint32_t T = 0; T < out.n_tiles; ++T) {
                if (T >= full_tiles) {
                    out.cull_always |= 1u << T;
                    continue;
                }
                lo[T] = lo2[T] = 1e300, hi[T] = hi2[T] = -1e300;
                for (size_t j = (size_t)T * 32; j < (size_t)T * 32 + 32; ++j) {
                    const uint32_t i = small[j];
                    const double r = radius_of(i);
                    lo[T] = std::min(lo[T], sw[i].c[ax] - r), hi[T] = std::max(hi[T], sw[i].c[ax] + r);
                    lo2[T] = std::min(lo2[T], sw[i].c[bx] - r), hi2[T] = std::max(hi2[T], sw[i].c[bx] + r);
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
            out.cull_axis2 = (uint32_t)bx;
            double rmin = 1e300, rmax = 0.0;
            for (size_t j = 0; j < full_tiles * 32; ++j) {
                const double r = std::fabs((double)desc->spheres[small[j]].radius);
                rmin = std::min(rmin, r), rmax = std::max(rmax, r);
            }
            out.cull_rmin = (float)rmin, out.cull_rmax = (float)rmax;
            out.cull_tab.assign(4 * kCullCells, 0u);
            // one pair of tables per axis: [tiles reaching cell c or beyond][tiles starting at cell c or before]
            auto axis_tables = [&](int axis, const std::vector<double> &tlo, const std::vector<double> &thi, uint32_t *tab, float &u0, float &inv_cell) {
                u0 = out.clip_min[axis];
                const double cell = std::max(((double)out.clip_max[axis] - (double)out.clip_min[axis]) / (double)kCullCells, 1e-30);
                inv_cell = (float)(1.0 / cell);
                for (int c = 0; c < kCullCells; ++c) {
                    // cell c as the DEVICE sees it: a coordinate u lands in cell clamp(int((u - u0) * inv_cell)); one extra cell
                    // of slack on each side covers the f32 rounding of that expression
                    const double c_lo = (c == 0) ? -1e300 : (double)u0 + (c - 1) * cell;
                    const double c_hi = (c == kCullCells - 1) ? 1e300 : (double)u0 + (c + 2) * cell;
                    for (uint32_t T = 0; T < (uint32_t)full_tiles; ++T) {
                        const double pad = 2e-3 + 1e-5 * std::max(std::fabs(tlo[T]), std::fabs(thi[T]));
                        if (thi[T] + pad >= c_lo) tab[c] |= 1u << T;                 // tiles reaching cell c or beyond
                        if (tlo[T] - pad <= c_hi) tab[kCullCells + c] |= 1u << T;   // tiles starting at cell c or before
                    }
                }
            };
            axis_tables(ax, lo, hi, out.cull_tab.data(), out.cull_u0, out.cull_inv_cell);
            axis_tables(bx, lo2, hi2, out.cull_tab.data() + 2 * kCullCells, out.cull_u0_2, out.cull_inv_cell_2);
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
                out.tile_sphere[(size_t)T * 32 + tile_bit_of_row(row)] = (uint16_t)small[j];
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

// ---- internal BVH for BVH mode ------------------------------------------------------------------
// The caller's tree (bvh.rs:64-94: random split axis, median split) defines the RESULT of BVHNode::ray_hit
// but can be arbitrarily bad for traversal (random_spheres / perlin_spheres: every sphere has the same y, so
// a third of the levels do not separate anything: ~1800 node visits per ray in the 10k-sphere scene).
// pt_scene_create therefore builds its own tree (longest-axis median split over sphere centres; spheres
// with a huge radius are kept out and tested for every ray) and keeps, per sphere, the AABB of its parent in
// the CALLER's tree, which is all that is needed to reproduce the reference's accept/reject decision.
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

// ---- uniform cell grid (pt_host.h GridPlan; walked by csrc/pt_grid.h) ------------------------------------------------------------------
// Eligibility and geometry are decided here, from the description alone: at least kGridMinItems spheres of similar size (those wider
// than 2.5 median radii are tested for every ray instead, at most kGridMaxLarge of them), a cell size and alignment picked from a
// small set by the estimated work per unit of ray length (records met per cell / h), bounded tables.
namespace {
constexpr uint32_t kGridMinItems = 1024u, kGridMaxLarge = 16u, kGridMaxCells = 1u << 20, kGridMaxRecords = 1u << 21;
constexpr uint32_t kGridNone = 0x7fffffffu, kGridLink = 0x80000000u;

struct GridGeom {
    uint32_t n[3];
    double gmin[3], h, ha[3];   // h: the cubic cell; ha[k]: the cell's size along axis k (h, or the whole extent along an axis with ONE cell)
};
// records a cell with `items` spheres takes: four in the last one, three in every one before it
// does the ball (c, R) reach into cell (x, y, z)? (plain spheres are registered by their padded BALL, not its box: the corner cells of
// the box drop out -- a ray the reference's test accepts passes through the ball, hence through a cell the ball reaches into)
inline bool grid_ball_in_cell(const GridGeom &g, const double c[3], double R, uint32_t x, uint32_t y, uint32_t z) {
    const uint32_t i[3] = {x, y, z};
    double d2 = 0.0;
    for (int k = 0; k < 3; ++k) {
        const double lo = g.gmin[k] + g.ha[k] * (double)i[k], hi = lo + g.ha[k];
        const double d = c[k] < lo ? lo - c[k] : (c[k] > hi ? c[k] - hi : 0.0);
        d2 += d * d;
    }
    return d2 <= R * R;
}
inline uint32_t grid_records_of(uint32_t items) { return items <= 4u ? 1u : 1u + (items - 4u + 2u) / 3u; }
inline void grid_cell_range(const GridGeom &g, const double lo[3], const double hi[3], uint32_t a[3], uint32_t b[3]) {
    for (int k = 0; k < 3; ++k) {
        const double fa = std::floor((lo[k] - g.gmin[k]) / g.ha[k]), fb = std::floor((hi[k] - g.gmin[k]) / g.ha[k]);
        const double top = (double)g.n[k] - 1.0;   // (comparisons written so that a NaN lands on a valid cell: nothing non-finite gets this far, but a cast of one is undefined)
        a[k] = fa > 0.0 ? (fa < top ? (uint32_t)fa : g.n[k] - 1u) : 0u;
        b[k] = fb > 0.0 ? (fb < top ? (uint32_t)fb : g.n[k] - 1u) : 0u;
    }
}
}  // namespace

bool plan_cell_grid(const pt_scene_desc *desc, const MotionIn *motion, double t_lo, double t_hi, const std::vector<float4> &sph, GridPlan &G) {
    G = GridPlan{};
    std::vector<uint32_t> large;
    std::vector<AccelItem> items = accel_items(desc, motion, t_lo, t_hi, large);
    if (items.size() < kGridMinItems) return false;
    std::vector<float> radii;
    for (const AccelItem &it : items) radii.push_back(it.r);
    std::nth_element(radii.begin(), radii.begin() + radii.size() / 2, radii.end());
    const double r_med = radii[radii.size() / 2];
    {   // the few wide ones leave the grid
        std::vector<AccelItem> keep;
        for (const AccelItem &it : items) {
            double w = 0.0;
            for (int k = 0; k < 3; ++k) w = std::max(w, 0.5 * ((double)it.mx[k] - (double)it.mn[k]));
            if (w > 2.5 * r_med) large.push_back(it.sphere);
            else keep.push_back(it);
        }
        items.swap(keep);
    }
    if (large.size() > kGridMaxLarge || items.size() < kGridMinItems) return false;
    for (const AccelItem &it : items)   // (a MovingSphere's swept box can overflow f32: such a scene keeps the tree)
        for (int k = 0; k < 3; ++k)
            if (!std::isfinite(it.mn[k]) || !std::isfinite(it.mx[k]) || !std::isfinite(it.c[k]) || !(it.mn[k] <= it.mx[k])) return false;
    std::sort(large.begin(), large.end());
    double bmin[3] = {1e300, 1e300, 1e300}, bmax[3] = {-1e300, -1e300, -1e300}, r_min = 1e300, r_max = 0.0;
    for (const AccelItem &it : items) {
        for (int k = 0; k < 3; ++k) bmin[k] = std::min(bmin[k], (double)it.mn[k]), bmax[k] = std::max(bmax[k], (double)it.mx[k]);
        r_min = std::min(r_min, (double)it.r), r_max = std::max(r_max, (double)it.r);
    }
    double hd2 = 0.0;
    for (int k = 0; k < 3; ++k) hd2 += 0.25 * (bmax[k] - bmin[k]) * (bmax[k] - bmin[k]);
    const double half_diag = std::sqrt(hd2);
    // rays from within three half-diagonals of the grid's centre walk one line (a camera framing the spheres is at about two)
    const double d_build = 3.0 * half_diag, infl = 1.0e-6;
    const auto inflate = [&](double r) { return infl * (d_build * d_build + r * r) / r; };
    const double w_med = 2.0 * (r_med + inflate(r_med));
    if (!(inflate(r_min) <= 0.5 * r_med) || !std::isfinite(half_diag) || !(half_diag > 0.0)) return false;   // too little precision left at that distance: the tree pads per ray
    // padded boxes (the walk's own rounding is covered by h / 1000 more, added per candidate geometry below)
    std::vector<double> lo(3 * items.size()), hi(3 * items.size()), ctr(3 * items.size()), rad(items.size());
    std::vector<char> ball(items.size());   // a plain sphere (its box is the ball's): registered by the ball; a moving one by the box of its sweep
    for (size_t i = 0; i < items.size(); ++i) {
        const double d = inflate(items[i].r);
        ball[i] = !(motion && motion[items[i].sphere].moving);
        rad[i] = (double)items[i].r + d;
        for (int k = 0; k < 3; ++k) lo[3 * i + k] = (double)items[i].mn[k] - d, hi[3 * i + k] = (double)items[i].mx[k] + d, ctr[3 * i + k] = (double)items[i].c[k];
    }
    // candidates: cell sizes around the padded median width, four alignments per axis that has more than one cell
    GridGeom best{};
    double best_cost = 1e300, best_sz = 0.0;
    const size_t stride = (items.size() + 19999) / 20000;
    std::vector<uint32_t> count;
    static const double kSizes[] = {0.625, 0.75, 0.875, 1.0, 1.25, 1.5, 2.0, 3.0};
    for (double sz : kSizes) {
        const double h = sz * w_med;
        for (int align = 0; align < 64; ++align) {
            GridGeom g{};
            g.h = h;
            bool skip = false;
            uint64_t cells = 1;
            for (int k = 0; k < 3; ++k) {
                const double ext = (bmax[k] - bmin[k]) + 2.0 * inflate(r_min) + 4.0e-3 * h;
                const int sh = (align >> (2 * k)) & 3;
                const bool single = ext <= 1.5 * h;
                if (single && sh != 0) skip = true;   // (alignment means nothing along an axis with one cell)
                const double start = bmin[k] - inflate(r_min) - 2.0e-3 * h - (single ? 0.0 : 0.25 * sh * h);
                g.gmin[k] = start;
                g.n[k] = single ? 1u : (uint32_t)std::ceil((bmax[k] + inflate(r_min) + 2.0e-3 * h - start) / h);
                g.ha[k] = single ? ext : h;   // (one cell along this axis: it spans the spheres' whole extent, up to 1.5 h)
                if (g.n[k] == 0u || g.n[k] > 1023u) skip = true;
                cells *= g.n[k];
            }
            if (skip || cells > kGridMaxCells) continue;
            count.assign((size_t)cells, 0u);
            for (size_t i = 0; i < items.size(); i += stride) {   // (big scenes: the candidates are compared on every stride-th sphere)
                double l[3], u[3];
                for (int k = 0; k < 3; ++k) l[k] = lo[3 * i + k] - 1.0e-3 * h, u[k] = hi[3 * i + k] + 1.0e-3 * h;
                uint32_t a[3], b[3];
                grid_cell_range(g, l, u, a, b);
                for (uint32_t z = a[2]; z <= b[2]; ++z)
                    for (uint32_t y = a[1]; y <= b[1]; ++y)
                        for (uint32_t x = a[0]; x <= b[0]; ++x)
                            if (!ball[i] || grid_ball_in_cell(g, &ctr[3 * i], rad[i] + 1.0e-3 * h, x, y, z)) count[((size_t)z * g.n[1] + y) * g.n[0] + x] += 1u;
            }
            uint64_t records = 0;
            for (uint32_t c : count) records += grid_records_of((uint32_t)std::min<uint64_t>((uint64_t)c * stride, 0xffffffffull));
            if (records > kGridMaxRecords) continue;
            const double cost = ((double)records / (double)cells + 0.35) / h;   // (+ the walk's own step per cell)
            if (cost < best_cost) best_cost = cost, best = g, best_sz = sz;
        }
    }
    if (!(best_cost < 1e300)) return false;
    const GridGeom &g = best;
    // The walk forms cell boundaries, entry points and the grid's corner in f32 AT the grid's coordinates; the registrations are padded by
    // h / 1000 for that rounding. Far from the origin an ulp outgrows the pad (r = 0.2 spheres around x = 20 000: ulp 2e-3, pad 6e-4): such a
    // field keeps the tree, whose boxes are padded per ray. (2^-22 |x| = two ulps at |x|; a quarter of the pad is the budget.)
    {
        double far = 0.0, h_min = g.h;
        for (int k = 0; k < 3; ++k) far = std::max(far, std::max(std::fabs(g.gmin[k]), std::fabs(g.gmin[k] + g.n[k] * g.ha[k]))), h_min = std::min(h_min, g.ha[k]);
        if (far * (1.0 / 4194304.0) > 0.25e-3 * h_min) return false;
    }
    const size_t cells = (size_t)g.n[0] * g.n[1] * g.n[2];
    std::vector<std::vector<uint32_t>> lists(cells);
    size_t regs = 0;
    for (size_t i = 0; i < items.size(); ++i) {
        double l[3], u[3];
        for (int k = 0; k < 3; ++k) l[k] = lo[3 * i + k] - 1.0e-3 * g.h, u[k] = hi[3 * i + k] + 1.0e-3 * g.h;
        uint32_t a[3], b[3];
        grid_cell_range(g, l, u, a, b);
        for (uint32_t z = a[2]; z <= b[2]; ++z)
            for (uint32_t y = a[1]; y <= b[1]; ++y)
                for (uint32_t x = a[0]; x <= b[0]; ++x)
                    if (!ball[i] || grid_ball_in_cell(g, &ctr[3 * i], rad[i] + 1.0e-3 * g.h, x, y, z)) lists[((size_t)z * g.n[1] + y) * g.n[0] + x].push_back(items[i].sphere), ++regs;
    }
    union FU { float f; uint32_t u; };
    const auto bits = [](float f) { FU q; q.f = f; return q.u; };
    const float4 never = make_float4(3.0e38f, 3.0e38f, 3.0e38f, 0.0f);
    std::vector<uint4> rec(5 * cells);
    for (size_t c = 0; c < cells; ++c) {
        std::vector<uint32_t> &L = lists[c];
        std::sort(L.begin(), L.end());
        size_t at = c, done = 0;
        for (;;) {
            const size_t left = L.size() - done, take = left <= 4 ? left : 3;
            uint32_t ids[4] = {kGridNone, kGridNone, kGridNone, kGridNone};
            float4 four[4] = {never, never, never, never};
            for (size_t j = 0; j < take; ++j) four[j] = sph[L[done + j]], ids[j] = L[done + j];
            // spheres 0 | 1 and 2 | 3 side by side, component by component: a 16-byte load fills two operand PAIRS of the packed f32 instructions
            for (size_t pr = 0; pr < 2; ++pr) {
                const float4 u = four[2 * pr], v = four[2 * pr + 1];
                rec[5 * at + 2 * pr] = make_uint4(bits(u.x), bits(v.x), bits(u.y), bits(v.y));
                rec[5 * at + 2 * pr + 1] = make_uint4(bits(u.z), bits(v.z), bits(u.w), bits(v.w));
            }
            done += take;
            if (done == L.size()) {
                rec[5 * at + 4] = make_uint4(ids[0], ids[1], ids[2], ids[3]);
                break;
            }
            const size_t next = rec.size() / 5;
            rec[5 * at + 4] = make_uint4(ids[0], ids[1], ids[2], kGridLink | (uint32_t)next);
            rec.resize(rec.size() + 5);
            at = next;
        }
    }
    if (rec.size() / 5 > kGridMaxRecords) return false;   // (the candidates of a big scene were only estimated)
    G.ok = true;
    for (int k = 0; k < 3; ++k) G.n[k] = g.n[k], G.gmin[k] = (float)g.gmin[k], G.centre[k] = (float)(0.5 * (bmin[k] + bmax[k]));
    G.h = (float)g.h;
    for (int k = 0; k < 3; ++k) G.ha[k] = (float)g.ha[k];
    // (the kernel works with these f32 values; the registration above used the doubles they were rounded from -- h / 1000 covers the difference)
    G.half_diag = (float)(half_diag * 1.0001), G.d_build = (float)(d_build * 0.999);
    G.n_records = (uint32_t)(rec.size() / 5);
    G.cells.swap(rec);
    G.large.swap(large);
    G.items_per_cell = (double)regs / (double)cells, G.records_per_cell = (double)G.n_records / (double)cells;
    size_t occupied = 0;
    for (const std::vector<uint32_t> &L : lists) occupied += L.empty() ? 0u : 1u;
    G.occupied = (double)occupied / (double)cells;
    // Where the walk pays (tools/grid_ab.py, grid against the 4-wide tree on one MI355X, round 6's walk): an even, DENSE field of spheres whose cells are
    // at most two median sphere widths wide -- config 5's single layer (4.1 spheres per cell, every cell occupied, cells of 1.25 widths) 1.5x; 10 000
    // spheres filling a cube with 3.3 ... 5.3 per cell and 83 ... 89 % of the cells occupied 1.11 ... 1.39x -- and not a looser field, for which the cost
    // estimate above picks cells of three widths (4.6 per cell, 89 % occupied: 0.89 ... 0.95x: most of a cell's spheres are far from the ray that
    // crosses it), a thin layer (2.3 per cell, 81 %: 0.92x) or a loose cloud (1.1 ... 2.5 per cell, 63 ... 81 %: 0.70 ... 0.93x; the tree skips empty
    // space, the walk steps through it). NOTES.md "Round 6 -- which fields get a cell grid" has the table.
    double occ_min = 0.83, per_cell_min = 2.5, sz_max = 2.0;
#ifdef PT_DEVKNOBS
    if (const char *e = getenv("PTGPU_GRID_OCC")) occ_min = atof(e);       // development builds: tools/grid_ab.py sweeps the thresholds
    if (const char *e = getenv("PTGPU_GRID_PER_CELL")) per_cell_min = atof(e);
    if (const char *e = getenv("PTGPU_GRID_SZ")) sz_max = atof(e);
#endif
    if (G.occupied < occ_min || G.items_per_cell < per_cell_min || best_sz > sz_max) {
        G = GridPlan{};
        return false;
    }
#ifdef PT_GRID_ROUNDS
    fprintf(stderr, "[ptgpu grid plan] %u x %u x %u cells of %.4f (median padded width %.4f), %zu spheres + %zu large, %.2f spheres and %.2f records per cell, %.0f %% of the cells occupied, d_build %.1f half_diag %.1f\n", G.n[0], G.n[1], G.n[2],
            g.h, w_med, items.size(), G.large.size(), G.items_per_cell, G.records_per_cell, 100.0 * G.occupied, d_build, half_diag);
#endif
    return true;
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

// ---- Scene::new, host half: everything pt_scene_create derives from the description before the device is touched ----
int plan_sphere_scene(const pt_scene_desc *desc, const MotionIn *motion, SpherePlan &P) {
    if (!desc) return fail(PT_ERR_INVALID_ARG, "desc is NULL");
    if (desc->n_spheres == 0 || !desc->spheres || !desc->sphere_material) return fail(PT_ERR_INVALID_ARG, "scene has no spheres");
    if (desc->n_materials == 0 || !desc->materials) return fail(PT_ERR_INVALID_ARG, "scene has no materials");
    if (desc->n_textures && !desc->textures) return fail(PT_ERR_INVALID_ARG, "textures is NULL");
    if (desc->n_spheres > 0x7fffffffu) return fail(PT_ERR_INVALID_ARG, "too many spheres");
    bool has_noise = false;
    // (an Isotropic row may sit in the table -- a world description shares it with its media -- but no sphere may use it)
    if (int rc = validate_tables(desc->n_materials, desc->materials, desc->n_textures, desc->textures, desc->perlin, true, &has_noise, 0, nullptr)) return rc;
    for (uint32_t i = 0; i < desc->n_spheres; ++i) {
        if (desc->sphere_material[i] >= desc->n_materials) return fail(PT_ERR_INVALID_ARG, "sphere %u: material index out of range", i);
        if (desc->materials[desc->sphere_material[i]].kind == PT_MAT_ISOTROPIC)
            return fail(PT_ERR_INVALID_ARG, "sphere %u: Isotropic is only valid as a medium's phase function", i);
    }
    if (desc->n_bvh_nodes) {
        if (!desc->bvh_nodes) return fail(PT_ERR_INVALID_ARG, "bvh_nodes is NULL");
        // (the caller's depth is irrelevant for sphere scenes: traversal runs over the internal tree)
        if (bvh_depth_checked(desc->bvh_nodes, desc->n_bvh_nodes, desc->n_spheres, desc->bvh_root) == 0)
            return fail(PT_ERR_INVALID_ARG, "malformed BVH (bad child index or cycle)");
    }
    ptsel::SceneTraits &tr = P.tr;
    tr = ptsel::SceneTraits{};
    tr.n_spheres = desc->n_spheres;
    tr.has_caller_bvh = desc->n_bvh_nodes != 0;
    tr.has_noise = has_noise;
    // the time interval the moving entries are defined over: the sweeps are bounded for ray times inside it
    double t_lo = 0.0, t_hi = 0.0;
    if (motion) {
        bool first = true;
        for (uint32_t i = 0; i < desc->n_spheres; ++i) {
            if (!motion[i].moving) continue;
            const double a0 = motion[i].time_start, a1 = a0 + 1.0 / (double)motion[i].inv_time_delta;
            if (!std::isfinite(a0) || !std::isfinite(a1)) return fail(PT_ERR_UNSUPPORTED, "moving sphere %u has a degenerate time interval", i);
            t_lo = first ? std::min(a0, a1) : std::min(t_lo, std::min(a0, a1));
            t_hi = first ? std::max(a0, a1) : std::max(t_hi, std::max(a0, a1));
            first = false;
        }
        tr.has_motion = !first;
        if (first) motion = nullptr;
        tr.time_lo = (float)t_lo, tr.time_hi = (float)t_hi;
    }
    P.has_motion = motion != nullptr;
    P.t_lo = t_lo, P.t_hi = t_hi;

    // flatten to the device layouts
    const uint32_t n_pad = ptsel::scan_pad(desc->n_spheres);
    P.sph.assign(desc->n_spheres, make_float4(0, 0, 0, 0));
    P.sph_r2.assign(n_pad, make_float4(3.0e38f, 3.0e38f, 3.0e38f, 0.0f));
    for (uint32_t i = 0; i < desc->n_spheres; ++i) {
        const pt_sphere &p = desc->spheres[i];
        P.sph[i] = make_float4(p.cx, p.cy, p.cz, p.radius);
        const volatile float r2 = p.radius * p.radius;  // sphere.rs:36, one f32 rounding
        P.sph_r2[i] = make_float4(p.cx, p.cy, p.cz, r2);
    }
    // per-sphere shading records (one 64-byte fetch per hit)
    P.shade.assign(4 * (size_t)desc->n_spheres, make_float4(0, 0, 0, 0));
    bool palette_ok = true;   // every scattering material's attenuation is a per-sphere constant or one of two checker colours
    bool word_ok = true;      // every attenuation fits one stack word: Noise -> its grey value; Constant / Checker2 / metal / glass -> a code
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
            } else if (t.kind == PT_TEX_CHECKER && desc->textures[t.odd].kind == PT_TEX_CONSTANT && desc->textures[t.even].kind == PT_TEX_CONSTANT) {
                flags = kShadeChecker2;
                const pt_texture &o = desc->textures[t.odd], &e = desc->textures[t.even];
                qa = make_float4(o.color[0], o.color[1], o.color[2], 0.f);
                qb = make_float4(e.color[0], e.color[1], e.color[2], 0.f);
            } else if (t.kind == PT_TEX_NOISE) {
                flags = kShadeNoise;
                qa = make_float4(t.scale, 0.f, 0.f, 0.f);
            }
        }
        if (m.kind == PT_MAT_DIELECTRIC) {
            // the two per-material constants of a Dielectric scatter, in the reference's f32 arithmetic (math.rs:77-78 r0 of schlick,
            // material.rs:105 1.0 / ref_idx): two IEEE divisions the kernel would otherwise repeat at every glass hit
            const volatile float one_minus = 1.0f - m.param, one_plus = 1.0f + m.param;
            volatile float r0 = one_minus / one_plus;
            r0 = r0 * r0;
            const volatile float inv = 1.0f / m.param;
            qb = make_float4(r0, inv, 0.f, 0.f);
        }
        if (m.kind == PT_MAT_LAMBERTIAN && (flags & (kShadeConst | kShadeChecker2)) == 0) palette_ok = false;
        if (m.kind == PT_MAT_LAMBERTIAN && flags == 0) word_ok = false;
        if (m.kind > PT_MAT_DIFFUSE_LIGHT) palette_ok = false, word_ok = false;
        // 1 / radius for the normal's division (pt_device.h divs3_known); NaN = "divide in full" for radii near the ends of the
        // exponent range
        {
            const float ar = std::fabs(p.radius);
            const volatile float inv_r = 1.0f / p.radius;
            qa.w = (ar >= 0x1p-20f && ar <= 0x1p20f) ? (float)inv_r : std::numeric_limits<float>::quiet_NaN();
        }
        union { uint32_t u; float f; } k{m.kind}, fl{flags}, tx{(uint32_t)m.texture};
        P.shade[4 * i] = make_float4(p.cx, p.cy, p.cz, p.radius);
        P.shade[4 * i + 1] = make_float4(k.f, fl.f, tx.f, m.param);
        P.shade[4 * i + 2] = qa;
        P.shade[4 * i + 3] = qb;
    }
    tr.palette_ok = palette_ok;
    tr.word_ok = word_ok && desc->n_spheres < 0xFFFFFu;
    P.mats.resize(desc->n_materials);
    for (uint32_t i = 0; i < desc->n_materials; ++i) {
        const pt_material &m = desc->materials[i];
        P.mats[i] = DMat{m.kind, m.albedo[0], m.albedo[1], m.albedo[2], m.param, m.texture, 0.f, 0.f};
    }
    P.texs.resize(desc->n_textures ? desc->n_textures : 1);
    for (uint32_t i = 0; i < desc->n_textures; ++i) {
        const pt_texture &t = desc->textures[i];
        P.texs[i] = DTex{t.kind, t.color[0], t.color[1], t.color[2], t.odd, t.even, t.scale, 0.f};
    }
    // BVH mode: per-sphere parent AABB + DFS rank from the CALLER's tree (they define the result)
    P.leaf_rank.assign(desc->n_spheres, 0);
    P.gate.assign(2 * (size_t)desc->n_spheres, make_float4(0, 0, 0, 0));
    if (desc->n_bvh_nodes) {
        union FU { uint32_t u; float f; };
        // a sphere that is not a leaf of the caller's tree can never be hit: chain count 0xffffffff = "never"
        const FU never{0xffffffffu};
        for (uint32_t i = 0; i < desc->n_spheres; ++i) P.gate[2 * i] = make_float4(0, 0, 0, never.f);
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
                P.rank_sphere.push_back(k);
                P.leaf_rank[k] = rank++;
                const pt_bvh_node &pn = desc->bvh_nodes[it.parent];
                FU cnt{0}, off{(uint32_t)(P.gate_chain.size() / 2)};
                for (size_t j = path.size() - 1; j-- > 0;) {   // grandparent upwards
                    const pt_bvh_node &up = desc->bvh_nodes[path[j]], &low = desc->bvh_nodes[path[j + 1]];
                    if (!implied_by(up, low)) {
                        P.gate_chain.push_back(make_float4(up.min[0], up.min[1], up.min[2], 0.f));
                        P.gate_chain.push_back(make_float4(up.max[0], up.max[1], up.max[2], 0.f));
                        ++cnt.u;
                    }
                }
                P.gate[2 * k] = make_float4(pn.min[0], pn.min[1], pn.min[2], cnt.f);
                P.gate[2 * k + 1] = make_float4(pn.max[0], pn.max[1], pn.max[2], off.f);
            } else {
                path.push_back(it.ref);
                st.push_back({desc->bvh_nodes[it.ref].rhs, it.ref, it.depth + 1});
                st.push_back({desc->bvh_nodes[it.ref].lhs, it.ref, it.depth + 1});
            }
        }
    }
    {   // the internal tree is built for every scene: BVH mode always may use it, list mode uses it for scenes too large for
        // the scan (there it needs no gate: closest t, ties to the lower list index)
        const std::vector<AccelItem> items = accel_items(desc, motion, t_lo, t_hi, P.bvh_large);
        P.titems.resize(items.size());
        for (size_t i = 0; i < items.size(); ++i) {
            P.titems[i].sphere = items[i].sphere, P.titems[i].r = items[i].r;
            memcpy(P.titems[i].c, items[i].c, 12), memcpy(P.titems[i].mn, items[i].mn, 12), memcpy(P.titems[i].mx, items[i].mx, 12);
        }
    }
    tr.grid_ok = plan_cell_grid(desc, motion, t_lo, t_hi, P.sph, P.grid);
    if (tr.grid_ok) {   // what drain_pairs4 reads for a queued sphere: ONE 64-byte fetch per exact test, as the tree's leaf slots
        P.grid.rec.resize(4 * (size_t)desc->n_spheres);
        for (uint32_t i = 0; i < desc->n_spheres; ++i) {
            union { uint32_t u; float f; } rk{P.leaf_rank[i]}, ix{i};
            P.grid.rec[4 * (size_t)i] = P.sph[i], P.grid.rec[4 * (size_t)i + 1] = P.gate[2 * (size_t)i], P.grid.rec[4 * (size_t)i + 2] = P.gate[2 * (size_t)i + 1];
            P.grid.rec[4 * (size_t)i + 3] = make_float4(rk.f, ix.f, 0.f, 0.f);
        }
    }
    P.pvec.assign(256, make_float4(0, 0, 0, 0));
    P.pperm.assign(768, 0);
    if (desc->perlin) {
        for (int i = 0; i < 256; ++i) {
            P.pvec[i] = make_float4(desc->perlin->randvec[i][0], desc->perlin->randvec[i][1], desc->perlin->randvec[i][2], 0.f);
            P.pperm[i] = desc->perlin->perm_x[i];
            P.pperm[256 + i] = desc->perlin->perm_y[i];
            P.pperm[512 + i] = desc->perlin->perm_z[i];
        }
    }
    if (desc->n_bvh_nodes) {
        P.leafrec.resize(4 * (size_t)desc->n_spheres);
        for (uint32_t i = 0; i < desc->n_spheres; ++i) {
            union { uint32_t u; float f; } rk{P.leaf_rank[i]};
            P.leafrec[4 * i] = P.sph[i], P.leafrec[4 * i + 1] = P.gate[2 * i], P.leafrec[4 * i + 2] = P.gate[2 * i + 1];
            P.leafrec[4 * i + 3] = make_float4(rk.f, 0.f, 0.f, 0.f);
        }
    }
    P.shade_rank.resize(4 * P.rank_sphere.size());
    for (size_t r = 0; r < P.rank_sphere.size(); ++r)
        for (int q = 0; q < 4; ++q) P.shade_rank[4 * r + q] = P.shade[4 * (size_t)P.rank_sphere[r] + q];
    if (motion) {
        P.mot.assign(2 * (size_t)desc->n_spheres, make_float4(0, 0, 0, 0));
        for (uint32_t i = 0; i < desc->n_spheres; ++i) {
            if (!motion[i].moving) continue;
            P.mot[2 * i] = make_float4(motion[i].delta[0], motion[i].delta[1], motion[i].delta[2], motion[i].inv_time_delta);
            P.mot[2 * i + 1] = make_float4(motion[i].time_start, 1.0f, 0.f, 0.f);
        }
    }
    P.has_prep = prepare_mfma(desc, motion, t_lo, t_hi, P.prep);
    tr.n_tiles = P.has_prep ? P.prep.n_tiles : 0u;
    return PT_OK;
}

// A world description seen as a sphere scene when every entry is a Sphere or a MovingSphere (the specialised kernels then
// apply); also the world-level facts the general kernel's selection needs.
int analyze_world(const pt_world_desc *desc, WorldAsSpheres &W) {
    if (!desc) return fail(PT_ERR_INVALID_ARG, "desc is NULL");
    // an EMPTY list is a valid world (HitableList::ray_hit returns None for every ray: the `final` preset)
    if (desc->n_hitables && !desc->hitables) return fail(PT_ERR_INVALID_ARG, "hitables is NULL");
    if (desc->n_hitables > 0x3fffffffu) return fail(PT_ERR_INVALID_ARG, "too many hitables");
    if (desc->n_hitables && (desc->n_materials == 0 || !desc->materials)) return fail(PT_ERR_INVALID_ARG, "world has no materials");
    if (desc->n_materials && !desc->materials) return fail(PT_ERR_INVALID_ARG, "materials is NULL");
    if (desc->n_hitables == 0 && desc->n_bvh_nodes) return fail(PT_ERR_INVALID_ARG, "BVH nodes over an empty list");
    if (desc->n_textures && !desc->textures) return fail(PT_ERR_INVALID_ARG, "textures is NULL");
    if (desc->n_transforms && !desc->transforms) return fail(PT_ERR_INVALID_ARG, "transforms is NULL");
    // pt_hitable.transform keeps a chain's level counts in bits 20..27: a plain index up there would be read as a chain
    if (desc->n_transforms > (1u << 20)) return fail(PT_ERR_UNSUPPORTED, "more than 2^20 transforms");
    bool has_noise = false;
    if (int rc = validate_tables(desc->n_materials, desc->materials, desc->n_textures, desc->textures, desc->perlin, true, &has_noise, desc->n_images, desc->images))
        return rc;
    W.has_noise = has_noise;
    W.has_image = false;
    for (uint32_t i = 0; i < desc->n_textures; ++i) W.has_image = W.has_image || desc->textures[i].kind == PT_TEX_IMAGE;
    W.all_spheres = true, W.sphere_like = true, W.has_media = false;
    W.is_graph = desc->n_nodes != 0;   // (a graph that flattens arrives here without nodes: flatten_world_graph)
    std::vector<uint8_t> member(desc->n_hitables, 0);   // 1: a child of a PT_HIT_MEDIUM_GROUP (asked through its group only)
    uint32_t members_left = 0;
    for (uint32_t i = 0; i < desc->n_hitables; ++i) {
        const pt_hitable &h = desc->hitables[i];
        if (h.kind == PT_HIT_MEDIUM_GROUP) {   // a ConstantMedium around a HitableList: this entry + the next p[0] entries (include/ptgpu.h)
            uint32_t n_members;
            memcpy(&n_members, &h.p[0], 4);
            if (members_left) return fail(PT_ERR_INVALID_ARG, "hitable %u: a medium group inside a medium group", i);
            // (a group's members are asked through the group only, so their materials are never validated -- which holds for a LIST, the
            //  flattener's own output. A scene graph's Hitable nodes can point at any entry, a member included: the interpreted walk would
            //  then shade with the member's unchecked material index. Graphs say "a medium around a list" with nodes, never with groups.)
            if (W.is_graph) return fail(PT_ERR_INVALID_ARG, "hitable %u: PT_HIT_MEDIUM_GROUP entries cannot be combined with scene-graph nodes (n_nodes = %u)", i, desc->n_nodes);
            if (n_members == 0 || (uint64_t)i + n_members > (uint64_t)desc->n_hitables - 1ull)
                return fail(PT_ERR_INVALID_ARG, "hitable %u: a medium group of %u entries does not fit the list", i, n_members);
            if (h.medium_material < 0 || (uint32_t)h.medium_material >= desc->n_materials || desc->materials[h.medium_material].kind != PT_MAT_ISOTROPIC)
                return fail(PT_ERR_INVALID_ARG, "hitable %u: medium_material must index an Isotropic material", i);
            if (h.transform >= 0) {   // outer levels only
                const uint32_t u = (uint32_t)h.transform;
                if ((u >> 28) != 0u || ((u >> 20) & 15u) != 0u || ((u >> 24) & 15u) == 0u || (uint64_t)(u & 0xfffffu) + ((u >> 24) & 15u) > desc->n_transforms)
                    return fail(PT_ERR_INVALID_ARG, "hitable %u: transform index / chain out of range", i);
            }
            W.has_media = W.has_chains = true;
            W.all_spheres = W.sphere_like = false;
            members_left = n_members;
            continue;
        }
        if (h.kind > PT_HIT_CUBOID) return fail(PT_ERR_INVALID_ARG, "hitable %u: unknown kind %u", i, h.kind);
        if (members_left) {   // a child of the group above: a shape under Instance levels of its own
            members_left -= 1;
            member[i] = 1;
            if (h.medium_material >= 0) return fail(PT_ERR_INVALID_ARG, "hitable %u: a medium inside a medium group", i);
            if (h.transform >= 0) {
                const uint32_t u = (uint32_t)h.transform, ext = u >> 20;
                const uint64_t last = ext ? (uint64_t)(u & 0xfffffu) + ((u >> 20) & 15u) : (uint64_t)u + 1u;
                if (last > desc->n_transforms || (ext && ((u >> 24) != 0u || ((u >> 20) & 15u) == 0u)))
                    return fail(PT_ERR_INVALID_ARG, "hitable %u: transform index / chain out of range", i);
            }
            continue;   // (its material is never looked at: a hit inside the group is the medium's)
        }
        if (h.material >= desc->n_materials) return fail(PT_ERR_INVALID_ARG, "hitable %u: material index out of range", i);
        if (desc->materials[h.material].kind == PT_MAT_ISOTROPIC)
            return fail(PT_ERR_INVALID_ARG, "hitable %u: Isotropic is only valid as a medium's phase function", i);
        if (h.transform >= 0) {   // a plain index, or first | inner levels << 20 | outer levels << 24 (include/ptgpu.h pt_hitable)
            const uint32_t u = (uint32_t)h.transform, ext = u >> 20;
            const uint64_t last = ext ? (uint64_t)(u & 0xfffffu) + ((u >> 20) & 15u) + ((u >> 24) & 15u) : (uint64_t)u + 1u;
            if (last > desc->n_transforms || (ext && ((u >> 28) != 0u || (((u >> 24) & 15u) != 0u && h.medium_material < 0))))
                return fail(PT_ERR_INVALID_ARG, "hitable %u: transform index / chain out of range", i);
        }
        if (h.medium_material >= 0) {
            if ((uint32_t)h.medium_material >= desc->n_materials || desc->materials[h.medium_material].kind != PT_MAT_ISOTROPIC)
                return fail(PT_ERR_INVALID_ARG, "hitable %u: medium_material must index an Isotropic material", i);
        }
        W.has_media = W.has_media || h.medium_material >= 0;
        if (h.transform >= 0 && ((uint32_t)h.transform >> 20) != 0u && (((uint32_t)h.transform >> 20) & 15u) + (((uint32_t)h.transform >> 24) & 15u) != 1u) W.has_chains = true;
        if (h.transform >= 0 && (((uint32_t)h.transform >> 24) & 15u) != 0u) W.has_chains = true;
        if (h.kind != PT_HIT_SPHERE || h.transform >= 0 || h.medium_material >= 0) W.all_spheres = false;
        if (h.kind > PT_HIT_MOVING_SPHERE || h.transform >= 0 || h.medium_material >= 0) W.sphere_like = false;
    }
    W.ref_depth = 0;
    if (W.is_graph) {   // interpreted (pt_graph.h): the leaves were checked above, the nodes by flatten_world_graph; media and BVHNodes live in the graph
        W.sphere_like = W.all_spheres = false;
        W.has_media = W.has_chains = true;
        return PT_OK;
    }
    if (desc->n_bvh_nodes) {
        if (!desc->bvh_nodes) return fail(PT_ERR_INVALID_ARG, "bvh_nodes is NULL");
        W.ref_depth = bvh_depth_checked(desc->bvh_nodes, desc->n_bvh_nodes, desc->n_hitables, desc->bvh_root);
        if (W.ref_depth == 0) return fail(PT_ERR_INVALID_ARG, "malformed BVH (bad child index or cycle)");
        for (uint32_t i = 0; i < desc->n_bvh_nodes; ++i)   // (the children of a medium group are reached through their group only)
            for (const int32_t c : {desc->bvh_nodes[i].lhs, desc->bvh_nodes[i].rhs})
                if (c < 0 && member[(uint32_t)~c]) return fail(PT_ERR_INVALID_ARG, "BVH node %u: leaf %d is a child of a medium group", i, ~c);
    }
    if (!(W.sphere_like && desc->n_hitables)) {
        W.sphere_like = false;
        return PT_OK;
    }
    W.sph.resize(desc->n_hitables), W.mat.resize(desc->n_hitables), W.motion.resize(desc->n_hitables);
    for (uint32_t i = 0; i < desc->n_hitables; ++i) {
        const pt_hitable &h = desc->hitables[i];
        W.mat[i] = h.material;
        if (h.kind == PT_HIT_SPHERE) {
            W.sph[i] = pt_sphere{h.p[0], h.p[1], h.p[2], h.p[3]};
            W.motion[i] = MotionIn{{0, 0, 0}, 0.f, 0.f, 0u};
        } else {
            W.sph[i] = pt_sphere{h.p[0], h.p[1], h.p[2], h.p[6]};
            W.motion[i] = MotionIn{{h.p[3], h.p[4], h.p[5]}, h.p[7], h.p[8], 1u};
        }
    }
    // Sphere hits have u = v = 0 (sphere.rs:47-48), so an Image texture is one texel for them: i = 0,
    // j = ((1 - 0) * height - 0.001) as i32 = height - 1 (texture.rs:28-33). Fold it into a Constant.
    W.folded.assign(desc->textures, desc->textures + desc->n_textures);
    for (pt_texture &t : W.folded) {
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
    pt_scene_desc &d = W.desc;
    d = pt_scene_desc{};
    d.n_spheres = desc->n_hitables, d.spheres = W.sph.data(), d.sphere_material = W.mat.data();
    d.n_materials = desc->n_materials, d.materials = desc->materials;
    d.n_textures = desc->n_textures, d.textures = W.folded.data(), d.perlin = desc->perlin;
    d.n_bvh_nodes = desc->n_bvh_nodes, d.bvh_nodes = desc->bvh_nodes, d.bvh_root = desc->bvh_root;
    d.has_sky = desc->has_sky;
    memcpy(d.sky, desc->sky, sizeof d.sky);
    return PT_OK;
}

// ---- scene graph -> list form (include/ptgpu.h pt_node) ------------------------------------------------------------------
namespace {
const char *node_kind_name(uint32_t k) {
    static const char *n[] = {"a shape", "a HitableList", "an Instance", "a ConstantMedium", "a BVHNode"};
    return k < 5u ? n[k] : "an unknown node";
}

// A graph that does not flatten is interpreted on the device (pt_graph.h), which trusts every index: all nodes reachable from the
// root are checked here, once each (shared children are NOT expanded: the walk visits a node as often as the reference would, but
// validity is per node), cycles are refused, and so is a nesting deeper than the walk's frames.
int validate_interpreted_graph(const pt_world_desc *d) {
    std::vector<uint8_t> colour(d->n_nodes, 0);     // 0 unseen, 1 on the current path, 2 done
    std::vector<uint32_t> height(d->n_nodes, 0);    // frames the walk needs at and below a node
    struct Item { uint32_t node; size_t next; std::vector<uint32_t> kids; };
    std::vector<Item> stack;
    auto children_of = [&](uint32_t n, std::vector<uint32_t> &out) -> int {
        const pt_node &N = d->nodes[n];
        out.clear();
        switch (N.kind) {
        case PT_NODE_HITABLE: {
            if (N.a >= d->n_hitables) return fail(PT_ERR_INVALID_ARG, "scene graph: node %u: hitable index %u out of range", n, N.a);
            const pt_hitable &h = d->hitables[N.a];
            if (h.transform >= 0 || h.medium_material >= 0)
                return fail(PT_ERR_INVALID_ARG, "scene graph: hitable %u carries its own transform / medium; in a graph these are Instance / ConstantMedium nodes", N.a);
            return PT_OK;
        }
        case PT_NODE_LIST:
            if ((uint64_t)N.a + N.b > d->n_node_children) return fail(PT_ERR_INVALID_ARG, "scene graph: node %u: children [%u, %u) exceed node_children", n, N.a, N.a + N.b);
            for (uint32_t j = 0; j < N.b; ++j) out.push_back(d->node_children[N.a + j]);
            return PT_OK;
        case PT_NODE_INSTANCE:
            if (N.a >= d->n_transforms) return fail(PT_ERR_INVALID_ARG, "scene graph: node %u: transform index %u out of range", n, N.a);
            out.push_back(N.b);
            return PT_OK;
        case PT_NODE_MEDIUM:
            if (N.a >= d->n_materials || d->materials[N.a].kind != PT_MAT_ISOTROPIC)
                return fail(PT_ERR_INVALID_ARG, "scene graph: node %u: a ConstantMedium's material %u must index an Isotropic material", n, N.a);
            out.push_back(N.b);
            return PT_OK;
        case PT_NODE_BVH:
            if (N.a >= d->n_bvh_nodes || !d->bvh_nodes) return fail(PT_ERR_INVALID_ARG, "scene graph: node %u: BVHNode row %u out of range", n, N.a);
            if (d->bvh_nodes[N.a].lhs < 0 || d->bvh_nodes[N.a].rhs < 0)
                return fail(PT_ERR_INVALID_ARG, "scene graph: node %u: the children of a BVHNode in a graph are node indices (>= 0)", n);
            out.push_back((uint32_t)d->bvh_nodes[N.a].lhs), out.push_back((uint32_t)d->bvh_nodes[N.a].rhs);
            return PT_OK;
        default: return fail(PT_ERR_INVALID_ARG, "scene graph: node %u: unknown kind %u", n, N.kind);
        }
    };
    const auto push = [&](uint32_t n) -> int {
        stack.push_back(Item{n, 0, {}});
        colour[n] = 1;
        return children_of(n, stack.back().kids);
    };
    if (int rc = push(d->root_node)) return rc;
    while (!stack.empty()) {
        Item &top = stack.back();
        const uint32_t n = top.node;
        if (top.next < top.kids.size()) {
            const uint32_t c = top.kids[top.next++];
            if (c >= d->n_nodes) return fail(PT_ERR_INVALID_ARG, "scene graph: node index %u out of range (child of node %u)", c, n);
            if (colour[c] == 1) return fail(PT_ERR_UNSUPPORTED, "scene graph: node %u contains itself (a cycle)", c);
            if (colour[c] == 0)
                if (int rc = push(c)) return rc;   // (`top` is stale from here on)
            continue;
        }
        uint32_t h = 0;
        for (uint32_t c : top.kids) h = std::max(h, height[c]);
        height[n] = h + 1u;
        colour[n] = 2;
        stack.pop_back();
    }
    if (height[d->root_node] > ptdev::kGraphDepth)
        return fail(PT_ERR_UNSUPPORTED, "scene graph: %u nested ray_hit calls below the root; the interpreted walk takes %u (a graph that flattens has no such limit)",
                    height[d->root_node], ptdev::kGraphDepth);
    return PT_OK;
}
struct Flattener {
    const pt_world_desc *d;
    FlatWorld *out;
    std::vector<uint32_t> path;    // nodes on the current descent (cycle check)
    std::vector<uint32_t> chain;   // Instance transforms met on the way down, outermost first
    int rc = PT_OK;
    unsigned long long visits = 0;   // nodes walked so far (budget against exponential expansion of shared children)
    bool needs_interpreter = false;  // refused because of a nesting the list form cannot express (not because the graph is malformed)

    bool on_path(uint32_t n) const { return std::find(path.begin(), path.end(), n) != path.end(); }
    int32_t encode_chain(const std::vector<uint32_t> &outer, const std::vector<uint32_t> &inner) {
        if (outer.empty() && inner.empty()) return -1;
        const size_t first = out->xf.size();
        for (uint32_t t : outer) out->xf.push_back(d->transforms[t]);
        for (uint32_t t : inner) out->xf.push_back(d->transforms[t]);
        // (a single inner level is also written in the extended form: a plain index above 2^20 - 1 cannot occur either way)
        return (int32_t)((uint32_t)first | ((uint32_t)inner.size() << 20) | ((uint32_t)outer.size() << 24));
    }
    // the boundary of a ConstantMedium: Instance* (shape); collects the Instances into `inner` and returns the leaf
    int boundary(uint32_t medium_node, uint32_t n, std::vector<uint32_t> &inner, uint32_t *leaf) {
        for (uint32_t guard = 0; guard <= d->n_nodes; ++guard) {
            if (n >= d->n_nodes) return fail(PT_ERR_INVALID_ARG, "scene graph: node index %u out of range", n);
            const pt_node &N = d->nodes[n];
            if (N.kind == PT_NODE_HITABLE || N.kind == PT_NODE_LIST) {   // (a List: the caller emits a medium group, emit_group)
                *leaf = n;
                return PT_OK;
            }
            if (N.kind == PT_NODE_MEDIUM || N.kind == PT_NODE_BVH) needs_interpreter = true;
            if (N.kind != PT_NODE_INSTANCE)
                return fail(PT_ERR_UNSUPPORTED, "scene graph: node %u is a ConstantMedium whose boundary contains %s (node %u); only Instance levels around one shape can bound a medium",
                            medium_node, node_kind_name(N.kind), n);
            if (N.a >= d->n_transforms) return fail(PT_ERR_INVALID_ARG, "scene graph: node %u: transform index %u out of range", n, N.a);
            inner.push_back(N.a);
            n = N.b;
        }
        return fail(PT_ERR_UNSUPPORTED, "scene graph: node %u is a ConstantMedium whose boundary never reaches a shape (a cycle of Instances)", medium_node);
    }
    // A ConstantMedium whose boundary is Instance* (HitableList of Instance* (shape)): one PT_HIT_MEDIUM_GROUP entry (the medium and the
    // Instances around it) + one entry per child, each under `between` (the Instances between the medium and the list) followed by its own.
    // Returns PT_ERR_UNSUPPORTED with needs_interpreter set when a child is anything else (a list, a medium, a BVHNode) or a chain is too long.
    int emit_group(uint32_t medium_node, uint32_t list_node, const std::vector<uint32_t> &outer, const std::vector<uint32_t> &between, int32_t medium_material, float density) {
        const pt_node &L = d->nodes[list_node];
        if ((uint64_t)L.a + L.b > d->n_node_children) return fail(PT_ERR_INVALID_ARG, "scene graph: node %u: children [%u, %u) exceed node_children", list_node, L.a, L.a + L.b);
        if (L.b == 0) {   // an empty boundary never answers: the medium is never hit (constant_medium.rs:39-40) -- nothing to emit
            return PT_OK;
        }
        if (outer.size() > 15 || between.size() > 15) {
            needs_interpreter = true;
            return fail(PT_ERR_UNSUPPORTED, "scene graph: node %u sits below %zu Instance levels on one side of a medium; at most 15 are supported", medium_node, outer.size());
        }
        struct Child { uint32_t leaf; std::vector<uint32_t> chain; };
        std::vector<Child> kids;
        for (uint32_t j = 0; j < L.b; ++j) {
            uint32_t n = d->node_children[L.a + j];
            Child c{0u, between};
            for (uint32_t guard = 0;; ++guard) {
                if (n >= d->n_nodes) return fail(PT_ERR_INVALID_ARG, "scene graph: node index %u out of range (child of node %u)", n, list_node);
                const pt_node &N = d->nodes[n];
                if (N.kind == PT_NODE_HITABLE) break;
                if (N.kind != PT_NODE_INSTANCE || guard > d->n_nodes || c.chain.size() >= 15) {
                    needs_interpreter = true;   // (a List, a medium or a BVHNode inside the boundary, a cycle, or more than 15 levels: the interpreter's case -- or its refusal)
                    return fail(PT_ERR_UNSUPPORTED, "scene graph: node %u is a ConstantMedium whose boundary list (node %u) holds %s (node %u)", medium_node, list_node,
                                node_kind_name(N.kind), n);
                }
                if (N.a >= d->n_transforms) return fail(PT_ERR_INVALID_ARG, "scene graph: node %u: transform index %u out of range", n, N.a);
                c.chain.push_back(N.a);
                n = N.b;
            }
            c.leaf = n;
            kids.push_back(std::move(c));
        }
        if (out->hit.size() + kids.size() + 1 >= (1u << 20)) return fail(PT_ERR_UNSUPPORTED, "scene graph: flattening yields more than 2^20 list entries (an Instance around a HitableList is distributed over its children)");
        pt_hitable g{};
        g.kind = PT_HIT_MEDIUM_GROUP;
        g.material = 0;
        g.transform = encode_chain(outer, {});
        g.medium_material = medium_material;
        g.density = density;
        const uint32_t n_members = (uint32_t)kids.size();
        memcpy(&g.p[0], &n_members, 4);
        out->hit.push_back(g);
        for (const Child &c : kids) {
            const pt_node &Lf = d->nodes[c.leaf];
            if (Lf.a >= d->n_hitables) return fail(PT_ERR_INVALID_ARG, "scene graph: node %u: hitable index %u out of range", c.leaf, Lf.a);
            pt_hitable h = d->hitables[Lf.a];
            if (h.transform >= 0 || h.medium_material >= 0)
                return fail(PT_ERR_INVALID_ARG, "scene graph: hitable %u carries its own transform / medium; in a graph these are Instance / ConstantMedium nodes", Lf.a);
            h.transform = encode_chain({}, c.chain);
            h.medium_material = -1;
            if (out->xf.size() >= (1u << 20)) return fail(PT_ERR_UNSUPPORTED, "scene graph: flattening needs more than 2^20 transform slots");
            out->hit.push_back(h);
        }
        return PT_OK;
    }
    int emit(uint32_t leaf_node, const std::vector<uint32_t> &outer, const std::vector<uint32_t> &inner, int32_t medium_material, float density) {
        const pt_node &L = d->nodes[leaf_node];
        if (L.a >= d->n_hitables) return fail(PT_ERR_INVALID_ARG, "scene graph: node %u: hitable index %u out of range", leaf_node, L.a);
        pt_hitable h = d->hitables[L.a];
        if (h.transform >= 0 || h.medium_material >= 0)
            return fail(PT_ERR_INVALID_ARG, "scene graph: hitable %u carries its own transform / medium; in a graph these are Instance / ConstantMedium nodes", L.a);
        if (outer.size() > 15 || inner.size() > 15)
            return fail(PT_ERR_UNSUPPORTED, "scene graph: node %u sits below %zu Instance levels on one side of a medium; at most 15 are supported", leaf_node,
                        std::max(outer.size(), inner.size()));
        if (out->hit.size() >= (1u << 20)) return fail(PT_ERR_UNSUPPORTED, "scene graph: flattening yields more than 2^20 list entries (an Instance around a HitableList is distributed over its children)");
        h.transform = encode_chain(outer, inner);
        if (out->xf.size() >= (1u << 20)) return fail(PT_ERR_UNSUPPORTED, "scene graph: flattening needs more than 2^20 transform slots");
        h.medium_material = medium_material;
        h.density = density;
        out->hit.push_back(h);
        return PT_OK;
    }
    int walk(uint32_t n) {
        if (n >= d->n_nodes) return fail(PT_ERR_INVALID_ARG, "scene graph: node index %u out of range", n);
        if (on_path(n)) return fail(PT_ERR_UNSUPPORTED, "scene graph: node %u contains itself (a cycle)", n);
        if (path.size() > 256) return fail(PT_ERR_UNSUPPORTED, "scene graph: deeper than 256 levels at node %u", n);
        // shared children are expanded once per path: a DAG of nested lists can cost exponential time on caller-supplied input
        if (++visits > (1ull << 22)) return fail(PT_ERR_UNSUPPORTED, "scene graph: flattening visits more than 2^22 nodes (shared children are expanded per path)");
        const pt_node &N = d->nodes[n];
        path.push_back(n);
        int r = PT_OK;
        switch (N.kind) {
        case PT_NODE_HITABLE: r = emit(n, {}, chain, -1, 0.f); break;
        case PT_NODE_LIST:
            if ((uint64_t)N.a + N.b > d->n_node_children) r = fail(PT_ERR_INVALID_ARG, "scene graph: node %u: children [%u, %u) exceed node_children", n, N.a, N.a + N.b);
            for (uint32_t j = 0; j < N.b && r == PT_OK; ++j) r = walk(d->node_children[N.a + j]);
            break;
        case PT_NODE_INSTANCE:
            if (N.a >= d->n_transforms) {
                r = fail(PT_ERR_INVALID_ARG, "scene graph: node %u: transform index %u out of range", n, N.a);
                break;
            }
            chain.push_back(N.a);
            r = walk(N.b);
            chain.pop_back();
            break;
        case PT_NODE_MEDIUM: {
            std::vector<uint32_t> inner;
            uint32_t leaf = 0;
            // (checked here, not after flattening: as int32 an index >= 2^31 would read as "no medium" and render the boundary solid)
            if (N.a >= d->n_materials || d->materials[N.a].kind != PT_MAT_ISOTROPIC) {
                r = fail(PT_ERR_INVALID_ARG, "scene graph: node %u: a ConstantMedium's material %u must index an Isotropic material", n, N.a);
                break;
            }
            if ((r = boundary(n, N.b, inner, &leaf)) == PT_OK)
                r = d->nodes[leaf].kind == PT_NODE_LIST ? emit_group(n, leaf, chain, inner, (int32_t)N.a, N.density) : emit(leaf, chain, inner, (int32_t)N.a, N.density);
            break;
        }
        case PT_NODE_BVH:
            needs_interpreter = true;
            r = fail(PT_ERR_UNSUPPORTED, "scene graph: node %u is a BVHNode below the root", n);
            break;
        default: r = fail(PT_ERR_INVALID_ARG, "scene graph: node %u: unknown kind %u", n, N.kind);
        }
        path.pop_back();
        return r;
    }
};
}  // namespace

int flatten_world_graph(const pt_world_desc *desc, FlatWorld &out, const pt_world_desc **use) {
    *use = desc;
    if (!desc || desc->n_nodes == 0) return PT_OK;
    if (!desc->nodes || (desc->n_node_children && !desc->node_children)) return fail(PT_ERR_INVALID_ARG, "scene graph: nodes / node_children is NULL");
    if (desc->n_hitables && !desc->hitables) return fail(PT_ERR_INVALID_ARG, "hitables is NULL");
    if (desc->n_transforms && !desc->transforms) return fail(PT_ERR_INVALID_ARG, "transforms is NULL");
    if (desc->n_materials && !desc->materials) return fail(PT_ERR_INVALID_ARG, "materials is NULL");   // (a ConstantMedium node looks its phase function up during the walk: found by tools/fuzz_desc.cpp)
    if (desc->root_node >= desc->n_nodes) return fail(PT_ERR_INVALID_ARG, "scene graph: root node %u out of range", desc->root_node);
    Flattener F{desc, &out, {}, {}};
    // a nesting the list form cannot express: the graph is validated and handed on as it is, to be interpreted (pt_graph.h)
    const auto interpret_instead = [&](int rc) -> int {
        if (rc != PT_ERR_UNSUPPORTED || !F.needs_interpreter) return rc;
        if (int v = validate_interpreted_graph(desc)) return v;
        out.hit.clear(), out.xf.clear();
        out.flat = *desc;
        out.interpreted = true;
        *use = &out.flat;
        return PT_OK;
    };
    std::vector<uint32_t> first_of_child;   // list entries each child of the root list starts at (BVH leaves index the root's children)
    const pt_node &R = desc->nodes[desc->root_node];
    if (R.kind == PT_NODE_LIST) {
        if ((uint64_t)R.a + R.b > desc->n_node_children) return fail(PT_ERR_INVALID_ARG, "scene graph: root list exceeds node_children");
        F.path.push_back(desc->root_node);
        for (uint32_t j = 0; j < R.b; ++j) {
            first_of_child.push_back((uint32_t)out.hit.size());
            if (int rc = F.walk(desc->node_children[R.a + j])) return interpret_instead(rc);
        }
    } else {
        first_of_child.push_back(0u);
        if (int rc = F.walk(desc->root_node)) return interpret_instead(rc);
    }
    first_of_child.push_back((uint32_t)out.hit.size());
    out.flat = *desc;
    out.flat.n_nodes = 0, out.flat.nodes = nullptr, out.flat.n_node_children = 0, out.flat.node_children = nullptr, out.flat.root_node = 0;
    out.flat.n_hitables = (uint32_t)out.hit.size(), out.flat.hitables = out.hit.data();
    out.flat.n_transforms = (uint32_t)out.xf.size(), out.flat.transforms = out.xf.data();
    if (desc->n_bvh_nodes) {
        if (!desc->bvh_nodes) return fail(PT_ERR_INVALID_ARG, "bvh_nodes is NULL");
        // BVH leaves index the root list's CHILDREN; each must be one list entry -- or one medium group, whose first entry then stands for it
        const size_t n_children = first_of_child.size() - 1;
        for (size_t c = 0; c < n_children; ++c) {
            const uint32_t span = first_of_child[c + 1] - first_of_child[c];
            uint32_t members = 0;
            if (span >= 2u && out.hit[first_of_child[c]].kind == PT_HIT_MEDIUM_GROUP) memcpy(&members, &out.hit[first_of_child[c]].p[0], 4);
            if (span != 1u && span != members + 1u)
                return fail(PT_ERR_UNSUPPORTED, "scene graph: BVH leaves index the root list's children, and child %zu flattens to %u list entries instead of one", c, span);
        }
        out.bvh.assign(desc->bvh_nodes, desc->bvh_nodes + desc->n_bvh_nodes);
        for (pt_bvh_node &nd : out.bvh)
            for (int32_t *c : {&nd.lhs, &nd.rhs})
                if (*c < 0) {
                    const uint32_t child = (uint32_t)~*c;
                    if (child >= n_children) return fail(PT_ERR_INVALID_ARG, "malformed BVH (bad child index or cycle)");
                    *c = ~(int32_t)first_of_child[child];
                }
        out.flat.bvh_nodes = out.bvh.data();
    }
    *use = &out.flat;
    return PT_OK;
}

void world_traits(const pt_world_desc *desc, const WorldAsSpheres &w, ptsel::SceneTraits &tr) {
    tr.is_world = true;
    tr.has_media = w.has_media;
    tr.has_image = w.has_image;
    tr.has_chains = w.has_chains;
    tr.has_noise = w.has_noise;
    tr.n_hitables = desc->n_hitables;
    tr.n_world_xf = desc->n_transforms;
    tr.ref_bvh_depth = w.ref_depth;
    tr.has_caller_bvh = desc->n_bvh_nodes != 0 && !w.is_graph;   // (an interpreted graph's BVHNode rows are part of the graph: no `-B` over it)
    tr.is_graph = w.is_graph;
    tr.atts_finite = true;
    for (uint32_t i = 0; i < desc->n_materials; ++i)
        for (int c = 0; c < 3; ++c) tr.atts_finite = tr.atts_finite && std::isfinite(desc->materials[i].albedo[c]);
    // (pt_scene.hip folds a Lambertian / Isotropic material's Constant texture into its record, and the kernel's colour() returns that
    //  colour without looking at it: an infinite Constant colour anywhere switches the shortcut off for the whole world as well)
    for (uint32_t i = 0; i < desc->n_textures; ++i)
        if (desc->textures[i].kind == PT_TEX_CONSTANT)
            for (int c = 0; c < 3; ++c) tr.atts_finite = tr.atts_finite && std::isfinite(desc->textures[i].color[c]);
    // texture.rs:86-88: 0.5 (1 + sin(scale p.z + 10 turb(p))) is finite where its argument is (pt_args.h kLazyNoise*: bounds on the scale
    // and on the Perlin gradients here, on the point in the kernel)
    tr.noise_finite = tr.atts_finite && (!w.has_noise || desc->perlin != nullptr);
    for (uint32_t i = 0; i < desc->n_textures; ++i)
        if (desc->textures[i].kind == PT_TEX_NOISE) tr.noise_finite = tr.noise_finite && std::fabs(desc->textures[i].scale) <= ptdev::kLazyNoiseScale;
    if (desc->perlin)
        for (int i = 0; i < 256; ++i)
            for (int c = 0; c < 3; ++c) tr.noise_finite = tr.noise_finite && std::fabs(desc->perlin->randvec[i][c]) <= ptdev::kLazyNoiseGradient;
}

}  // namespace pthostside
