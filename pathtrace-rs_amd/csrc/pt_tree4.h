// pt_tree4.h -- the 4-wide internal tree of the tree kernels: node layout and the build rules shared by the DEVICE
// builder (pt_build.hip) and its host restatement (ptgpu.hip, used as the reference the device build is tested against
// and as the fallback when no sort scratch can be allocated).
//
// The tree replaces, for TRAVERSAL only, the caller's BVHNode tree (bvh.rs:64-94,268-333: random split axis, median
// split): results never depend on it (closest hit by (t, tie-break)). Build rule, level by level:
// a segment of n >= 2 spheres gets c = ceil(n / m) children of (nearly) equal size, m = the smallest power of 4 with
// 4 m >= n; the segment is first ordered along the longest axis of its centroid bounds and cut after the first
// ceil(c / 2) children, then each side holding two children is ordered along ITS longest axis and cut again. Ordering
// is a STABLE sort by the coordinate (ties keep the order they had), which is what one global stable radix sort per
// phase does on the device. A child with one sphere is a leaf slot; inner children come first in the node and get
// consecutive node indices in level order.
#pragma once
#include <stdint.h>
#include <string.h>

namespace ptdev {

#if defined(__HIPCC__)
#define PT_HD __host__ __device__ inline
#else
#define PT_HD inline
#endif

struct DNode4 {  // 128 B
    float lo[3][4];      // child boxes, plane arrays: lo[axis][child]
    float hi[3][4];
    int32_t child[4];    // inner children first: consecutive nodes, child[j] = child[0] + j (>= 0); then leaves (~sphere < 0); kNoChild4 = empty
    float cx, cy, cz;    // centre of the node's own box
    uint32_t padh;       // two f16 (rounded up): lo = 6e-6 / r_min below the node, hi = that * |h_node|^2 + 1e-4
};
constexpr int32_t kNoChild4 = 0x7fffffff;

// One sphere as the builders see it: centre used for the splits and the box of its whole sweep (moving spheres).
struct TreeItem {
    uint32_t sphere;
    float c[3], mn[3], mx[3], r;
};

// Box + smallest radius of a subtree (what a parent needs from a child).
struct TreeBox {
    float mn[3], mx[3], rmin;
};

// f32 -> u32 whose unsigned order is the float order (-0 < +0, NaNs at the ends; the builders never see NaN centres)
PT_HD uint32_t tree_orderable(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Children of a segment of n >= 2 spheres: count c (2..4), cut[j] = offset of child j inside the segment, and how many
// children lie on the low side of the first cut.
struct TreePlan {
    uint32_t c, half, cut[5];
};
PT_HD TreePlan tree_plan(uint32_t n) {
    uint64_t m = 1;
    while (4 * m < n) m *= 4;
    TreePlan p;
    p.c = (uint32_t)((n + m - 1) / m);
    for (uint32_t j = 0; j <= 4; ++j) p.cut[j] = j <= p.c ? (uint32_t)((uint64_t)n * j / p.c) : n;
    p.half = (p.c + 1) / 2;
    return p;
}

// axis a segment is ordered along: the longest extent of its centroid bounds (max - min per axis, f32), the lower axis on ties
PT_HD int tree_axis_of_extents(float ex, float ey, float ez) {
    int axis = 0;
    float best = ex;
    if (ey > best) axis = 1, best = ey;
    if (ez > best) axis = 2;
    return axis;
}

// smallest f16 >= v (0 < v far below the f16 limit), independent of how the first conversion rounds
PT_HD double tree_f16_value(uint16_t bits) { return (double)(float)__builtin_bit_cast(_Float16, bits); }
PT_HD uint16_t tree_f16_up(double v) {
    uint16_t u = __builtin_bit_cast(uint16_t, (_Float16)(float)v);
    while (tree_f16_value(u) < v) ++u;
    while (u > 0 && tree_f16_value((uint16_t)(u - 1)) >= v) --u;
    return u;
}

// Finishes node `w` once its child slots (boxes, refs) are filled: the node's own box, its centre and the pad constants
// of the conservative box test -- >= 4x the bound on how far the reference's f32 discriminant inflates a sphere
// (0.65e-6 (|o - c|^2 + r^2) / r), with |o - c_sphere|^2 <= 2 (|o - c_node|^2 + |h_node|^2); stored as two f16 rounded
// UP: k = 6e-6 / r_min and k |h|^2 + 1e-4 (+ f32 rounding of the centre). Returns the node's box for its parent.
PT_HD TreeBox tree_finish_node(DNode4 &w, const TreeBox ch[4], uint32_t c) {
    TreeBox out;
    for (int k = 0; k < 3; ++k) out.mn[k] = 3e38f, out.mx[k] = -3e38f;
    out.rmin = 3e38f;
    for (uint32_t j = 0; j < 4; ++j) {
        if (j >= c) {   // empty slot: never hit (and checked by its ref)
            for (int k = 0; k < 3; ++k) w.lo[k][j] = 3.0e38f, w.hi[k][j] = -3.0e38f;
            w.child[j] = kNoChild4;
            continue;
        }
        for (int k = 0; k < 3; ++k) {
            w.lo[k][j] = ch[j].mn[k], w.hi[k][j] = ch[j].mx[k];
            out.mn[k] = ch[j].mn[k] < out.mn[k] ? ch[j].mn[k] : out.mn[k];
            out.mx[k] = ch[j].mx[k] > out.mx[k] ? ch[j].mx[k] : out.mx[k];
        }
        out.rmin = ch[j].rmin < out.rmin ? ch[j].rmin : out.rmin;
    }
    double h2 = 0.0, cmag = 0.0;
    float ctr[3];
    for (int k = 0; k < 3; ++k) {
        ctr[k] = (float)(0.5 * ((double)out.mn[k] + (double)out.mx[k]));
        const double ha = (double)out.mx[k] - (double)ctr[k], hb = (double)ctr[k] - (double)out.mn[k];
        const double h = ha > hb ? ha : hb;
        h2 += h * h, cmag += ctr[k] < 0.0f ? -(double)ctr[k] : (double)ctr[k];
    }
    const double pk = 6.0e-6 / (double)out.rmin;
    w.cx = ctr[0], w.cy = ctr[1], w.cz = ctr[2];
    w.padh = (uint32_t)tree_f16_up(pk * 1.001) | ((uint32_t)tree_f16_up((pk * h2 + 1.0e-4 + 1.0e-6 * cmag) * 1.001) << 16);
    return out;
}

// ---- the node the tree kernels READ: 64 bytes -------------------------------------------------------------------
// A divergent 16-byte load costs one tag lookup per lane in the vector L1 whatever it returns, and a visit of the 128-byte
// DNode4 takes eight of them per lane: the tree kernel was bound by exactly that (profiles/r02_c5: TA / TD ~85 % busy; four
// MORE loads per visit cost +38 % of the frame). The packed node holds the same four child boxes in FOUR loads:
//   * planes as f16 OFFSETS from the node's own min corner (`origin`, f32), the lower planes rounded down and the upper
//     ones up, so the decoded box contains the DNode4 box (at most extent / 1024 larger per side); the kernel feeds
//     them to mixed-precision FMAs (v_fma_mix_f32: f16 plane x f32 1/d + f32 constant), no conversion instructions;
//   * no child references: inner children come first and are consecutive nodes (base + slot), leaves follow, and a leaf's
//     sphere is found through its (node, slot) number in the slot records built beside the nodes (PtSlotRec);
//   * the two constants of the conservative pad as powers of two (rounded up), and the pad measured from `origin`
//     with the node's full extent E: every sphere below lies within |origin - o| + |E| of the ray origin.
// meta = first inner child (16 bits) | inner children (3) | children (3) | e_k (5): k = 2^(e_k - 31) | e_0 (5): p0 = 2^(e_0 - 14).
struct DNode4Q {
    uint16_t plane[3][2][4];   // [axis][0 = lower plane, 1 = upper plane][child]
    float origin[3];
    uint32_t meta;
};
static_assert(sizeof(DNode4Q) == 64, "packed node is one half cache line");

// greatest f16 <= v (v >= 0; anything above the f16 range gives the largest finite f16)
PT_HD uint16_t tree_f16_down(double v) {
    if (!(v > 0.0)) return 0;
    if (v >= 65504.0) return 0x7bffu;
    uint16_t u = __builtin_bit_cast(uint16_t, (_Float16)(float)v);
    if (u > 0x7bffu) u = 0x7bffu;
    while (u > 0 && tree_f16_value(u) > v) --u;
    while (u < 0x7bffu && tree_f16_value((uint16_t)(u + 1)) <= v) ++u;
    return u;
}
// smallest f16 >= v including +inf (v >= 0)
PT_HD uint16_t tree_f16_ceil(double v) {
    if (!(v > 0.0)) return 0;
    if (v > 65504.0) return 0x7c00u;
    return tree_f16_up(v);
}
// smallest e with 2^(e - bias) >= v, or 32 when none in 0..31 does
PT_HD uint32_t tree_pow2_up(double v, int bias) {
    for (uint32_t e = 0; e < 32u; ++e) {
        double p = 1.0;
        for (int i = (int)e - bias; i > 0; --i) p *= 2.0;
        for (int i = (int)e - bias; i < 0; ++i) p *= 0.5;
        if (p >= v) return e;
    }
    return 32u;
}

// Packs one finished DNode4. Returns false when the node cannot be expressed (pad constants out of range, node numbers
// beyond 16 bits): the scene then walks the binary tree instead.
PT_HD bool tree_pack_node(const DNode4 &w, DNode4Q &q) {
    uint32_t nc = 0, ni = 0;
    for (int j = 0; j < 4; ++j) {
        if (w.child[j] == kNoChild4) break;
        ++nc;
        if (w.child[j] >= 0) ++ni;
    }
    const uint32_t base = ni ? (uint32_t)w.child[0] : 0u;
    bool ok = base + 3u < 65536u;
    double e2 = 0.0, mag = 0.0;
    for (int k = 0; k < 3; ++k) {
        float mn = 3.0e38f, mx = -3.0e38f;
        for (uint32_t j = 0; j < nc; ++j) {
            mn = w.lo[k][j] < mn ? w.lo[k][j] : mn;
            mx = w.hi[k][j] > mx ? w.hi[k][j] : mx;
        }
        const bool sane = nc > 0 && mn <= mx && mn > -3.0e38f && mx < 3.0e38f;
        // (a NaN origin makes every plane distance NaN, which the visit counts as a hit: conservative)
        q.origin[k] = sane ? mn : __builtin_nanf("");
        for (uint32_t j = 0; j < 4u; ++j) {
            const bool used = sane && j < nc;
            // (an empty slot -- or every slot of a node with a NaN corner would be "hit": such nodes have none -- must never be:
            //  lower plane at the f16 maximum above an upper plane at zero makes far < near on every axis the ray moves along)
            q.plane[k][0][j] = used ? tree_f16_down((double)w.lo[k][j] - (double)mn) : (uint16_t)(j < nc ? 0u : 0x7bffu);
            q.plane[k][1][j] = used ? tree_f16_ceil((double)w.hi[k][j] - (double)mn) : (uint16_t)0;
        }
        const double ext = sane ? (double)mx - (double)mn : 0.0;
        e2 += ext * ext, mag += ext + (sane ? (mn < 0.0f ? -(double)mn : (double)mn) : 0.0);
    }
    const double pk = tree_f16_value((uint16_t)(w.padh & 0xffffu));   // 6e-6 / r_min (rounded up by tree_finish_node)
    const uint32_t ek = tree_pow2_up(pk, 31);
    ok = ok && ek < 32u;
    double pk2 = 1.0;
    for (uint32_t i = ek; i < 31u; ++i) pk2 *= 0.5;
    const uint32_t e0 = tree_pow2_up((pk2 * e2 + 1.0e-4 + 2.0e-6 * mag) * 1.001, 14);
    ok = ok && e0 < 32u;
    q.meta = (base & 0xffffu) | (ni << 16) | (nc << 19) | ((ek & 31u) << 22) | ((e0 & 31u) << 27);
    return ok;
}

}  // namespace ptdev
