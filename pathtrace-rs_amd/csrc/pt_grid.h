// pt_grid.h -- closest hit over a UNIFORM CELL GRID: what the 4-wide tree kernels walk instead of the tree when the scene has one
// (pt_host.h GridPlan: many spheres of similar size, spread evenly enough; KArgs::grid_cells != nullptr).
//
// Why. A visit of the 4-wide tree tests FOUR BOXES (146 VALU instructions) and a ray of BASELINE config 5 -- 10 000 spheres of radius 0.2
// on a 100 x 100 lattice -- needs 10.4 of them plus 3.4 exact sphere tests. A cell of the grid holds the spheres themselves: a visit is
// one 80-byte record (five 16-byte loads: up to four spheres and their list indices), four reference discriminants (sphere.rs:33-37,
// the reference's own operation order: the filter is EXACT, a sphere with a discriminant <= 0 does nothing in the reference either) and
// one step of a 3D-DDA; the few spheres with a positive discriminant go through sphere.rs:38-64 and the accept rule right there, so the
// lane's nearest hit -- its walk's limit -- is exact after every cell.
//
// What makes it the same closest hit. The winner is the (t, tie-break) minimum over every sphere whose reference test accepts the ray
// (DESIGN.md 4.2 / 4.4), so the structure only has to PRESENT every such sphere before the walk ends:
//   * a sphere is registered in every cell its box overlaps, the box padded by how far the reference's f32 discriminant can inflate the
//     sphere for a ray whose origin lies within KArgs::grid_d_build of it (pt_tree4.h: 0.65e-6 (|o - c|^2 + r^2) / r; the plan uses 1e-6)
//     and by h / 1000 for the walk's own rounding (cell boundaries are recomputed from integer cell numbers, never accumulated);
//   * the walk visits every cell the line passes through from its entry into the grid's box to the first cell whose entry parameter lies
//     beyond best * 1.0005 + 5e-4 (the tree's culling slack: the reference's root can fall below the true parameter);
//   * a ray from FARTHER away (|o - centre| + half diagonal > grid_d_build: the reference's discriminant is coarser than the registration
//     assumed) walks 27 parallel lines, its own and those displaced by -+ its own inflation bound per axis: a point within that bound of
//     a registered box lies, per axis, in the cell of one of the three displaced points (the bound is below the cell size; beyond that the
//     ray looks at every record). Rare by construction: a camera framing the spheres sits at two half diagonals.
// Large spheres (KArgs::grid_large) are tested for every ray first, as the tree kernels do with theirs.
#pragma once

namespace ptdev {

constexpr uint32_t kGridLinkBit = 0x80000000u;

template <bool MOVING, bool COUNT>
__device__ __forceinline__ void grid_trace(const KArgs &A, unsigned long long *w_keys, f3 o, f3 d, const DivA &av, float time, bool start, Steal4 &cnt) {
    const uint32_t lane = threadIdx.x & 63u;
    const float a = av.a;
    float best = kMaxT;
    int idx = -1;
    uint32_t rank = 0;
    const f3 rcp = mk3(recip_exact(d.x), recip_exact(d.y), recip_exact(d.z));   // ray.rs:14 (read by the gates of a BVH world)
    if (start) {
        for (uint32_t j = 0; j < A.n_grid_large; ++j) {
            const int k = (int)A.grid_large[j];
            bvh_leaf(A, k, sphere_at<MOVING>(A, k, A.spheres[k], time), o, d, rcp, av, best, idx, rank);   // (`rank` is only used among these)
        }
    }
    float limit = trav4_limit(idx < 0 ? kMaxT : best);
#ifdef PT_GRID_ROUNDS
    if (COUNT && lane == 0u) cnt.leaves += 1u;
#endif

    // one record: the reference's discriminant of its (up to) four spheres, then sphere.rs:38-64 + the accept rule for the positive ones.
    // Returns the record's last word (a list index or a link).
#ifdef PT_GRID_ROUNDS
    uint32_t dbg_visits = 0u, dbg_rounds = 0u;
#endif
    const auto visit = [&](uint32_t rec) -> uint32_t {
#ifdef PT_GRID_ROUNDS
        dbg_visits += 1u;
#endif
        const uint4 *cp = A.grid_cells + (size_t)rec * 5u;
        uint4 q0 = cp[0], q1 = cp[1], q2 = cp[2], q3 = cp[3], m = cp[4];
        // (all five loads in flight at once: left alone, the scheduler issues them one by one, each behind the arithmetic of the one before --
        //  four round trips to the L2 per record instead of one -- to save the sixteen registers this takes)
        asm volatile("" : "+v"(q0.x), "+v"(q0.y), "+v"(q0.z), "+v"(q0.w), "+v"(q1.x), "+v"(q1.y), "+v"(q1.z), "+v"(q1.w), "+v"(q2.x), "+v"(q2.y), "+v"(q2.z), "+v"(q2.w),
                     "+v"(q3.x), "+v"(q3.y), "+v"(q3.z), "+v"(q3.w), "+v"(m.x), "+v"(m.y), "+v"(m.z), "+v"(m.w));
#ifndef PT_GRID_ROUNDS
        if (COUNT) cnt.visits += 1u;
#endif
        float bj[4], dj[4];
        uint32_t pos = 0u;
        const uint4 qs[4] = {q0, q1, q2, q3};
        const uint32_t ks[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 c = make_float4(__uint_as_float(qs[j].x), __uint_as_float(qs[j].y), __uint_as_float(qs[j].z), __uint_as_float(qs[j].w));
            if (MOVING) c = sphere_at<MOVING>(A, (int)(ks[j] < A.n_spheres ? ks[j] : 0u), c, time);
            const float ocx = o.x - c.x, ocy = o.y - c.y, ocz = o.z - c.z;
            const float b = (ocx * d.x + ocy * d.y) + ocz * d.z;
            const float cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - c.w * c.w;
            const float disc = b * b - a * cc;
            bj[j] = b, dj[j] = disc;
            pos |= (disc > 0.0f && ks[j] < A.n_spheres) ? (1u << j) : 0u;
        }
        while (wave_any(pos != 0u)) {
            if (pos != 0u) {
                const uint32_t j = (uint32_t)__builtin_ctz(pos);
                pos &= pos - 1u;
                const float b = j == 0u ? bj[0] : (j == 1u ? bj[1] : (j == 2u ? bj[2] : bj[3]));
                const float disc = j == 0u ? dj[0] : (j == 1u ? dj[1] : (j == 2u ? dj[2] : dj[3]));
                const uint32_t k = j == 0u ? ks[0] : (j == 1u ? ks[1] : (j == 2u ? ks[2] : ks[3]));
#ifndef PT_GRID_ROUNDS
                if (COUNT) cnt.leaves += 1u;
#endif
                const float t = sphere_hit_t(av, b, disc, true);
                if (t < kMaxT) {
                    // equal t: the DFS-last leaf in a BVH world (bvh.rs:47-53), the lower list index in a list world (hitable_list.rs:48). The
                    // ranks are only fetched for such a tie (a dependent load per candidate otherwise) and once for the winner at the end.
                    bool better = idx < 0 || t < best;
                    if (!better && t == best && (int)k != idx) better = A.gate ? (A.leaf_rank[k] > A.leaf_rank[idx]) : ((int)k < idx);
                    if (better && (!A.gate || gate_pass(A, (int)k, o, rcp))) best = t, idx = (int)k;
                }
            }
        }
        limit = trav4_limit(idx < 0 ? kMaxT : best);
        return m.w;
    };

    // ---- how many lines this ray walks
    const float gx = o.x - A.grid_centre[0], gy = o.y - A.grid_centre[1], gz = o.z - A.grid_centre[2];
    const float dist = __builtin_sqrtf((gx * gx + gy * gy) + gz * gz) * 1.001f + A.grid_half_diag;
    const bool far = start && !(dist <= A.grid_d_build);
    const float delta = far ? A.grid_infl * (dist * dist + A.grid_rmax2) * 1.001f : 0.0f;
    const bool full = far && !(delta <= A.grid_h);   // (also a NaN origin)
    const uint32_t my_lines = !start || full ? 0u : (far ? 27u : 1u);
    const uint32_t n_lines = wave_any(my_lines > 1u) ? 27u : 1u;
    const float h = A.grid_h, inv_h = A.grid_inv_h;
    const int nx = (int)A.grid_n[0], ny = (int)A.grid_n[1], nz = (int)A.grid_n[2];
    const bool par_x = __builtin_fabsf(d.x) < 1.0e-30f, par_y = __builtin_fabsf(d.y) < 1.0e-30f, par_z = __builtin_fabsf(d.z) < 1.0e-30f;
    const float rx = par_x ? 0.0f : rcp.x, ry = par_y ? 0.0f : rcp.y, rz = par_z ? 0.0f : rcp.z;
    const int sx = d.x >= 0.0f ? 1 : -1, sy = d.y >= 0.0f ? 1 : -1, sz = d.z >= 0.0f ? 1 : -1;
    const float kInf = __builtin_inff();
    for (uint32_t line = 0; line < n_lines; ++line) {
        const uint32_t l3 = (line + 13u) % 27u;
        const float ox = o.x + delta * (float)((int)(l3 % 3u) - 1), oy = o.y + delta * (float)((int)((l3 / 3u) % 3u) - 1), oz = o.z + delta * (float)((int)(l3 / 9u) - 1);
        // clip against the grid's box
        const float lox = A.grid_min[0], loy = A.grid_min[1], loz = A.grid_min[2];
        const float hix = lox + (float)nx * h, hiy = loy + (float)ny * h, hiz = loz + (float)nz * h;
        const float ax0 = (lox - ox) * rx, ax1 = (hix - ox) * rx, ay0 = (loy - oy) * ry, ay1 = (hiy - oy) * ry, az0 = (loz - oz) * rz, az1 = (hiz - oz) * rz;
        const bool in_x = ox >= lox && ox <= hix, in_y = oy >= loy && oy <= hiy, in_z = oz >= loz && oz <= hiz;
        const float nrx = par_x ? (in_x ? -kInf : kInf) : __builtin_fminf(ax0, ax1), frx = par_x ? (in_x ? kInf : -kInf) : __builtin_fmaxf(ax0, ax1);
        const float nry = par_y ? (in_y ? -kInf : kInf) : __builtin_fminf(ay0, ay1), fry = par_y ? (in_y ? kInf : -kInf) : __builtin_fmaxf(ay0, ay1);
        const float nrz = par_z ? (in_z ? -kInf : kInf) : __builtin_fminf(az0, az1), frz = par_z ? (in_z ? kInf : -kInf) : __builtin_fmaxf(az0, az1);
        const float t_in = __builtin_fmaxf(__builtin_fmaxf(nrx, nry), __builtin_fmaxf(nrz, 0.0f));
        const float t_out = __builtin_fminf(__builtin_fminf(frx, fry), frz);
        bool alive = line < my_lines && t_in <= t_out;
        const float px = ox + d.x * t_in, py = oy + d.y * t_in, pz = oz + d.z * t_in;
        int ix = min(max((int)__builtin_floorf((px - lox) * inv_h), 0), nx - 1);
        int iy = min(max((int)__builtin_floorf((py - loy) * inv_h), 0), ny - 1);
        int iz = min(max((int)__builtin_floorf((pz - loz) * inv_h), 0), nz - 1);
        float tnx = par_x ? kInf : ((lox + (float)(ix + (sx > 0 ? 1 : 0)) * h) - ox) * rx;
        float tny = par_y ? kInf : ((loy + (float)(iy + (sy > 0 ? 1 : 0)) * h) - oy) * ry;
        float tnz = par_z ? kInf : ((loz + (float)(iz + (sz > 0 ? 1 : 0)) * h) - oz) * rz;
        float tcur = t_in;
        uint32_t rec = alive ? (uint32_t)((iz * ny + iy) * nx + ix) : 0u;
        for (;;) {
            const bool go = alive && tcur <= __builtin_fminf(t_out, limit);
            if (!wave_any(go)) break;
#ifdef PT_GRID_ROUNDS   // development aid: wave-level rounds instead of record visits, calls instead of positive discriminants
            if (COUNT && lane == 0u) cnt.visits += 1u;
            dbg_rounds += 1u;
#endif
            if (go) {
                const uint32_t last = visit(rec);
                if (last & kGridLinkBit) {
                    rec = last & ~kGridLinkBit;   // the cell continues in another record
                } else {
                    tcur = __builtin_fminf(__builtin_fminf(tnx, tny), tnz);
                    if (tnx <= tny && tnx <= tnz) {
                        ix += sx;
                        alive = (uint32_t)ix < (uint32_t)nx;
                        tnx = ((lox + (float)(ix + (sx > 0 ? 1 : 0)) * h) - ox) * rx;
                    } else if (tny <= tnz) {
                        iy += sy;
                        alive = (uint32_t)iy < (uint32_t)ny;
                        tny = ((loy + (float)(iy + (sy > 0 ? 1 : 0)) * h) - oy) * ry;
                    } else {
                        iz += sz;
                        alive = (uint32_t)iz < (uint32_t)nz;
                        tnz = ((loz + (float)(iz + (sz > 0 ? 1 : 0)) * h) - oz) * rz;
                    }
                    rec = (uint32_t)((iz * ny + iy) * nx + ix);
                }
            }
        }
    }
    if (__builtin_expect(wave_any(full), 0)) {   // a ray from so far away that the reference's discriminant is coarser than a cell: every record
        for (uint32_t r = 0; r < A.grid_records; ++r)
            if (full) (void)visit(r);
    }
#ifdef PT_GRID_ROUNDS   // histograms: record visits per ray [24..55], rounds per call [56..87]
    if (COUNT && start) atomicAdd(&A.debug[24u + min(dbg_visits, 31u)], 1ull);
    if (COUNT && lane == 0u) atomicAdd(&A.debug[56u + min(dbg_rounds, 31u)], 1ull);
#endif
    if (start) w_keys[lane] = idx < 0 ? ~0ull : key4_of(A, best, idx);
}

}  // namespace ptdev
