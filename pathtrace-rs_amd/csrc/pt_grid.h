// pt_grid.h -- closest hit over a UNIFORM CELL GRID: what the GRID flavour of the 4-wide tree kernels walks instead of the tree
// (pt_host.h GridPlan: at least 1 024 spheres of similar size in an even, dense field -- looser clouds keep the tree, which skips empty space).
//
// Why. A visit of the 4-wide tree tests FOUR BOXES (146 VALU instructions) and a ray of BASELINE config 5 -- 10 000 spheres of radius 0.2
// on a 100 x 100 lattice -- needs 10.4 of them plus 3.4 exact sphere tests. A cell of the grid holds the spheres themselves: a visit is
// one 80-byte record (five 16-byte loads: up to four spheres and their list indices), four reference discriminants (sphere.rs:33-37 in
// the reference's own operation order: the filter is EXACT, a sphere whose discriminant is not positive does nothing in the reference
// either) and one step of a 3D-DDA: 2.3 records and 1.9 queued spheres per ray on that scene (8.8 -> 12.5 Grays/s).
//
// Structure of a call (one per trip of the kernel's main loop, all rays of the wave to their end, like bvh4_trace):
//   walk rounds   every lane with cells left visits one record; spheres with a positive discriminant are QUEUED (owner lane | sphere), and
//                 an ESTIMATE of their hit parameter (hardware square root and reciprocal) tightens the lane's walk limit;
//   drain         when no lane can go on (or a queue fills): the queued pairs are tested exactly by the whole wave (drain_pairs4:
//                 sphere.rs:38-64, the ancestor gate of a BVH world, ds_min_u64 into the owner's (t, tie-break) key -- the records are the
//                 per-sphere slot records of GridPlan::rec) and every lane's limit becomes the EXACT one of its key. A lane whose
//                 estimate was too optimistic (the gate refused the hit, a root within a hair of t_min) simply walks on.
// The estimate therefore decides nothing: a lane is finished only when the entry parameter of its next cell lies beyond
// trav4_limit(exact key) or outside the grid.
//
// What makes it the same closest hit. The winner is the (t, tie-break) minimum over every sphere whose reference test accepts the ray
// (DESIGN.md 4.2 / 4.4), so the structure only has to PRESENT every such sphere before the walk ends:
//   * a sphere is registered in every cell its BALL reaches into (a MovingSphere: the box of its sweep), padded by how far the reference's
//     f32 discriminant can inflate the sphere for a ray whose origin lies within KArgs::grid_d_build of it (pt_tree4.h: 0.65e-6 (|o - c|^2 + r^2) / r; the plan uses 1e-6)
//     and by h / 1000 for the walk's own rounding (cell boundaries are recomputed from integer cell numbers, never accumulated);
//   * the walk visits every cell the line passes through from its entry into the grid's box to the first cell whose entry parameter lies
//     beyond best * 1.0005 + 5e-4 (the tree's culling slack: the reference's root can fall below the true parameter);
//   * a ray from FARTHER away (|o - centre| + half diagonal > grid_d_build: the reference's discriminant is coarser than the registration
//     assumed) does not walk the grid at all: it walks the scene's 4-wide tree, whose boxes are padded for the ray at hand (bvh4_trace,
//     called for such lanes at the end; the other lanes of the wave help). How often that happens is a property of the scene (bounce
//     origins on a huge ground far from the spheres): d_build is a cost knob, not a limit.
// Large spheres (KArgs::grid_large) are tested for every ray first, as the tree kernels do with theirs.
#pragma once

namespace ptdev {

constexpr uint32_t kGridLinkBit = 0x80000000u;

// The DDA of one lane: the line (origin + t direction), the cell it is in, the parameters at
// which it crosses the next cell boundary per axis.
struct GridWalk {
    float ox, oy, oz, rx, ry, rz;   // line origin; 1 / direction (0 along an axis the line does not move along)
    float tnx, tny, tnz;            // parameter of the next boundary crossing per axis (+inf along such an axis)
    float tcur, t_out;              // entry parameter of the current cell; where the line leaves the grid's box
    int ix, iy, iz;
    bool alive;
};

__device__ __forceinline__ void grid_walk_start(const KArgs &A, GridWalk &w, f3 o, f3 d, f3 rcp, bool active) {
    const float hx = A.grid_h[0], hy = A.grid_h[1], hz = A.grid_h[2];
    const int nx = (int)A.grid_n[0], ny = (int)A.grid_n[1], nz = (int)A.grid_n[2];
    const float kInf = __builtin_inff();
    const bool par_x = __builtin_fabsf(d.x) < 1.0e-30f, par_y = __builtin_fabsf(d.y) < 1.0e-30f, par_z = __builtin_fabsf(d.z) < 1.0e-30f;
    w.ox = o.x, w.oy = o.y, w.oz = o.z;
    w.rx = par_x ? 0.0f : rcp.x, w.ry = par_y ? 0.0f : rcp.y, w.rz = par_z ? 0.0f : rcp.z;
    // clip against the grid's box
    const float lox = A.grid_min[0], loy = A.grid_min[1], loz = A.grid_min[2];
    const float hix = lox + (float)nx * hx, hiy = loy + (float)ny * hy, hiz = loz + (float)nz * hz;
    const float ax0 = (lox - o.x) * w.rx, ax1 = (hix - o.x) * w.rx, ay0 = (loy - o.y) * w.ry, ay1 = (hiy - o.y) * w.ry, az0 = (loz - o.z) * w.rz, az1 = (hiz - o.z) * w.rz;
    const bool in_x = o.x >= lox && o.x <= hix, in_y = o.y >= loy && o.y <= hiy, in_z = o.z >= loz && o.z <= hiz;
    const float nrx = par_x ? (in_x ? -kInf : kInf) : __builtin_fminf(ax0, ax1), frx = par_x ? (in_x ? kInf : -kInf) : __builtin_fmaxf(ax0, ax1);
    const float nry = par_y ? (in_y ? -kInf : kInf) : __builtin_fminf(ay0, ay1), fry = par_y ? (in_y ? kInf : -kInf) : __builtin_fmaxf(ay0, ay1);
    const float nrz = par_z ? (in_z ? -kInf : kInf) : __builtin_fminf(az0, az1), frz = par_z ? (in_z ? kInf : -kInf) : __builtin_fmaxf(az0, az1);
    const float t_in = __builtin_fmaxf(__builtin_fmaxf(nrx, nry), __builtin_fmaxf(nrz, 0.0f));
    w.t_out = __builtin_fminf(__builtin_fminf(frx, fry), frz);
    w.alive = active && t_in <= w.t_out;
    const float px = o.x + d.x * t_in, py = o.y + d.y * t_in, pz = o.z + d.z * t_in;
    w.ix = min(max((int)__builtin_floorf((px - lox) * A.grid_inv_h[0]), 0), nx - 1);
    w.iy = min(max((int)__builtin_floorf((py - loy) * A.grid_inv_h[1]), 0), ny - 1);
    w.iz = min(max((int)__builtin_floorf((pz - loz) * A.grid_inv_h[2]), 0), nz - 1);
    w.tnx = par_x ? kInf : ((lox + (float)(w.ix + (d.x >= 0.0f ? 1 : 0)) * hx) - o.x) * w.rx;
    w.tny = par_y ? kInf : ((loy + (float)(w.iy + (d.y >= 0.0f ? 1 : 0)) * hy) - o.y) * w.ry;
    w.tnz = par_z ? kInf : ((loz + (float)(w.iz + (d.z >= 0.0f ? 1 : 0)) * hz) - o.z) * w.rz;
    w.tcur = t_in;
}
__device__ __forceinline__ uint32_t grid_walk_cell(const KArgs &A, const GridWalk &w) { return (uint32_t)((w.iz * (int)A.grid_n[1] + w.iy) * (int)A.grid_n[0] + w.ix); }
// one step: into the neighbour across the nearest boundary (the boundary of the NEW cell is formed from its integer number). Straight-line
// code on purpose: the walk issues a record's loads and takes this step in their shadow.
__device__ __forceinline__ void grid_walk_step(const KArgs &A, GridWalk &w, f3 d) {
    // (every field through a local copy: selects between the FIELDS of `w` become selects between their addresses, and the walk ends up in scratch)
    const float tnx = w.tnx, tny = w.tny, tnz = w.tnz, ox = w.ox, oy = w.oy, oz = w.oz, rx = w.rx, ry = w.ry, rz = w.rz;
    const int ix = w.ix, iy = w.iy, iz = w.iz;
    const uint32_t gnx = A.grid_n[0], gny = A.grid_n[1], gnz = A.grid_n[2];
    const float lox = A.grid_min[0], loy = A.grid_min[1], loz = A.grid_min[2];
    const bool step_x = tnx <= tny && tnx <= tnz, step_y = !step_x && tny <= tnz;
    const bool up = (step_x ? d.x : (step_y ? d.y : d.z)) >= 0.0f;
    const int i_new = (step_x ? ix : (step_y ? iy : iz)) + (up ? 1 : -1);
    const uint32_t n_axis = step_x ? gnx : (step_y ? gny : gnz);
    const float lo_axis = step_x ? lox : (step_y ? loy : loz), h_axis = step_x ? A.grid_h[0] : (step_y ? A.grid_h[1] : A.grid_h[2]);
    const float o_axis = step_x ? ox : (step_y ? oy : oz), r_axis = step_x ? rx : (step_y ? ry : rz);
    const float tn_new = ((lo_axis + (float)(i_new + (up ? 1 : 0)) * h_axis) - o_axis) * r_axis;
    w.tcur = __builtin_fminf(__builtin_fminf(tnx, tny), tnz);
    w.alive = (uint32_t)i_new < n_axis;
    w.ix = step_x ? i_new : ix, w.iy = step_y ? i_new : iy, w.iz = (step_x || step_y) ? iz : i_new;
    w.tnx = step_x ? tn_new : tnx, w.tny = step_y ? tn_new : tny, w.tnz = (step_x || step_y) ? tnz : tn_new;
}

// the same step in a grid that is ONE cell thick along y (a layer of spheres, BASELINE config 5: 99 x 1 x 99): the line leaves such a cell through
// its top or bottom only by leaving the grid -- t_out knows -- so y never steps (grid_trace sets tny = +inf) and every three-way select is a two-way one
__device__ __forceinline__ void grid_walk_step_xz(const KArgs &A, GridWalk &w, f3 d) {
    const float tnx = w.tnx, tnz = w.tnz, ox = w.ox, oz = w.oz, rx = w.rx, rz = w.rz;
    const int ix = w.ix, iz = w.iz;
    const bool step_x = tnx <= tnz;
    const bool up = (step_x ? d.x : d.z) >= 0.0f;
    const int i_new = (step_x ? ix : iz) + (up ? 1 : -1);
    const uint32_t n_axis = step_x ? A.grid_n[0] : A.grid_n[2];
    const float lo_axis = step_x ? A.grid_min[0] : A.grid_min[2], h_axis = step_x ? A.grid_h[0] : A.grid_h[2];
    const float o_axis = step_x ? ox : oz, r_axis = step_x ? rx : rz;
    const float tn_new = ((lo_axis + (float)(i_new + (up ? 1 : 0)) * h_axis) - o_axis) * r_axis;
    w.tcur = __builtin_fminf(tnx, tnz);
    w.alive = (uint32_t)i_new < n_axis;
    w.ix = step_x ? i_new : ix, w.iz = step_x ? iz : i_new;
    w.tnx = step_x ? tn_new : tnx, w.tnz = step_x ? tnz : tn_new;
}

template <bool MOVING, bool COUNT, int BLK>
__device__ __forceinline__ void grid_trace(const KArgs &A, uint16_t *s_stack, uint32_t *leafq, uint32_t *w_pairs, unsigned long long *w_keys, uint32_t *w_park, unsigned long long &parked,
                                           f3 o, f3 d, const DivA &av, float time, bool start, Steal4 &cnt, unsigned long long *sec = nullptr) {
    const int tid = threadIdx.x;
    const uint32_t lane = (uint32_t)tid & 63u;
#ifdef PT_SECTIONS
    unsigned long long sub_last = __builtin_readcyclecounter();   // sec[5] walk rounds, sec[6] drains, sec[7] rounds (count)
#define PT_SUBT(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); sec[i] += now_ - sub_last; sub_last = now_; } while (0)
#else
    (void)sec;
#define PT_SUBT(i) do { } while (0)
#endif
    const uint32_t owner_tag = lane << kPairLaneShift;
    const float a = av.a;
    const f3 rcp = mk3(recip_exact(d.x), recip_exact(d.y), recip_exact(d.z));   // ray.rs:14 (the gates of a BVH world read it; the walk where it is finite)
    // ---- is this ray the grid's? (origin within d_build of every sphere: what the registration was padded for)
    const float gx = o.x - A.grid_centre[0], gy = o.y - A.grid_centre[1], gz = o.z - A.grid_centre[2];
    const float dist = __builtin_sqrtf((gx * gx + gy * gy) + gz * gz) * 1.001f + A.grid_half_diag;
    // PARKED walks (round 6). The rounds of a call thin out -- 37, 27, 16, 9, 5, 3, 2 ... lanes walking in rounds 0, 1, 2, ... on config 5 -- and a
    // round costs the wave the same ~160 instructions and one L2 round trip whether 37 lanes walk or one: a third of all rounds ran for the
    // last four stragglers. When at most A.grid_park_max lanes are still walking (after A.grid_park_after rounds, in a wave that has other
    // work), they PARK: eight words of walk state go to the wave's LDS slots, the call ends with its drain, the lanes sit out this trip's shading
    // (pt_kernel.h: `grid_parked`) and RESUME in the next call, whose early rounds run anyway. Nothing about the result changes: a resumed walk
    // visits the cells it would have visited, its key (w_keys, LDS) and its ray (the lane's registers) are untouched in between.
    const bool resume = start && ((parked >> lane) & 1ull) != 0ull;
    const bool fresh = start && !resume;
    const bool far = fresh && !(dist <= A.grid_d_build);   // (also a NaN origin)
    float limit = kMaxT;
    if (fresh && !far) {
        float best = kMaxT;
        int idx = -1;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < A.n_grid_large; ++j) {
            const int k = (int)A.grid_large[j];
            bvh_leaf(A, k, sphere_at<MOVING>(A, k, A.spheres[k], time), o, d, rcp, av, best, idx, rank);
        }
        w_keys[lane] = idx < 0 ? ~0ull : key4_of(A, best, idx);
        limit = trav4_limit(idx < 0 ? kMaxT : best);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    GridWalk w;
    grid_walk_start(A, w, o, d, rcp, start && !far);
    const bool layer = A.grid_n[1] == 1u;   // (wave-uniform) one cell along y: the walk steps in x and z only
    if (layer) w.tny = __builtin_inff();
    uint32_t rec = grid_walk_cell(A, w);
    bool cont = false;   // `rec` continues the cell the lane is already in (its walk has stepped on: such a record is visited whatever the limit says)
    if (parked != 0ull) {   // (wave-uniform) somebody resumes
        if (resume) {
            const uint32_t *q = w_park + kGridParkWords * __builtin_amdgcn_mbcnt_hi((uint32_t)(parked >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)parked, 0u));
            const uint4 qa = *reinterpret_cast<const uint4 *>(q), qb = *reinterpret_cast<const uint4 *>(q + 4);
            w.tnx = __uint_as_float(qa.x), w.tny = __uint_as_float(qa.y), w.tnz = __uint_as_float(qa.z), w.tcur = __uint_as_float(qa.w);
            w.t_out = __uint_as_float(qb.x);
            w.ix = (int)(qb.y & 1023u), w.iy = (int)((qb.y >> 10) & 1023u), w.iz = (int)((qb.y >> 20) & 1023u);
            w.alive = (qb.y >> 30) != 0u;
            cont = (qb.z & kGridLinkBit) != 0u;
            rec = qb.z & ~kGridLinkBit;
            // (the drain that ended the last call may have found this ray a hit: the exact limit of its key; an empty key's t field is a NaN pattern: not < kMaxT)
            limit = trav4_limit(__uint_as_float((uint32_t)(w_keys[lane] >> 32)));
        }
        parked = 0ull;
        __builtin_amdgcn_wave_barrier();
    }
    const uint32_t n_started = (uint32_t)__popcll(wave_ballot(start));
    uint32_t rounds_done = 0u;   // (wave-uniform)
    const float ia = __builtin_amdgcn_rcpf(a);
    uint32_t qn = 0;
#ifdef PT_GRID_ROUNDS
    uint32_t dbg_visits = 0u, dbg_rounds = 0u;
    if (COUNT && lane == 0u) cnt.leaves += 1u;
#endif
#ifdef PT_ROUNDCAP   // (timing experiment, WRONG images: walks cut after PT_ROUNDCAP rounds -- what the rounds beyond would be worth if they cost nothing)
    uint32_t cap_rounds = 0;
#endif
    for (;;) {
        bool go = cont || (w.alive && w.tcur <= __builtin_fminf(w.t_out, limit));
#ifdef PT_ROUNDCAP
        if (cap_rounds++ >= (uint32_t)(PT_ROUNDCAP)) go = false, cont = false, w.alive = false;
#endif
        {   // park the last few walkers? First the drain -- every limit becomes exact, a lane whose estimate was too optimistic walks on --, then whoever still has
            // cells left parks, if the wave's slots hold them all (otherwise the call simply goes on)
            const uint32_t n_go = (uint32_t)__popcll(wave_ballot(go));
            if (n_go != 0u && n_go <= A.grid_park_max && rounds_done >= A.grid_park_after && n_started >= 32u) {
                PT_SUBT(5);
                if (drain_pairs4<MOVING, BLK>(A, A.grid_rec, leafq, w_pairs, w_keys, qn, o, d, av, time, owner_tag)) limit = trav4_limit(__uint_as_float((uint32_t)(w_keys[lane] >> 32)));
                go = cont || (w.alive && w.tcur <= __builtin_fminf(w.t_out, limit));
                PT_SUBT(6);
                const unsigned long long gm = wave_ballot(go);
                if (gm == 0ull) break;
                if ((uint32_t)__popcll(gm) <= kGridParkMax) {
                    if (go) {
                        uint32_t *q = w_park + kGridParkWords * __builtin_amdgcn_mbcnt_hi((uint32_t)(gm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)gm, 0u));
                        *reinterpret_cast<uint4 *>(q) = make_uint4(__float_as_uint(w.tnx), __float_as_uint(w.tny), __float_as_uint(w.tnz), __float_as_uint(w.tcur));
                        *reinterpret_cast<uint4 *>(q + 4) = make_uint4(__float_as_uint(w.t_out), ((uint32_t)w.ix & 1023u) | (((uint32_t)w.iy & 1023u) << 10) | (((uint32_t)w.iz & 1023u) << 20) | (w.alive ? 1u << 30 : 0u),   // (a walk that has left the grid holds -1 or n somewhere: masked, and never read again)
                                                                       rec | (cont ? kGridLinkBit : 0u), 0u);
                    }
                    parked = gm;
                    break;
                }
            }
        }
        rounds_done += 1u;
        if (!wave_any(go) || wave_any(qn > A.drain_at)) {
            PT_SUBT(5);
            // exact tests of everything queued; the limit becomes the exact one of the lane's key (an empty key's t field is a NaN pattern: not < kMaxT)
#ifdef PT_GRID_ROUNDS
            const bool forced = wave_any(go);   // (a full queue forced this drain: lanes were still walking)
            const uint32_t queued_now = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_sum(qn), 63);
#endif
            if (drain_pairs4<MOVING, BLK>(A, A.grid_rec, leafq, w_pairs, w_keys, qn, o, d, av, time, owner_tag)) limit = trav4_limit(__uint_as_float((uint32_t)(w_keys[lane] >> 32)));
            go = cont || (w.alive && w.tcur <= __builtin_fminf(w.t_out, limit));
#ifdef PT_GRID_ROUNDS
            const unsigned long long dbg_on = wave_ballot(go);
            if (COUNT && lane == 0u && queued_now != 0u) {   // [92] drains with pairs, [93] of them forced by a full queue, [94] pairs drained, [95] lanes that walk on after a drain
                atomicAdd(&A.debug[92], 1ull);
                atomicAdd(&A.debug[93], forced ? 1ull : 0ull);
                atomicAdd(&A.debug[94], (unsigned long long)queued_now);
                atomicAdd(&A.debug[95], (unsigned long long)__popcll(dbg_on));
            }
#endif
            PT_SUBT(6);
            if (!wave_any(go)) break;
        }
#ifdef PT_SECTIONS
        sec[7] += 1ull;
#endif
#ifdef PT_GRID_ROUNDS   // development aid: wave-level rounds instead of record visits, calls instead of positive discriminants
        if (COUNT && lane == 0u) cnt.visits += 1u;
        const unsigned long long dbg_gm = wave_ballot(go);
        if (COUNT && lane == 0u) {   // lanes walking in round r of a call ([96 + r], r < 15; later rounds in [111]) and how many calls reach it ([112 + r]); drains in [92..95]
            const uint32_t r = dbg_rounds < 15u ? dbg_rounds : 15u;
            atomicAdd(&A.debug[96u + r], (unsigned long long)__popcll(dbg_gm));
            atomicAdd(&A.debug[112u + r], 1ull);
        }
        dbg_rounds += 1u;
#endif
        if (go) {
            const uint4 *cp = A.grid_cells + (size_t)rec * 5u;
            uint4 q0 = cp[0], q1 = cp[1], q2 = cp[2], q3 = cp[3], m = cp[4];
            // the step into the next cell does not depend on what the record holds: taken here, in the shadow of the loads
            if (!cont) {
                if (layer) grid_walk_step_xz(A, w, d);
                else grid_walk_step(A, w, d);
            }
            // (all five loads in flight at once: left alone, the scheduler issues them one by one, each behind the arithmetic of the one
            //  before -- four round trips to the L2 per record instead of one -- to save the sixteen registers this takes)
            asm volatile("" : "+v"(q0.x), "+v"(q0.y), "+v"(q0.z), "+v"(q0.w), "+v"(q1.x), "+v"(q1.y), "+v"(q1.z), "+v"(q1.w), "+v"(q2.x), "+v"(q2.y), "+v"(q2.z), "+v"(q2.w),
                         "+v"(q3.x), "+v"(q3.y), "+v"(q3.z), "+v"(q3.w), "+v"(m.x), "+v"(m.y), "+v"(m.z), "+v"(m.w));
#ifdef PT_GRID_ROUNDS
            dbg_visits += 1u;
#else
            if (COUNT) cnt.visits += 1u;
#endif
            // the reference's discriminants (sphere.rs:33-37, its operation order), two spheres per packed instruction
            typedef float v2f __attribute__((ext_vector_type(2)));
            const uint32_t ks[4] = {m.x, m.y, m.z, m.w};
            const uint4 qp[2][2] = {{q0, q1}, {q2, q3}};
            float nb_min = __builtin_inff(), d_sel = -1.0f;   // the positive candidate whose closest approach comes first: the one the estimate is formed for
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                v2f X = {__uint_as_float(qp[pr][0].x), __uint_as_float(qp[pr][0].y)}, Y = {__uint_as_float(qp[pr][0].z), __uint_as_float(qp[pr][0].w)};
                v2f Z = {__uint_as_float(qp[pr][1].x), __uint_as_float(qp[pr][1].y)}, Rr = {__uint_as_float(qp[pr][1].z), __uint_as_float(qp[pr][1].w)};
                if (MOVING) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const uint32_t k = ks[2 * pr + e];
                        const float4 c = sphere_at<MOVING>(A, (int)(k < A.n_spheres ? k : 0u), make_float4(X[e], Y[e], Z[e], Rr[e]), time);
                        X[e] = c.x, Y[e] = c.y, Z[e] = c.z;
                    }
                }
                const v2f ocx = o.x - X, ocy = o.y - Y, ocz = o.z - Z;
                const v2f b = (ocx * d.x + ocy * d.y) + ocz * d.z;
                const v2f cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - Rr * Rr;
                const v2f disc = b * b - a * cc;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const uint32_t k = ks[2 * pr + e];
                    // a sphere the ray is leaving through its surface (every scattered ray's own): its far root (-b + sqrt(disc)) / a lies
                    // below 2.5e-4 when -cc < 5e-4 b, four times under t_min -- the reference rejects both roots (sphere.rs:41,52), whatever the
                    // last bits of its quotient: the computed root is off by ~ eps b / a, under 1e-5 while b < 100 a (the bound is relative to a:
                    // the C ABI takes rays of any direction length, and with a << 1 a bound on b alone would let that error reach t_min): not queued
                    const bool leaving = b[e] > 0.0f && b[e] < 100.0f * a && -cc[e] < 5.0e-4f * b[e];
                    // (an empty slot of a record -- its index word is kGridNone or the link -- holds the sphere (3e38, 3e38, 3e38; 0): oc.oc overflows to +inf for
                    //  any finite origin, so its discriminant is b b - a inf = -inf or NaN, never > 0: no test of the index word is needed. MOVING
                    //  records go through sphere_at, which wants a valid index: they keep the test.)
                    const bool pos = disc[e] > 0.0f && (!MOVING || k < A.n_spheres) && !leaving;
                    leafq[qn * BLK + tid] = owner_tag | k;
                    qn += pos ? 1u : 0u;
#ifndef PT_GRID_ROUNDS
                    if (COUNT) cnt.leaves += pos ? 1u : 0u;
#endif
                    const bool nearer = pos && -b[e] < nb_min;
                    nb_min = nearer ? -b[e] : nb_min, d_sel = nearer ? disc[e] : d_sel;
                }
            }
            // where that candidate would be hit, roughly: the walk pauses beyond it until the drain has the exact answer
            float tq = __builtin_inff();
            {
                const float sq = __builtin_amdgcn_sqrtf(d_sel);
                const float t1 = (nb_min - sq) * ia, t2 = (nb_min + sq) * ia;
                tq = t1 > 2.0e-3f ? t1 : (t2 > 2.0e-3f ? t2 : tq);   // (no candidate: d_sel = -1, the root is NaN and nothing compares true)
            }
            limit = __builtin_fminf(limit, tq * 1.001f + 1.0e-3f);
            cont = (m.w & kGridLinkBit) != 0u;   // the cell continues in another record
            rec = cont ? (m.w & ~kGridLinkBit) : grid_walk_cell(A, w);
        }
    }
    PT_SUBT(5);
#undef PT_SUBT
#ifdef PT_GRID_ROUNDS   // histograms: record visits per ray [24..55], rounds per call [56..87]
    if (COUNT && start) atomicAdd(&A.debug[24u + min(dbg_visits, 31u)], 1ull);
    if (COUNT && lane == 0u) atomicAdd(&A.debug[56u + min(dbg_rounds, 31u)], 1ull);
#endif
    // rays from beyond d_build: the 4-wide tree, whose boxes are padded for the ray at hand (lanes without such a ray help: bvh4_trace shares work)
    // (-DPT_SECTIONS builds: the tree walk's cycles land in the same section slots -- a null `sec` here faulted such builds on config 5)
    if (__builtin_expect(wave_any(far), 0)) bvh4_trace<MOVING, COUNT, BLK>(A, s_stack, leafq, w_pairs, w_keys, o, d, av, time, far, cnt, sec);
}

}  // namespace ptdev
