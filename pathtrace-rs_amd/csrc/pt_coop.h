// pt_coop.h -- wave-cooperative mode of the wide list kernels (one workgroup per CU, palette stacks): ONE pixel per wave.
//
// Why. A pixel's samples are ONE serial RNG stream (scene.rs:96-111), so a pixel is a chain of dependent ray iterations -- 64 for a
// sky pixel, ~700 for a glass pixel at 64 spp -- and in the per-lane kernel a wave-iteration costs the same ~2 200 instructions
// whether 64 of its lanes hold a ray or one does (a wave ALONE on the GPU needs ~6.6 us for it). That latency, not throughput, is
// what the end of a frame, the heaviest pixel, and every shard of a multi-GPU frame wait for. Here the 64 lanes of a wave work on
// ONE ray: each lane tests ceil(n / 64) spheres with the reference's exact arithmetic (sphere.rs:29-66), the winner is the
// lexicographic minimum of (t, list index) -- or (t, DFS rank) behind the ancestor-AABB gate in a BVH world -- over the wave, which
// is what hitable_list.rs:40-56 / bvh.rs:37-62 return (pt_prefilter.h, DESIGN.md 4.2 "order-independent closest hit"), and everything else (camera,
// scatter, RNG) is computed redundantly by all lanes on wave-uniform data: a few hundred dependent instructions per ray instead of
// 2 200. Per ray that is ~15x the work of the per-lane kernel, so it only ever runs on waves that would otherwise idle.
//
// Who runs it. A wave whose main loop has ended (the work list is dry and its own pixels are done or handed over) becomes a WORKER:
// it offers a token, polls the hand-over queue, and traces the pixels it pops to their last sample. A wave still in its main loop
// ("normal" wave) hands a pixel over -- at a sample boundary, where a pixel's whole state is its RNG stream, colour sum and counters:
// 64 bytes -- when the list is dry and a token is on offer, the pixel with the most estimated work left first. The RNG stream
// travels with the pixel, so the pixel's value cannot depend on who traces it (tests: every `-m gpu` frame on a wide kernel goes
// through this; small frames migrate almost entirely).
//
// Protocol. Everything other waves read is written with write-through (`sc1`, agent-scope relaxed atomic) stores and read with
// `sc1` loads; a payload is drained (`s_waitcnt vmcnt(0)`) before its flag is stored. No agent-scope fence (a release writes back the
// XCD's whole L2, an acquire drops a CU's L1: microseconds each) and NO word that many waves poll: one word takes ~88 M accesses/s,
// which 4 096 waves polling every few microseconds exceed many times over -- the first versions of this file (a token counter and a
// FIFO of idle workers behind the work counter) made a config-3 frame 1.5 ms slower with NOTHING handed over, every access to the
// hot line waiting tens of microseconds in its queue. Every wave has a MAILBOX instead (KArgs::tail_box + 16 w, 128 bytes):
//     [0..6] one pixel   [8] state = generation * 4 + {0 busy / never idle, 1 IDLE, 2 CLAIMED, 3 PIXEL}   [9] exit = generation * 4 + 1
//   worker w, idle:  state = IDLE; poll own state and exit words (s_sleep between polls). PIXEL -> trace it, start over. EXIT -> take the
//                    IDLE back with a CAS and return (lost the CAS: a pixel is on its way, trace it first).
//   normal wave:     once the list is dry it PROBES one mailbox per poll -- a pseudo-random other wave's: a load of a line nobody else
//                    is looking at, issued at the top of the iteration and read at its end. IDLE and the policy below says go:
//                    CAS(state, IDLE -> CLAIMED); won: write the pixel, drain, state = PIXEL.
//   end:             the control block behind KArgs::work_counter counts workgroups that entered the kernel ([6]) and workgroups all
//                    of whose waves have left their main loops ([1]; the waves of a workgroup count in LDS first). The wave whose
//                    count makes them equal is the last one that could hand anything over (workgroups that are not resident yet are
//                    not counted; if they start late they find no idle worker and trace their pixels themselves, and the last of
//                    THEM repeats this): it stores EXIT into every mailbox of the grid, 64 per instruction.
// When a normal wave hands over (the policy; KArgs::tail_live_max / tail_streak): the list is dry, the probed worker is idle, and
// either the wave is down to its last few pixels (the hand-over shortens ITS chain: a wave-iteration costs the same for 3 live
// lanes as for 64) or its last few probes ALL found idle workers (most of the GPU is idle, i.e. latency-bound: a worker costs ~15x
// the instructions per ray, which only idle issue slots can pay for -- handing over whenever somebody was idle made a config-3 frame
// 8x slower). The pixel with the most estimated work left goes first.
#pragma once

#ifdef PT_DEVKNOBS   // development builds can switch parts of the mode off (PTGPU_COOP_DBG) to price them
#define PT_COOP_DBG(bit) (!(A.tail_dbg & (bit)))
#else
#define PT_COOP_DBG(bit) true
#endif

namespace ptdev {

enum : uint32_t { kCtlWork = 0, kCtlDone = 1, kCtlStarted = 6 };
#ifndef PT_COOP_PRIO
#define PT_COOP_PRIO 1
#endif
constexpr int kCoopBusyPrio = PT_COOP_PRIO;   // s_setprio of a wave of the wide frame kernels while it is in its main loop (workers: 0)
constexpr uint32_t kBoxIdle = 1u, kBoxClaimed = 2u, kBoxPixel = 3u;

__device__ __forceinline__ uint32_t ctl_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t wt_load(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wt_store(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wt_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// max over the 64 lanes of a wave, valid in lane 63 (same DPP steps as wave_inclusive_sum; 0 is the identity)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) {
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

// One handed-over pixel (64 bytes): [0..3] xoshiro256+ state, [4..5] colour sum | pixel, [6] samples done | rays so far
__device__ __forceinline__ void coop_store_pixel(uint64_t *box, const Rng &rng, f3 col, uint32_t pxy, uint32_t samples_done, uint32_t rays) {
    wt_store(box + 0, rng.s0), wt_store(box + 1, rng.s1), wt_store(box + 2, rng.s2), wt_store(box + 3, rng.s3);
    wt_store(box + 4, (uint64_t)__float_as_uint(col.x) | ((uint64_t)__float_as_uint(col.y) << 32));
    wt_store(box + 5, (uint64_t)__float_as_uint(col.z) | ((uint64_t)pxy << 32));
    wt_store(box + 6, (uint64_t)samples_done | ((uint64_t)rays << 32));
}

// A normal wave's lane hands its pixel (at a sample boundary) to the worker whose mailbox it found IDLE; false: somebody else claimed
// that worker first, the pixel stays.
__device__ __forceinline__ bool coop_hand_over(const KArgs &A, uint32_t worker, const Rng &rng, f3 col, uint32_t pxy, uint32_t samples_done, uint32_t rays) {
    uint64_t *box = A.tail_box + 16u * (size_t)worker;
    const unsigned long long stamp = (unsigned long long)A.tail_gen << 2;
    if (atomicCAS(reinterpret_cast<unsigned long long *>(box + 8), stamp | kBoxIdle, stamp | kBoxClaimed) != (stamp | kBoxIdle)) return false;
    coop_store_pixel(box, rng, col, pxy, samples_done, rays);
    wt_drain();
    wt_store(box + 8, stamp | kBoxPixel);
    return true;
}

// The spheres one lane tests for every ray of every pixel its wave traces: lane l holds list entries l, l + 64, ... (the scan table's
// (cx, cy, cz, r * r) rows; launch(): at most kCoopChunks * 64 = 512 spheres when this mode is on) -- loaded ONCE per worker, so a ray's scan reads no memory.
constexpr int kCoopChunks = 8;
#ifndef PT_COOP_SPH_LDS
#define PT_COOP_SPH_LDS 0   // 1: a worker reads its spheres from the LDS table for every ray instead of holding them in 32 registers (NOTES.md)
#endif
struct CoopSpheres {
    float4 s[kCoopChunks];
};
__device__ __forceinline__ void coop_load_spheres(CoopSpheres &mine, const float4 *sph, uint32_t n) {
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int j = 0; j < kCoopChunks; ++j) {
        const uint32_t k = 64u * (uint32_t)j + lane;
        mine.s[j] = k < n ? sph[k] : make_float4(3.0e38f, 3.0e38f, 3.0e38f, 0.0f);   // (never hit: its discriminant is -inf or NaN, like the scan table's padding)
    }
}

// scene.rs:105-116 for ONE pixel from sample `samples_done` on, all 64 lanes on each ray. Every value below is wave-uniform except
// inside the scan; `s_shade`: the LDS shading records.
template <bool MOVING, bool GATED>
__device__ __forceinline__ void coop_trace_pixel(const KArgs &A, const GateSrc &G, const float4 *mot, const float4 *s_par, const CoopSpheres &mine, const float4 *sph, const float4 *s_shade,
                                                 const PerlinLds &pn, const uint64_t *box, unsigned long long &wave_rays) {
    const uint32_t lane = threadIdx.x & 63u;
    Rng rng;
    rng.s0 = wt_load(box + 0), rng.s1 = wt_load(box + 1), rng.s2 = wt_load(box + 2), rng.s3 = wt_load(box + 3);
    const uint64_t w4 = wt_load(box + 4), w5 = wt_load(box + 5), w6 = wt_load(box + 6);
    f3 col = mk3(__uint_as_float((uint32_t)w4), __uint_as_float((uint32_t)(w4 >> 32)), __uint_as_float((uint32_t)w5));
    const uint32_t pxy = (uint32_t)(w5 >> 32);
    uint32_t sample = (uint32_t)w6, rays = (uint32_t)(w6 >> 32);
    const uint32_t rays_before = rays;
#ifdef PT_DEVKNOBS   // (PTGPU_TIMING=1: one log entry per handed-over pixel, pt_launch.hip prints the timeline)
    const unsigned long long log_t0 = A.wave_end ? wall_clock64() : 0ull;
    const uint32_t log_sample0 = sample;
#endif
    const float4 pcnt = s_par[12];
    const uint32_t max_depth = __float_as_uint(pcnt.z), n_samples = __float_as_uint(pcnt.w);
    const float fpx = (float)(pxy & 0xffffu), fpy = (float)((pxy >> 16) * A.shard_count + A.shard_index);
    const uint32_t n = A.n_spheres;
#ifdef PT_COOPSEC   // development aid: cycles per section of the worker's loop (camera, scan, reduce, shade, fold) and counts, into debug[80..]
    unsigned long long csec[8] = {0, 0, 0, 0, 0, 0, 0, 0}, clast = __builtin_readcyclecounter();
#define PT_CSEC(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); csec[i] += now_ - clast; clast = now_; } while (0)
#else
#define PT_CSEC(i) do { } while (0)
#endif
    f3 level = mk3(0.f, 0.f, 0.f);   // lane L keeps the attenuation of level L of the path (launch(): max_depth <= 64 when this mode is on)
    while (sample < n_samples) {
        // ---- scene.rs:107-108 + camera.rs:56-68 (the same expressions as the main loop's camera role). The camera is read from the
        // LDS parameter block for every sample (behind an opaque offset: hoisted out of the loop its 24 floats would be live across
        // the whole scan, next to the lane's 32 sphere registers)
        uint32_t par_off = 0u;
        asm volatile("" : "+v"(par_off));
        const float4 *par = s_par + par_off;
        const float4 c0 = par[4], c1 = par[5], c2 = par[6], c3 = par[7], c4 = par[8], c5 = par[9], pn2 = par[3];
        const f3 cam_origin = mk3(c0.x, c0.y, c0.z), cam_llc = mk3(c0.w, c1.x, c1.y), cam_horizontal = mk3(c1.z, c1.w, c2.x), cam_vertical = mk3(c2.y, c2.z, c2.w),
                 cam_u = mk3(c3.x, c3.y, c3.z), cam_v = mk3(c3.w, c4.x, c4.y);
        const float cam_time0 = c5.y, cam_time1 = c5.z, cam_lens_radius = c5.w;
        const float u = rng_plus(rng, fpx) * pn2.z;
        const float v = rng_plus(rng, fpy) * pn2.w;
        float sx, sy;
        for (;;) {   // math.rs:6-13
            sx = rng_pm1(rng), sy = rng_pm1(rng);
            if (((sx * sx + sy * sy) + 0.0f) < 1.0f) break;
        }
        const float rdx = cam_lens_radius * sx, rdy = cam_lens_radius * sy;
        const f3 offset = add3(scale3(cam_u, rdx), scale3(cam_v, rdy));
        const float tdraw = rng_f32(rng);   // camera.rs:59
        const float rtime = MOVING ? cam_time0 + tdraw * (cam_time1 - cam_time0) : 0.0f;
        f3 o = add3(cam_origin, offset);
        f3 d = normalize3(sub3(sub3(add3(add3(cam_llc, scale3(cam_horizontal, u)), scale3(cam_vertical, v)), cam_origin), offset));
        uint32_t depth = 0;
        f3 V;
        PT_CSEC(0);
        for (;;) {   // scene.rs:49-71, one level per trip
            rays += 1u;
            // ---- hitable_list.rs:40-56 over sphere.rs:29-66. Every lane forms the reference's discriminant for its spheres and
            // remembers the last positive one; the roots (sphere_hit_t) are then formed ONCE, for all lanes that have a candidate.
            // A lane with two positives (about one ray in thirty) sends the wave through the per-sphere form below.
            const float a = dot3(d, d);
            const DivA av{a, recip_unit_range(a), in_unit_range(a)};
            const f3 rcp = GATED ? mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z) : mk3(0.f, 0.f, 0.f);   // ray.rs:14 (the gate of a BVH world reads it)
            unsigned long long key = ~0ull;
            // smaller t wins; equal t: the lower list index (hitable_list.rs:48) or, in a BVH world, the DFS-later leaf (bvh.rs:47-53)
            auto offer = [&](float t, uint32_t k) {
                if (t < kMaxT) {
                    const uint32_t low = GATED ? (((0xffffu - G.rank[k]) << 16) | k) : k;
                    const unsigned long long kq = ((unsigned long long)__float_as_uint(t) << 32) | low;
                    if (kq < key && (!GATED || gate_pass_from(A, G, (int)k, o, rcp))) key = kq;
                }
            };
            auto disc_of = [&](int j, float &b) -> float {
                const uint32_t k = 64u * (uint32_t)j + lane;
#if PT_COOP_SPH_LDS
                const float4 c = sphere_at_m<MOVING>(mot, (int)(k < n ? k : n - 1u), k < n ? sph[k] : make_float4(3.0e38f, 3.0e38f, 3.0e38f, 0.0f), rtime);   // (read from the LDS table per ray: 32 registers fewer)
#else
                const float4 c = sphere_at_m<MOVING>(mot, (int)(k < n ? k : n - 1u), mine.s[j], rtime);
#endif
                const float ocx = o.x - c.x, ocy = o.y - c.y, ocz = o.z - c.z;
                b = (ocx * d.x + ocy * d.y) + ocz * d.z;
                const float cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - c.w;
                return b * b - a * cc;
            };
            float bsel = 0.0f, dsel = -1.0f;
            uint32_t ksel = 0u, npos = 0u;
#pragma unroll
            for (int g = 0; g < kCoopChunks / 4; ++g) {
                if (256u * (uint32_t)g < n) {   // (wave-uniform)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int j = 4 * g + q;
                        float b;
                        const float disc = disc_of(j, b);
                        const bool pos = disc > 0.0f;
                        bsel = pos ? b : bsel, dsel = pos ? disc : dsel, ksel = pos ? 64u * (uint32_t)j + lane : ksel;
                        npos += pos ? 1u : 0u;
                    }
                }
            }
            if (wave_any(npos != 0u)) offer(sphere_hit_t(av, bsel, dsel, npos != 0u), ksel);
            if (__builtin_expect(wave_any(npos > 1u), 0)) {
#pragma unroll
                for (int j = 0; j < kCoopChunks; ++j) {
                    if (64u * (uint32_t)j < n) {
                        float b;
                        const float disc = disc_of(j, b);
                        const bool again = npos > 1u && disc > 0.0f;
                        if (wave_any(again)) offer(sphere_hit_t(av, b, disc, again), 64u * (uint32_t)j + lane);
                    }
                }
            }
            PT_CSEC(1);
            unsigned long long best = ~0ull;
            for (unsigned long long hm = wave_ballot(key != ~0ull); hm != 0ull; hm &= hm - 1ull) {   // (one or two lanes, typically)
                const int l = __builtin_ctzll(hm);
                const unsigned long long kq = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), l) << 32) |
                                              (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, l);
                best = kq < best ? kq : best;
            }
            PT_CSEC(2);
            // ---- scene.rs:49-71 one level of ray_trace (the same expressions as the main loop's shading block, PAL flavour)
            bool scattered = false;
            if (best == ~0ull) {   // scene.rs:40-47
                const float4 psky = par[11];
                if (psky.w != 0.0f) {
                    V = mk3(psky.x, psky.y, psky.z);
                } else {
                    const float t = 0.5f * (d.y + 1.0f);
                    const float w1s = 1.0f - t;
                    V = mk3(w1s + (t * 0.5f) * 0.3f, w1s + (t * 0.7f) * 0.3f, w1s + (t * 1.0f) * 0.3f);
                }
            } else {
                const float t_hit = __uint_as_float((uint32_t)(best >> 32));
                const uint32_t idx = (uint32_t)best & 0xffffu;
                const float4 sp = sphere_at_m<MOVING>(mot, (int)idx, s_shade[4u * idx], rtime), q1 = s_shade[4u * idx + 1u], qa = s_shade[4u * idx + 2u], qb = s_shade[4u * idx + 3u];
                o = add3(o, scale3(d, t_hit));   // ray.rs:24-26
                const f3 point = o;
                const f3 normal = divs3_known(sub3(point, mk3(sp.x, sp.y, sp.z)), sp.w, qa.w);   // sphere.rs:42
                const uint32_t kind = __float_as_uint(q1.x), flags = __float_as_uint(q1.y);
                const float param = q1.w;
                f3 emitted = mk3(0.f, 0.f, 0.f);   // material.rs:161-167
                if (kind == PT_MAT_DIFFUSE_LIGHT) {
                    if (flags & kShadeConst) emitted = mk3(qa.x, qa.y, qa.z);
                    else if (flags & kShadeChecker2) emitted = checker_is_odd(10.0f * point.x, 10.0f * point.y, 10.0f * point.z) ? mk3(qa.x, qa.y, qa.z) : mk3(qb.x, qb.y, qb.z);
                    else emitted = texture_value(A.texs, pn, (int32_t)__float_as_uint(q1.z), point);
                }
                f3 att = mk3(1.f, 1.f, 1.f);   // (Dielectric, material.rs:117; the palette's white entry)
                if (depth < max_depth) {
                    f3 raw = d;
                    if (kind == PT_MAT_LAMBERTIAN) {   // material.rs:52-67 (the palette kernels only take Constant / two-Constant Checker albedos)
                        const f3 target = add3(add3(point, normal), random_unit_vector(rng));
                        const bool even = (flags & kShadeChecker2) && !checker_is_odd(10.0f * point.x, 10.0f * point.y, 10.0f * point.z);
                        att = even ? mk3(qb.x, qb.y, qb.z) : mk3(qa.x, qa.y, qa.z);
                        raw = sub3(target, point);
                        scattered = true;
                    } else if (kind == PT_MAT_METAL) {   // material.rs:69-89
                        const f3 reflected = reflect3(d, normal);
                        if (dot3(reflected, normal) > 0.0f) {
                            att = mk3(qa.x, qa.y, qa.z);
                            raw = add3(reflected, scale3(random_in_unit_sphere(rng), param));
                            scattered = true;
                        }
                    } else if (kind == PT_MAT_DIELECTRIC) {   // material.rs:91-124
                        const float ref_idx = param;
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
                            ni_over_nt = qb.y;   // 1.0 / ref_idx (f32, from the host)
                        }
                        f3 refracted;
                        bool use_refract = false;
                        if (refract3(d, outward_normal, ni_over_nt, refracted)) {
                            const float reflect_prob = qb.x + (1.0f - qb.x) * pow5_ref(1.0f - cosine);   // math.rs:76-80
                            if (rng_f32(rng) > reflect_prob) use_refract = true;
                        }
                        raw = use_refract ? refracted : reflect3(d, normal);
                        scattered = true;
                    }
                    if (scattered) d = normalize3(raw);
                }
                V = emitted;
                if (scattered) {
                    if (lane == depth) level = att;
                    depth += 1u;
                }
            }
            PT_CSEC(3);
            if (!scattered) break;
        }
        // scene.rs:62-64 unwound, innermost level first: emitted (= 0) + attenuation * deeper
        for (int k = (int)depth - 1; k >= 0; --k) {
            const int ks = __builtin_amdgcn_readfirstlane(k);
            const f3 c = mk3(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(level.x), ks)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(level.y), ks)),
                             __int_as_float(__builtin_amdgcn_readlane(__float_as_int(level.z), ks)));
            V = mk3(0.0f + c.x * V.x, 0.0f + c.y * V.y, 0.0f + c.z * V.z);
        }
        col = add3(col, V);   // scene.rs:110
        sample += 1u;
        PT_CSEC(4);
    }
#ifdef PT_COOPSEC
    if (lane == 0u) {
        for (int i = 0; i < 5; ++i) atomicAdd(&A.debug[80 + i], csec[i]);
        atomicAdd(&A.debug[86], (unsigned long long)(rays - rays_before));
        atomicAdd(&A.debug[87], 1ull);
    }
#endif
    // scene.rs:113-116
    const float4 pf = s_par[10];   // inv_ns, mix_prev, mix_new, prev_zero
    col = scale3(col, pf.x);
    if (lane == 0u) {
        float *out = A.rgb + ((pxy >> 16) * A.width + (pxy & 0xffffu)) * 3u;
        const bool pz = pf.w != 0.0f;
        const float p0 = pz ? 0.0f : out[0], p1 = pz ? 0.0f : out[1], p2 = pz ? 0.0f : out[2];
        out[0] = p0 * pf.y + col.x * pf.z;
        out[1] = p1 * pf.y + col.y * pf.z;
        out[2] = p2 * pf.y + col.z * pf.z;
        if (A.tile_cost) atomicAdd(&A.tile_cost[((pxy >> 16) >> kTileLog2) * A.tiles_x + ((pxy & 0xffffu) >> kTileLog2)], rays);
        atomicAdd(&A.debug[88], 1ull), atomicAdd(&A.debug[89], (unsigned long long)(rays - rays_before));   // pt_scene_coop_counters
#ifdef PT_DEVKNOBS
        if (A.wave_end) {
            const unsigned long long slot = atomicAdd(&A.debug[91], 1ull);
            if (slot < 130000ull) {
                unsigned long long *e = A.wave_end + 65536 + 4 * slot;
                e[0] = log_t0, e[1] = wall_clock64(), e[2] = (unsigned long long)(rays - rays_before) | ((unsigned long long)log_sample0 << 32), e[3] = pxy | ((unsigned long long)rays_before << 32);
            }
        }
#endif
    }
    wave_rays += (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(rays - rays_before));   // scene.rs:57 (the rays before the hand-over were counted by the wave that traced them)
}

// The wave that left its main loop last: nobody can hand anything over any more -- EXIT into every mailbox of the grid.
__device__ __forceinline__ void coop_broadcast_exit(const KArgs &A, uint32_t n_waves) {
    const uint64_t v = ((uint64_t)A.tail_gen << 2) | 1u;
    for (uint32_t w = threadIdx.x & 63u; w < n_waves; w += 64u) wt_store(A.tail_box + 16u * (size_t)w + 9, v);
}

// A wave that has left its main loop: trace handed-over pixels until the EXIT arrives. `me`: its number in the grid.
template <bool MOVING, bool GATED>
__device__ __forceinline__ void coop_worker(const KArgs &A, const GateSrc &G, const float4 *mot, const float4 *s_par, const float4 *sph, const float4 *s_shade,
                                            const PerlinLds &pn, uint32_t me, unsigned long long &wave_rays) {
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t *box = A.tail_box + 16u * (size_t)me;
    const uint64_t stamp = (uint64_t)A.tail_gen << 2;
    CoopSpheres mine;
    bool loaded = false;
    // A worker spends ~15x the instructions per ray of a per-lane wave: it is meant to use issue slots nobody else wants. The SIMD's
    // arbiter serves the higher priority first: waves still in their main loops run at priority kCoopBusyPrio, workers at 0
    // (config 3 +0.8 %, 256 spp +0.7 ... 1.6 %, the eighth of config 4 10.86 -> 10.64 ms; priority 3 measured the same: tools/shard_ab.sh).
    if (kCoopBusyPrio != 0) __builtin_amdgcn_s_setprio(0);
    for (;;) {
        uint32_t kind = 0u;
        if (lane == 0u) {
            if (wt_load(box + 9) == (stamp | 1u)) kind = 2u;   // the EXIT is already here: no need to go idle
            else wt_store(box + 8, stamp | kBoxIdle);
        }
        kind = (uint32_t)__builtin_amdgcn_readfirstlane((int)kind);
        if (kind == 2u) return;
        for (uint32_t nap = 1u;;) {
            uint32_t k = 0u;
            if (lane == 0u) {
                const uint64_t st = wt_load(box + 8), ex = wt_load(box + 9);
                if (st == (stamp | kBoxPixel)) k = 1u;
                else if (ex == (stamp | 1u)) {
                    // leave -- unless a normal wave of a workgroup that started late has claimed this mailbox meanwhile: its pixel
                    // is on the way and is traced first (the state word is taken back atomically, so no pixel can land in a dead box)
                    k = atomicCAS(reinterpret_cast<unsigned long long *>(box + 8), stamp | kBoxIdle, stamp) == (stamp | kBoxIdle) ? 2u : 0u;
                }
            }
            kind = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
            if (kind != 0u) break;
            // (4 096 ... 16 384 cycles between two looks: an idle worker's polls are instructions its SIMD's busy waves do not get --
            // with 1 024-cycle naps they were 8 % of ALL VALU instructions of a config-3 frame, each with one lane switched on:
            // tools/bbprof.py, profiles/r04_bbprof_lanes.txt)
            for (uint32_t i = 0; i < nap; ++i) __builtin_amdgcn_s_sleep(64);
#ifndef PT_COOP_NAP
#define PT_COOP_NAP 4u
#endif
            nap = nap < PT_COOP_NAP ? nap * 2u : PT_COOP_NAP;
        }
        if (kind == 2u) return;
        // The payload loads below must stay BEHIND the load that saw PIXEL. The hardware keeps them there (the writer drained its
        // write-through payload stores before it stored the flag, and a wave's vector loads of one kind return in order); the compiler is
        // told here: relaxed atomics to different addresses are otherwise only held back by the branch above, which it need not honour.
        asm volatile("" ::: "memory");
        if (!loaded) coop_load_spheres(mine, sph, A.n_spheres), loaded = true;
        coop_trace_pixel<MOVING, GATED>(A, G, mot, s_par, mine, sph, s_shade, pn, box, wave_rays);
    }
}

}  // namespace ptdev
