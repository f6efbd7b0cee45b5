// pt_query.hip -- closest-hit QUERIES on explicit rays (pt_closest_hit): the unit the reference's own #[bench] functions time
// (bench.rs:8-26: one `ray_hit` on the centre ray of a preset -- hitable_list.rs:68-75, spheres_soa.rs:464-485, bvh.rs:361-379) and the
// home of SURVEY 8 row a7, SpheresSoA (collision/spheres_soa.rs:12-392), which nothing but those benches ever calls. Not on the render
// path: the frame kernels answer the same question for their own rays with other machinery (pt_kernel.h); here every mode is the
// reference's algorithm as written, one ray per lane, so that a query's RESULT (t, entry, point, normal) can be compared bit for bit
// with the oracle's restatement of the same function and its TIME with the reference's benchmark definition.
//   PT_QUERY_LIST        HitableList::ray_hit (hitable_list.rs:40-56) over Sphere / MovingSphere::ray_hit (sphere.rs:29-66, moving_sphere.rs:38-73)
//   PT_QUERY_BVH         BVHNode::ray_hit over the CALLER's tree (bvh.rs:37-62, aabb.rs:46-58), as an explicit-stack walk: both children
//                        with the original t_max, `lhs.t < rhs.t ? lhs : rhs` == of the hits with the smallest t the DFS-last one
//   PT_QUERY_SOA_SCALAR  SpheresSoA::hit_scalar (spheres_soa.rs:105-155)
//   PT_QUERY_SOA_SSE4_1 / _AVX2   hit_sse4_1 / hit_avx2 (:161-268, :274-391): 4 / 8 running minima by lane, horizontal minimum, lowest lane
// SpheresSoA's arithmetic is NOT Sphere::ray_hit's: co = centre - origin, discriminant = nb*nb - c with no `a` (it assumes a unit
// direction), t = nb -+ sqrt with no division, normal = (p - c) * (1 / r). An image rendered through it would differ from the
// reference's own (the live path is sphere.rs), which is why it is a query mode and not a render mode.
#include "pt_device.h"
#include "pt_host.h"

namespace ptdev {

struct QArgs {
    const float4 *spheres;      // cx, cy, cz, radius (list order)
    const float4 *spheres_r2;   // cx, cy, cz, radius * radius (sphere.rs:36, rounded once on the host)
    const float4 *motion;       // MovingSphere records (pt_args.h KArgs::motion) or nullptr
    const pt_bvh_node *nodes;   // the caller's tree (PT_QUERY_BVH)
    int32_t root;
    uint32_t n_spheres, n_rays;
    const float *rays;          // [n_rays][7] origin, direction, time
    float t_min, t_max;
    float *hits;                // [n_rays][8] t, entry (u32 bits; 0xffffffff: miss, 0xfffffffe: traversal stack exhausted), point, normal
};

constexpr int kQueryBlock = 128, kQueryStack = 48;

__device__ __forceinline__ f3 centre_at(const QArgs &A, uint32_t k, float4 c, float time) {
    if (A.motion) {   // moving_sphere.rs:29-31
        const float4 m0 = A.motion[2 * k], m1 = A.motion[2 * k + 1];
        if (m1.y != 0.0f) {
            const float s = (time - m1.x) * m0.w;
            return mk3(c.x + s * m0.x, c.y + s * m0.y, c.z + s * m0.z);
        }
    }
    return mk3(c.x, c.y, c.z);
}

// sphere.rs:29-66 / moving_sphere.rs:38-73: true when the entry is hit in (t_min, t_max); t_out = the accepted root
__device__ __forceinline__ bool sphere_query(f3 o, f3 d, float a, f3 centre, float r2, float t_min, float t_max, float &t_out) {
    const f3 oc = sub3(o, centre);
    const float b = dot3(oc, d);
    const float c = dot3(oc, oc) - r2;
    const float disc = b * b - a * c;
    if (disc > 0.0f) {
        const float sq = __builtin_sqrtf(disc);
        float t = (-b - sq) / a;
        if (t < t_max && t > t_min) { t_out = t; return true; }
        t = (-b + sq) / a;
        if (t < t_max && t > t_min) { t_out = t; return true; }
    }
    return false;
}

// aabb.rs:46-58 with the SSE min / max NaN rule (second operand)
__device__ __forceinline__ bool aabb_query(const pt_bvh_node &n, f3 o, f3 rcp, float t_min, float t_max) {
    const auto mn = [](float x, float y) { return x < y ? x : y; };
    const auto mx = [](float x, float y) { return x > y ? x : y; };
    const float ax = (n.min[0] - o.x) * rcp.x, ay = (n.min[1] - o.y) * rcp.y, az = (n.min[2] - o.z) * rcp.z;
    const float bx = (n.max[0] - o.x) * rcp.x, by = (n.max[1] - o.y) * rcp.y, bz = (n.max[2] - o.z) * rcp.z;
    const float lox = mx(mn(ax, bx), t_min), loy = mx(mn(ay, by), t_min), loz = mx(mn(az, bz), t_min);
    const float hix = mn(mx(ax, bx), t_max), hiy = mn(mx(ay, by), t_max), hiz = mn(mx(az, bz), t_max);
    return hix > lox && hiy > loy && hiz > loz;
}

template <int MODE>
__global__ __launch_bounds__(kQueryBlock) void pt_query_kernel(const QArgs A) {
    __shared__ int32_t s_stack[kQueryStack * kQueryBlock];
    const uint32_t i = blockIdx.x * kQueryBlock + threadIdx.x;
    if (i >= A.n_rays) return;
    const float *R = A.rays + 7 * (size_t)i;
    const f3 o = mk3(R[0], R[1], R[2]), d = mk3(R[3], R[4], R[5]);
    const float time = R[6], t_min = A.t_min, t_max = A.t_max;
    float best_t = t_max;
    uint32_t best = 0xffffffffu;
    bool soa_normal = false;
    if (MODE == PT_QUERY_LIST) {   // hitable_list.rs:40-56: in list order, each entry asked with t_max = closest_so_far
        const float a = dot3(d, d);
        for (uint32_t k = 0; k < A.n_spheres; ++k) {
            const float4 c = A.spheres_r2[k];
            float t;
            if (sphere_query(o, d, a, centre_at(A, k, c, time), c.w, t_min, best_t, t)) best_t = t, best = k;
        }
    } else if (MODE == PT_QUERY_BVH) {
        const float a = dot3(d, d);
        const f3 rcp = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);   // ray.rs:14
        int sp = 0;
        int32_t *st = s_stack + threadIdx.x;
        if (A.root >= 0) st[(sp++) * kQueryBlock] = A.root;
        while (sp > 0) {
            const int32_t ref = st[(--sp) * kQueryBlock];
            if (ref < 0) {   // a Hitable::Sphere / MovingSphere leaf: asked with the ORIGINAL t_max (bvh.rs:44-46)
                const uint32_t k = (uint32_t)~ref;
                const float4 c = A.spheres_r2[k];
                float t;
                // bvh.rs:47-53 `lhs.t < rhs.t ? lhs : rhs` over the whole tree: the smallest t, and of equal ones the DFS-later leaf
                if (sphere_query(o, d, a, centre_at(A, k, c, time), c.w, t_min, t_max, t) && (best == 0xffffffffu || t <= best_t)) best_t = t, best = k;
            } else {
                const pt_bvh_node n = A.nodes[ref];
                if (aabb_query(n, o, rcp, t_min, t_max)) {
                    if (sp + 2 > kQueryStack) { best = 0xfffffffeu; break; }
                    st[(sp++) * kQueryBlock] = n.rhs;   // lhs is walked first
                    st[(sp++) * kQueryBlock] = n.lhs;
                }
            }
        }
    } else if (MODE == PT_QUERY_SOA_SCALAR) {   // spheres_soa.rs:105-155
        for (uint32_t k = 0; k < A.n_spheres; ++k) {
            const float4 c = A.spheres_r2[k];
            const f3 co = sub3(mk3(c.x, c.y, c.z), o);
            const float nb = dot3(co, d);
            const float cc = dot3(co, co) - c.w;
            const float disc = nb * nb - cc;
            if (disc > 0.0f) {
                const float sq = __builtin_sqrtf(disc);
                float t = nb - sq;
                if (t < t_min) t = nb + sq;
                if (t > t_min && t < best_t) best_t = t, best = k;
            }
        }
        soa_normal = true;
    } else {   // spheres_soa.rs:161-268 / :274-391: lane l of the reference's vector holds entries l, l + L, ...
        constexpr uint32_t L = MODE == PT_QUERY_SOA_SSE4_1 ? 4u : 8u;
        float hit_t[L];
        uint32_t hit_k[L];
#pragma unroll
        for (uint32_t l = 0; l < L; ++l) hit_t[l] = t_max, hit_k[l] = 0xffffffffu;
        for (uint32_t base = 0; base < A.n_spheres; base += L) {
#pragma unroll
            for (uint32_t l = 0; l < L; ++l) {
                const uint32_t k = base + l;
                if (k < A.n_spheres) {   // (the padding entries of spheres_soa.rs:55-62 can never hit: their discriminant is -inf or NaN)
                    const float4 c = A.spheres_r2[k];
                    const float cox = c.x - o.x, coy = c.y - o.y, coz = c.z - o.z;
                    const float nb = (cox * d.x + coy * d.y) + coz * d.z;
                    const float cc = ((cox * cox + coy * coy) + coz * coz) - c.w;
                    const float disc = nb * nb - cc;
                    if (disc > 0.0f) {
                        const float sq = __builtin_sqrtf(disc);
                        const float t0 = nb - sq, t1 = nb + sq;
                        const float t = t0 > t_min ? t0 : t1;
                        if (t > t_min && t < hit_t[l]) hit_t[l] = t, hit_k[l] = k;
                    }
                }
            }
        }
        float m = hit_t[0];
#pragma unroll
        for (uint32_t l = 1; l < L; ++l) m = hit_t[l] < m ? hit_t[l] : m;
        if (m < t_max) {
#pragma unroll
            for (uint32_t l = L; l-- > 0;)   // (descending, so the LOWEST lane holding the minimum is the one kept: :232-236, :355-359)
                if (hit_t[l] == m) best_t = m, best = hit_k[l];
        }
        soa_normal = true;
    }
    float *H = A.hits + 8 * (size_t)i;
    H[0] = best_t;
    H[1] = __uint_as_float(best);
    if (best < 0xfffffffeu) {
        const float4 s = A.spheres[best];
        const f3 p = add3(o, scale3(d, best_t));   // ray.rs:24-26
        const f3 centre = soa_normal ? mk3(s.x, s.y, s.z) : centre_at(A, best, s, time);
        const f3 rel = sub3(p, centre);
        const volatile float one = 1.0f;
        const f3 nrm = soa_normal ? scale3(rel, one / s.w) : divs3(rel, s.w);   // (p - c) * radius_inv (spheres_soa.rs:41,142) | (p - c) / radius (sphere.rs:42)
        H[2] = p.x, H[3] = p.y, H[4] = p.z, H[5] = nrm.x, H[6] = nrm.y, H[7] = nrm.z;
    } else {
        H[2] = H[3] = H[4] = H[5] = H[6] = H[7] = 0.0f;
    }
}

}  // namespace ptdev

using namespace pthostside;

extern "C" int pt_closest_hit(pt_scene *s, uint32_t mode, uint32_t n_rays, const float *d_rays7, float t_min, float t_max, float *d_hits8, void *hip_stream) {
    if (!s || !d_rays7 || !d_hits8) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    if (s->tr.is_world || !s->d_spheres) return fail(PT_ERR_UNSUPPORTED, "closest-hit queries take Sphere / MovingSphere worlds");
    if (mode > PT_QUERY_SOA_AVX2) return fail(PT_ERR_INVALID_ARG, "unknown query mode %u", mode);
    if (mode >= PT_QUERY_SOA_SCALAR && s->tr.has_motion) return fail(PT_ERR_UNSUPPORTED, "SpheresSoA holds Hitable::Sphere entries only (spheres_soa.rs:52 panics on anything else)");
    if (mode == PT_QUERY_BVH && (s->bvh_root < 0 || !s->d_ref_nodes)) return fail(PT_ERR_UNSUPPORTED, "the scene was created without BVH nodes");
    if (n_rays == 0) return PT_OK;
    HIP_TRY(hipSetDevice(s->device));
    ptdev::QArgs A;
    A.spheres = s->d_spheres, A.spheres_r2 = s->d_spheres_r2, A.motion = s->tr.has_motion ? s->d_motion : nullptr;
    A.nodes = s->d_ref_nodes, A.root = s->bvh_root, A.n_spheres = s->tr.n_spheres, A.n_rays = n_rays;
    A.rays = d_rays7, A.t_min = t_min, A.t_max = t_max, A.hits = d_hits8;
    hipStream_t stream = reinterpret_cast<hipStream_t>(hip_stream);
    const dim3 grid((n_rays + ptdev::kQueryBlock - 1) / ptdev::kQueryBlock), block(ptdev::kQueryBlock);
    switch (mode) {
    case PT_QUERY_LIST: hipLaunchKernelGGL(ptdev::pt_query_kernel<PT_QUERY_LIST>, grid, block, 0, stream, A); break;
    case PT_QUERY_BVH: hipLaunchKernelGGL(ptdev::pt_query_kernel<PT_QUERY_BVH>, grid, block, 0, stream, A); break;
    case PT_QUERY_SOA_SCALAR: hipLaunchKernelGGL(ptdev::pt_query_kernel<PT_QUERY_SOA_SCALAR>, grid, block, 0, stream, A); break;
    case PT_QUERY_SOA_SSE4_1: hipLaunchKernelGGL(ptdev::pt_query_kernel<PT_QUERY_SOA_SSE4_1>, grid, block, 0, stream, A); break;
    default: hipLaunchKernelGGL(ptdev::pt_query_kernel<PT_QUERY_SOA_AVX2>, grid, block, 0, stream, A); break;
    }
    HIP_TRY(hipGetLastError());
    return PT_OK;
}
