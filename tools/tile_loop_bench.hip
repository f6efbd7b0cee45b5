// tile_loop_bench.hip -- latency/throughput of the MFMA prefilter's tile loop in isolation (design input).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
union Frag { uint4 u; half8 h; };

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4 *afrag, int n_tiles, int iters, unsigned *out, long long *cyc) {
    extern __shared__ uint4 s_afrag[];
    __shared__ unsigned short queue[20 * 256];
    for (int i = threadIdx.x; i < n_tiles * 128; i += 256) s_afrag[i] = afrag[i];
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63;
    Frag b0[2], b1[2];
    for (int c = 0; c < 2; ++c) { b0[c].u = afrag[(c * 64 + lane + 7) % (n_tiles * 128)]; b1[c].u = afrag[(c * 64 + lane + 13) % (n_tiles * 128)]; }
    const float16v zero = {0};
    unsigned cnt0 = 0, cnt1 = 0, total = 0;
    long long t0 = clock64();
    for (int it = 0; it < ((MODE == 4) ? 0 : iters); ++it) {
        Frag a0, a1, n0, n1;
        a0.u = s_afrag[lane]; a1.u = s_afrag[64 + lane];
        for (int T = 0; T < n_tiles; ++T) {
            const int Tn = (T + 1 < n_tiles) ? T + 1 : T;
            n0.u = s_afrag[(Tn * 2) * 64 + lane]; n1.u = s_afrag[(Tn * 2 + 1) * 64 + lane];
            float16v acc0, acc1;
            if (MODE == 5) {
                for (int r = 0; r < 16; ++r) { acc0[r] = __uint_as_float(a0.u.x + r + T); acc1[r] = __uint_as_float(a1.u.y + r + it); }
                asm volatile("" : "+v"(acc0), "+v"(acc1));
            } else {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0.h, b0[0].h, zero, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0.h, b1[0].h, zero, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1.h, b0[1].h, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1.h, b1[1].h, acc1, 0, 0, 0);
            }
            a0 = n0; a1 = n1;
            unsigned m0 = 0, m1 = 0;
            if (MODE == 0 || MODE == 5) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { m0 = __builtin_amdgcn_alignbit(m0, __float_as_uint(acc0[r]), 31); m1 = __builtin_amdgcn_alignbit(m1, __float_as_uint(acc1[r]), 31); }
            } else if (MODE == 1) {   // 4-way trees
                unsigned p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int r = 0; r < 16; ++r) { p[r >> 2] = __builtin_amdgcn_alignbit(p[r >> 2], __float_as_uint(acc0[r]), 31); p[4 + (r >> 2)] = __builtin_amdgcn_alignbit(p[4 + (r >> 2)], __float_as_uint(acc1[r]), 31); }
                m0 = (p[0] << 12) | (p[1] << 8) | (p[2] << 4) | p[3];
                m1 = (p[4] << 12) | (p[5] << 8) | (p[6] << 4) | p[7];
            } else if (MODE == 3) {   // MFMA only: a single dependent VALU op per tile
                m0 = __float_as_uint(acc0[0]) & __float_as_uint(acc1[15]) & 0x80000000u;
            } else {                  // MODE 2: no extraction at all (MFMA + min reduce only)
                float mn = acc0[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mn = fminf(mn, fminf(acc0[r], acc1[r]));
                m0 = mn < -1e30f;
            }
            unsigned m = m0 | (m1 << 16);
            m &= (it == 123456) ? 0xffffffffu : 0u;   // candidates are rare: keep the loop but never enter it here
            while (__any(m != 0u)) {
                if (m != 0u) {
                    const unsigned bit = 31u - (unsigned)__builtin_clz(m);
                    m &= ~(1u << bit);
                    const unsigned set = bit >> 4, c = set ? cnt1 : cnt0;
                    if (c < 10) queue[(set * 10 + c) * 256 + tid] = (unsigned short)bit;
                    cnt0 += set ^ 1u; cnt1 += set;
                }
            }
            total += m0 + m1;
        }
    }
    if (MODE == 4) {
        total = 0;
        t0 = clock64();
        for (int it = 0; it < iters; ++it) {
            Frag a0, a1;
            a0.u = s_afrag[lane]; a1.u = s_afrag[64 + lane];
            float16v p0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0.h, b0[0].h, zero, 0, 0, 0);
            float16v p1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0.h, b1[0].h, zero, 0, 0, 0);
            p0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1.h, b0[1].h, p0, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1.h, b1[1].h, p1, 0, 0, 0);
            for (int T = 0; T < n_tiles; ++T) {
                const int Tn = (T + 1 < n_tiles) ? T + 1 : T;
                a0.u = s_afrag[(Tn * 2) * 64 + lane]; a1.u = s_afrag[(Tn * 2 + 1) * 64 + lane];
                float16v q0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0.h, b0[0].h, zero, 0, 0, 0);
                float16v q1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0.h, b1[0].h, zero, 0, 0, 0);
                q0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1.h, b0[1].h, q0, 0, 0, 0);
                q1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1.h, b1[1].h, q1, 0, 0, 0);
                unsigned m0 = 0, m1 = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) { m0 = __builtin_amdgcn_alignbit(m0, __float_as_uint(p0[r]), 31); m1 = __builtin_amdgcn_alignbit(m1, __float_as_uint(p1[r]), 31); }
                unsigned m = m0 | (m1 << 16);
                m &= (it == 123456) ? 0xffffffffu : 0u;
                while (__any(m != 0u)) { if (m != 0u) { const unsigned bit = 31u - (unsigned)__builtin_clz(m); m &= ~(1u << bit); queue[(bit & 15) * 256 + tid] = (unsigned short)bit; cnt0 += 1; } }
                total += m0 + m1;
                p0 = q0; p1 = q1;
            }
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * 256 + tid] = total + cnt0 + cnt1;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int n_tiles = 16, iters = 200;
    std::vector<uint16_t> h(n_tiles * 128 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (i * 37 % 512);   // positive f16 values ~1..1.5
    uint4 *d; unsigned *out; long long *cyc;
    (void)hipMalloc(&d, h.size() * 2); (void)hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, 4096 * 256 * 4); (void)hipMalloc(&cyc, 4096 * 8);
    for (int mode = 0; mode < 6; ++mode) for (int bpc : {1, 2, 3}) {
        const int grid = 256 * bpc;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), n_tiles * 2048, 0, d, n_tiles, iters, out, cyc);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), n_tiles * 2048, 0, d, n_tiles, iters, out, cyc);
            if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(grid), dim3(256), n_tiles * 2048, 0, d, n_tiles, iters, out, cyc);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), n_tiles * 2048, 0, d, n_tiles, iters, out, cyc);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), n_tiles * 2048, 0, d, n_tiles, iters, out, cyc);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), n_tiles * 2048, 0, d, n_tiles, iters, out, cyc);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        long long c0; (void)hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
        printf("mode %d waves/SIMD %d: %.3f ms, %.0f clock64 ticks per tile per wave (wall), %.1f ns per tile-wave throughput per SIMD\n", mode, bpc, ms,
               (double)c0 / (iters * n_tiles), ms * 1e6 / (iters * n_tiles) / bpc);
    }
    return 0;
}
