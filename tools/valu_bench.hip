// valu_bench.hip -- VALU issue-rate microbenchmark for gfx950 (design input for the sphere scan).
// Measures lane-ops/s of: plain v_mul+v_add (no FMA), v_fma, v_pk_mul+v_pk_add, v_pk_fma with
// independent chains, at 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s) {
    float a[8], b = s;
    float2v p[8], q = {s, s * 1.0001f};
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = float2v{a[i], a[i] + 0.5f}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) { a[i] = a[i] * b; a[i] = a[i] + b; }                    // 2 ops (mul, add) unfused
            if (MODE == 1) { a[i] = __builtin_fmaf(a[i], b, b); a[i] = __builtin_fmaf(a[i], b, b); }  // 2 fma
            if (MODE == 2) { p[i] = p[i] * q; p[i] = p[i] + q; }                    // 2 pk ops = 4 lane-ops
            if (MODE == 3) { p[i] = __builtin_elementwise_fma(p[i], q, q); p[i] = __builtin_elementwise_fma(p[i], q, q); }
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
    float *d; hipMalloc(&d, 256 * 2048 * 4 * 8);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    printf("CUs %d clock %d kHz\n", cus, prop.clockRate);
    const char *names[4] = {"mul+add (unfused)", "fma", "pk_mul+pk_add", "pk_fma"};
    const int opsper[4] = {2, 2, 4, 4};  // lane-ops (instr-lanes x width) per inner statement pair
    for (int mode = 0; mode < 4; ++mode) {
        for (int bpc : {1, 2, 4, 8}) {
            int iters = 20000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(cus * bpc), dim3(256), 0, 0, d, iters, 1.0000001f);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(cus * bpc), dim3(256), 0, 0, d, iters, 1.0000001f);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(cus * bpc), dim3(256), 0, 0, d, iters, 1.0000001f);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(cus * bpc), dim3(256), 0, 0, d, iters, 1.0000001f);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double laneops = (double)cus * bpc * 256 * iters * 8.0 * opsper[mode];
            double instr = (double)cus * bpc * 4 * iters * 8.0 * 2;  // wave-instructions
            printf("%-20s waves/SIMD %d: %.2f T lane-ops/s, %.2f cycles/wave-instr/SIMD @2.4GHz (%.3f ms)\n", names[mode], bpc,
                   laneops / ms / 1e9, 2.4e9 * (ms * 1e-3) / (instr / (cus * 4.0)), ms);
        }
    }
    return 0;
}
