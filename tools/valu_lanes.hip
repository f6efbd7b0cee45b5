// valu_lanes.hip -- two small measurements behind bench.py's roofline block and csrc/pt_coop.h (development aid; gfx950).
//   valu_lanes cal L      every wave runs unfused v_mul / v_add chains with only its first L lanes switched on. Run under
//                         `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU`: the ratio
//                         SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64) must come out as L / 64 -- it is what bench.py reports as
//                         `valu_lane_utilisation` (tools/valu_lane_calibration.sh, profiles/r04_valu_lane_calibration.txt).
//   valu_lanes lone       ONE wave alone on the GPU: cycles per wave-instruction of dependent and independent chains of plain and of
//                         packed f32 operations (what a cooperative worker wave pays per instruction; s_memtime around the loop).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float float2v __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void cal(float *out, int iters, float s, int lanes) {
    float a[8], r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    if ((int)(threadIdx.x & 63) < lanes) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { a[i] = a[i] * s; a[i] = a[i] + s; }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) r += a[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>   // 0 dependent plain, 1 eight independent plain chains, 2 dependent packed, 3 eight independent packed chains
__global__ __launch_bounds__(64) void lone(float *out, unsigned long long *cycles, int iters, float s) {
    float a[8];
    float2v p[8], q = {s, s * 1.0001f};
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = float2v{a[i], a[i] + 0.5f}; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) { a[0] = a[0] * s; a[0] = a[0] + s; }
            if (MODE == 1) { a[i] = a[i] * s; a[i] = a[i] + s; }
            if (MODE == 2) { p[0] = p[0] * q; p[0] = p[0] + q; }
            if (MODE == 3) { p[i] = p[i] * q; p[i] = p[i] + q; }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i] + p[i].x + p[i].y;
    out[threadIdx.x] = r;
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
}

int main(int argc, char **argv) {
    float *d; unsigned long long *c;
    hipMalloc(&d, 256 * 4096 * 4); hipMalloc(&c, 8);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    if (argc >= 3 && !strcmp(argv[1], "cal")) {
        const int lanes = atoi(argv[2]);
        hipLaunchKernelGGL(cal, dim3(prop.multiProcessorCount * 4), dim3(256), 0, 0, d, 4000, 1.0000001f, lanes);
        hipDeviceSynchronize();
        printf("cal: %d of 64 lanes active\n", lanes);
        return 0;
    }
    const char *names[4] = {"plain, one dependent chain", "plain, eight independent chains", "packed, one dependent chain", "packed, eight independent chains"};
    for (int mode = 0; mode < 4; ++mode) {
        const int iters = 4000;
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(lone<0>, dim3(1), dim3(64), 0, 0, d, c, iters, 1.0000001f);
            if (mode == 1) hipLaunchKernelGGL(lone<1>, dim3(1), dim3(64), 0, 0, d, c, iters, 1.0000001f);
            if (mode == 2) hipLaunchKernelGGL(lone<2>, dim3(1), dim3(64), 0, 0, d, c, iters, 1.0000001f);
            if (mode == 3) hipLaunchKernelGGL(lone<3>, dim3(1), dim3(64), 0, 0, d, c, iters, 1.0000001f);
            hipDeviceSynchronize();
        }
        unsigned long long cyc = 0;
        hipMemcpy(&cyc, c, 8, hipMemcpyDeviceToHost);
        // s_memtime counts at 100 MHz on gfx950; report both raw ticks and per wave-instruction
        printf("lone wave, %-34s: %llu ticks for %d wave-instructions = %.3f ticks each\n", names[mode], cyc, iters * 16, (double)cyc / (iters * 16.0));
    }
    return 0;
}
