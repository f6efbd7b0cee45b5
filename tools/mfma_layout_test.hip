// mfma_layout_test.hip -- verifies the A/B/D lane mapping of v_mfma_f32_32x32x16_f16 on gfx950:
//   A[i][k]: lane l holds row i = l & 31, k = 8*(l >> 5) + e (e = 0..7)
//   B[k][j]: lane l holds col j = l & 31, k = 8*(l >> 5) + e
//   D[i][j]: lane l holds col j = l & 31, row i = (r & 3) + 8*(r >> 2) + 4*(l >> 5), r = 0..15
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

__global__ void k(const float *A, const float *B, float *D) {  // A[32][16], B[16][32], D[32][32]
    const int l = threadIdx.x;
    half8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (_Float16)A[(l & 31) * 16 + 8 * (l >> 5) + e];
        b[e] = (_Float16)B[(8 * (l >> 5) + e) * 32 + (l & 31)];
    }
    float16v c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

int main() {
    float hA[32 * 16], hB[16 * 32], hD[32 * 32], *dA, *dB, *dD;
    for (int i = 0; i < 32 * 16; ++i) hA[i] = (float)((i * 7 + 3) % 23) - 11.0f;          // asymmetric, exactly representable
    for (int i = 0; i < 16 * 32; ++i) hB[i] = (float)((i * 5 + 1) % 19) * 0.25f - 2.0f;
    (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dD, sizeof hD);
    (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double s = 0; for (int kk = 0; kk < 16; ++kk) s += (double)hA[i * 16 + kk] * hB[kk * 32 + j];
        if (fabs(s - hD[i * 32 + j]) > 1e-3) { if (bad < 5) printf("mismatch D[%d][%d] = %f want %f\n", i, j, hD[i * 32 + j], s); ++bad; }
    }
    printf("mfma_f32_32x32x16_f16 layout check: %d mismatches of 1024\n", bad);
    return bad != 0;
}
