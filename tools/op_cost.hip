// op_cost.hip -- issue cost of single gfx950 instructions at the occupancy of the path-tracing kernels (development aid).
// For each instruction: 8 independent register chains, 4 waves per SIMD (1024 threads per CU), reported as SIMD cycles per
// wave-instruction relative to the measured v_mul_f32 rate and at the nominal 2.4 GHz. Build: hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s) {
    float a0 = threadIdx.x * 0.001f + 1.f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    unsigned long long q0 = threadIdx.x, q1 = q0 + 1, q2 = q0 + 2, q3 = q0 + 3;
    float b = s;
    unsigned u = threadIdx.x;
    typedef float f16v __attribute__((ext_vector_type(16)));
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    f16v m0 = {0}, m1 = {0}, m2 = {0}, m3 = {0};
    h8 ha = {(_Float16)1.f}, hb = {(_Float16)0.5f};
    for (int it = 0; it < iters; ++it) {
#define ONE(i) \
        if (OP == 0) asm volatile("v_nop"); \
        if (OP == 1) asm volatile("v_mov_b32 %0, %0" : "+v"(a##i)); \
        if (OP == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a##i) : "v"(b)); \
        if (OP == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a##i) : "v"(b)); \
        if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b)); \
        if (OP == 5) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a##i) : "v"(b)); \
        if (OP == 6) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a##i) : "v"(b)); \
        if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b)); \
        if (OP == 8) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a##i) : "v"(b)); \
        if (OP == 9) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##i)); \
        if (OP == 10) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a##i)); \
        if (OP == 11) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##i) : "v"(b)); \
        if (OP == 12) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##i)); \
        if (OP == 13) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a##i) : "v"(b) : "vcc"); \
        if (OP == 14) asm volatile("v_div_fmas_f32 %0, %0, %1, %0" : "+v"(a##i) : "v"(b)); \
        if (OP == 15) asm volatile("v_div_fixup_f32 %0, %0, %1, %0" : "+v"(a##i) : "v"(b)); \
        if (OP == 16) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a##i), "v"(b) : "vcc"); \
        if (OP == 17) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a##i) : "v"(u)); \
        if (OP == 18) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b)); \
        if (OP == 19) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a##i) : "v"(b)); \
        if (OP == 24) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a##i) : "v"(b) : "vcc"); \
        if (OP == 25) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(a##i) : "s20"); \
        if (OP == 26) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b)); \
        if (OP == 27) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a##i)); \
        if (OP == 28) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a##i) : "v"(b)); \
        if (OP == 29) asm volatile("s_nop 0"); \
        if (OP == 32) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(a##i)); \
        if (OP == 33) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b)); \
        if (OP == 34) asm volatile("v_cmp_lt_f32 s[22:23], %0, %1\n\tv_cndmask_b32 %0, %0, %1, s[22:23]" : "+v"(a##i) : "v"(b) : "s22", "s23"); \
        if (OP == 35) asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a##i) : "v"(b)); \
        if (OP == 36) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n\tv_mul_f32 %0, %0, %1" : "+v"(a##i) : "v"(b)); \
        if (OP == 50) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(a##i) : "v"(b)); \
        if (OP == 51) asm volatile("v_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1" : "+v"(a##i) : "v"(b)); \
        if (OP == 52) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a##i), "+v"(b)); \
        if (OP == 53) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(a##i)); \
        if (OP == 54) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a##i)); \
        if (OP == 55) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a##i)); \
        if (OP == 56) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b)); \
        if (OP == 57) asm volatile("v_lshrrev_b32 %0, 8, %0" : "+v"(a##i)); \
        if (OP == 58) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a##i) : "v"(b)); \
        if (OP == 59) asm volatile("v_pack_b32_f16 %0, %0, %1" : "+v"(a##i) : "v"(b)); \
        if (OP == 60) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(a##i) : "v"(b)); \
        if (OP == 61) asm volatile("v_mul_f32 %0, %0, %1\n\ts_nop 0" : "+v"(a##i) : "v"(b)); \
        if (OP == 63) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(a##i) : "v"(b)); \
        if (OP == 64) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a##i) : "v"(b)); \
        if (OP == 65) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a##i) : "v"(b)); \
        if (OP == 66) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a##i) : "v"(b));
        REP8(ONE)
#undef ONE
#define DBL(i) \
        if (OP == 20) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(d##i)); \
        if (OP == 21) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(q##i) : "v"(u) : "vcc"); \
        if (OP == 22) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d##i) : "v"(b)); \
        if (OP == 23) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d##i)); \
        if (OP == 40) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(d##i)); \
        if (OP == 41) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(d##i)); \
        if (OP == 42) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(d##i)); \
        if (OP == 43) asm volatile("v_lshl_add_u64 %0, %0, 0, %0" : "+v"(d##i)); \
        if (OP == 44) asm volatile("v_lshlrev_b64 %0, 17, %0" : "+v"(d##i)); \
        if (OP == 45) asm volatile("v_mov_b64 %0, %0" : "+v"(d##i)); \
        if (OP == 46) asm volatile("v_lshrrev_b64 %0, 19, %0" : "+v"(d##i));
        DBL(0) DBL(1) DBL(2) DBL(3) DBL(0) DBL(1) DBL(2) DBL(3)
#undef DBL
        if (OP == 30) {
            m0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, m0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, m1, 0, 0, 0);
            m2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, m2, 0, 0, 0);
            m3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, m3, 0, 0, 0);
            m0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, m0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, m1, 0, 0, 0);
            m2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, m2, 0, 0, 0);
            m3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, m3, 0, 0, 0);
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3) + (float)(q0 + q1 + q2 + q3) + m0[0] + m1[1] + m2[2] + m3[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
float run(float *d, int cus, int bpc, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<OP>, dim3(cus * bpc), dim3(256), 0, 0, d, iters, 1.0000001f);   // warm the clocks
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(cus * bpc), dim3(256), 0, 0, d, iters, 1.0000001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float *d;
    hipMalloc(&d, 256 * 4096 * 4);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, bpc = 4, iters = 8000;
    struct Row { const char *name; float ms; int per_iter; };
    std::vector<Row> rows;
#define RUN(op, name, n) rows.push_back(Row{name, run<op>(d, cus, bpc, iters), n}); fprintf(stderr, "%s %.3f\n", name, rows.back().ms);
    RUN(2, "v_mul_f32", 8) RUN(0, "v_nop", 8) RUN(1, "v_mov_b32", 8) RUN(3, "v_add_f32", 8) RUN(19, "v_sub_f32", 8) RUN(4, "v_fma_f32", 8)
    RUN(5, "v_xor_b32", 8) RUN(6, "v_alignbit_b32", 8) RUN(7, "v_cndmask_b32 (vcc)", 8) RUN(8, "v_lshl_add_u32", 8) RUN(26, "v_and_or_b32", 8)
    RUN(18, "v_max3_f32", 8) RUN(27, "v_cvt_f32_u32", 8) RUN(9, "v_rcp_f32", 8) RUN(10, "v_sqrt_f32", 8) RUN(11, "v_mul_lo_u32", 8)
    RUN(12, "v_mov_b32_dpp", 8) RUN(13, "v_div_scale_f32", 8) RUN(14, "v_div_fmas_f32", 8) RUN(15, "v_div_fixup_f32", 8) RUN(16, "v_cmp_lt_f32", 8)
    RUN(24, "v_add_co + v_addc (pair)", 8) RUN(25, "v_readlane_b32", 8) RUN(17, "ds_bpermute_b32 (+wait)", 8)
    RUN(28, "v_cndmask_b32 (sgpr pair)", 8) RUN(29, "s_nop 0", 8) RUN(32, "v_bfe_u32", 8) RUN(33, "v_perm_b32", 8)
    RUN(34, "v_cmp + v_cndmask (pair)", 8) RUN(35, "v_mul + v_add (pair)", 8) RUN(36, "v_cndmask(vcc) + v_mul (pair)", 8)
    RUN(50, "v_bitop3_b32", 8) RUN(51, "v_xor + v_xor (pair)", 8) RUN(52, "v_permlane32_swap_b32", 8) RUN(53, "v_cvt_f16_f32", 8) RUN(54, "v_cvt_f32_f16", 8)
    RUN(55, "v_ffbl_b32", 8) RUN(56, "v_add3_u32", 8) RUN(57, "v_lshrrev_b32", 8) RUN(58, "v_and_b32", 8) RUN(59, "v_pack_b32_f16", 8) RUN(60, "v_cvt_pkrtz_f16_f32", 8)
    RUN(66, "v_cvt_pk_f16_f32", 8)
    RUN(61, "v_mul + s_nop (pair)", 8) RUN(63, "v_fma_f32 (inline const)", 8) RUN(64, "v_fmac_f32 (VOP2)", 8) RUN(65, "v_min_f32", 8)
    RUN(43, "v_lshl_add_u64", 8) RUN(44, "v_lshlrev_b64", 8) RUN(45, "v_mov_b64", 8) RUN(46, "v_lshrrev_b64", 8)
    RUN(40, "v_pk_mul_f32", 8) RUN(41, "v_pk_add_f32", 8) RUN(42, "v_pk_fma_f32", 8)
    RUN(20, "v_mul_f64", 8) RUN(23, "v_fma_f64", 8) RUN(22, "v_cvt_f64_f32", 8) RUN(21, "v_mad_u64_u32", 8) RUN(30, "v_mfma_f32_32x32x16_f16", 8)
    const double base = rows[0].ms / (double)(iters * rows[0].per_iter);
    printf("CUs %d, %d waves per SIMD, 8 independent chains; loop overhead included (~1 SALU pair per 8 instructions)\n", cus, bpc);
    for (const Row &r : rows) {
        const double per = r.ms / (double)(iters * r.per_iter);                         // ms per wave-instruction slot per wave
        const double cyc = per * 1e-3 * 2.4e9 / bpc;                                    // SIMD cycles per wave-instruction (or pair) @ 2.4 GHz nominal: bpc waves share a SIMD
        printf("%-28s %7.3f ms  %5.2f x v_mul_f32  %5.2f cycles/instr/SIMD @2.4GHz\n", r.name, r.ms, per / base, cyc);
    }
    return 0;
}
