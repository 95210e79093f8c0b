// developer tool: issue cost of the instruction kinds the traversal loops are made of — vector f32 / f64 / conversions, scalar ALU,
// exec-mask bookkeeping, cross-lane moves — alone and mixed, at 1 .. 8 waves per SIMD.  Prints cycles per wave-instruction per
// SIMD at the clock attribute (the sustained clock is lower; compare rows, not absolute values).
//    hipcc --offload-arch=gfx950 -O3 -o issue_rates issue_rates.hip && ./issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 2048
#define REP 16  // instructions (or groups) per loop iteration
template <int OP> __global__ void __launch_bounds__(256) k(float* out, float seed, int sseed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    int s0 = sseed, s1 = sseed + 1, s2 = sseed + 2, s3 = sseed + 3;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int r = 0; r < REP / 4; ++r) {
            if (OP == 0) asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
            if (OP == 1) asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
            if (OP == 2) asm volatile("v_mul_f32 %0, %0, %4\n s_add_u32 %2, %2, 1\n v_mul_f32 %1, %1, %4\n s_add_u32 %3, %3, 1" : "+v"(a0), "+v"(a1), "+s"(s0), "+s"(s1) : "v"(seed) : "scc");
            if (OP == 3) asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"((double)seed));
            if (OP == 4) asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
            if (OP == 5) asm volatile("v_max3_f32 %0, %0, %1, %4\n v_max3_f32 %1, %1, %2, %4\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %3, %3, %0, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
            if (OP == 6) asm volatile("v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2" : "+v"(d0), "+v"(d1) : "v"(d2));
            if (OP == 7) asm volatile("v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2\n v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2" : "+v"(d0), "+v"(d1) : "v"(d2));
            if (OP == 8) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");
            if (OP == 9) asm volatile("s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[22:23], vcc\n s_or_b64 exec, exec, s[22:23]" : : : "s20", "s21", "s22", "s23", "scc");
            if (OP == 10) asm volatile("ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"((threadIdx.x * 4 + 4) & 255));
            if (OP == 11) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n s_bcnt1_i32_b64 %2, s[20:21]\n v_cmp_lt_f32 s[22:23], %1, %0\n s_bcnt1_i32_b64 %3, s[22:23]" : "+v"(a0), "+v"(a1), "+s"(s0), "+s"(s1) : : "s20", "s21", "s22", "s23", "scc");
            if (OP == 12) asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
            if (OP == 13) asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));
            if (OP == 14) asm volatile("v_mul_f32 %0, %0, %4\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n s_add_u32 %2, %2, 1" : "+v"(a0), "+v"(a1), "+s"(s0), "+s"(s1) : "v"(seed) : "scc");  // 1 VALU : 3 SALU
            if (OP == 15) asm volatile("v_mul_f32_dpp %0, %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %1, %1, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %2, %2, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %3, %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
            if (OP == 16) asm volatile("v_sub_f32 %0, %0, %4\n v_sub_f32 %1, %1, %4\n v_min_f32 %2, %2, %0\n v_max_f32 %3, %3, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
            if (OP == 17) asm volatile("v_lshl_add_u32 %0, %0, 2, %1\n v_add_u32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_bfe_u32 %3, %0, 3, 5" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3) + (float)(s0 + s1 + s2 + s3);
}
template <int OP> void run(const char* name, float* out, double ghz, int valu_per_group, int salu_per_group) {
    printf("%-34s", name);
    for (int wps : {1, 2, 4, 6, 8}) {
        const int blocks = 256 * wps;  // one block of 4 waves per CU and per wave-per-SIMD
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 1);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double groups = (double)blocks * 4 * ITER * (REP / 4);  // wave-level groups of 4 instructions
        const double cyc_per_group = (ms * 1e-3 * ghz * 1e9) / (groups / (256.0 * 4));  // SIMD cycles per group
        printf("  %dw: %6.2f", wps, cyc_per_group / 4.0);
    }
    printf("   cycles per instruction per SIMD\n");
}
int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    int khz = 0;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz * 1e-6;
    printf("clock %.2f GHz (attribute)\n", ghz);
    run<0>("v_mul_f32", out, ghz, 4, 0);
    run<12>("v_fma_f32", out, ghz, 4, 0);
    run<16>("v_sub/min/max_f32", out, ghz, 4, 0);
    run<17>("v int (lshl_add, add, and, bfe)", out, ghz, 4, 0);
    run<5>("v_max3_f32", out, ghz, 4, 0);
    run<15>("v_mul_f32 dpp quad_perm", out, ghz, 4, 0);
    run<6>("v_pk_mul_f32", out, ghz, 4, 0);
    run<7>("v_pk_fma_f32", out, ghz, 4, 0);
    run<3>("v_mul_f64", out, ghz, 4, 0);
    run<4>("v_cvt_f64_f32", out, ghz, 4, 0);
    run<13>("v_cvt_f32_f64", out, ghz, 4, 0);
    run<8>("v_cmp + v_cndmask (vcc)", out, ghz, 4, 0);
    run<1>("s_add_u32", out, ghz, 0, 4);
    run<2>("v_mul_f32 : s_add_u32 1:1", out, ghz, 2, 2);
    run<14>("v_mul_f32 : s_add_u32 1:3", out, ghz, 1, 3);
    run<9>("s_and_saveexec + s_or exec", out, ghz, 0, 4);
    run<11>("v_cmp -> sgpr + s_bcnt1", out, ghz, 2, 2);
    run<10>("ds_bpermute_b32 x4 + wait", out, ghz, 0, 0);
    return 0;
}
