// developer tool: issue rate of the VALU instructions the box test is made of, on the card it runs on (wave-instructions
// per cycle per SIMD; 0.25 = full rate for a wave64 on a 16-lane SIMD).   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
#define CH 8
template <int OP> __global__ void __launch_bounds__(256) k(float* out, float seed) {
    float a[CH];
    double d[CH];
    for (int c = 0; c < CH; ++c) a[c] = seed + c + threadIdx.x, d[c] = a[c];
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (OP == 0) a[c] = a[c] * 1.0001f;                                  // v_mul_f32
            if (OP == 1) d[c] = d[c] * 1.0001;                                   // v_mul_f64
            if (OP == 2) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[c]) : "v"(a[c])); }
            if (OP == 3) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[c]) : "v"(d[c])); }
            if (OP == 4) a[c] = __builtin_fminf(a[c], seed);                     // v_min_f32
            if (OP == 5) a[c] = __builtin_fmaf(a[c], 1.0001f, seed);             // v_fma_f32
            if (OP == 6) d[c] = __builtin_fma(d[c], 1.0001, 0.5);                // v_fma_f64
            if (OP == 7) { asm volatile("v_rcp_f32 %0, %1" : "=v"(a[c]) : "v"(a[c])); }
        }
    }
    float s = 0;
    for (int c = 0; c < CH; ++c) s += a[c] + (float)d[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char* name, float* out, double ghz) {
    const int blocks = 256 * 8;  // 8 blocks of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 4 * ITER * CH;
    const double per_simd_per_cycle = wave_instr / (256.0 * 4) / (ms * 1e-3 * ghz * 1e9);
    printf("%-14s %8.3f ms  %.3f wave-instr / cycle / SIMD (%.1f cycles each)\n", name, ms, per_simd_per_cycle, 1.0 / per_simd_per_cycle);
}
int main() {
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz * 1e-6;
    printf("clock %.2f GHz (attribute; the sustained clock may be lower)\n", ghz);
    run<0>("v_mul_f32", out, ghz);
    run<5>("v_fma_f32", out, ghz);
    run<4>("v_min_f32", out, ghz);
    run<1>("v_mul_f64", out, ghz);
    run<6>("v_fma_f64", out, ghz);
    run<2>("v_cvt_f64_f32", out, ghz);
    run<3>("v_cvt_f32_f64", out, ghz);
    run<7>("v_rcp_f32", out, ghz);
    return 0;
}
