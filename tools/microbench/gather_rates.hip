// developer tool: what a wave pays for loads whose 64 lanes each name a different cache line (the node fetches of a BVH walk) —
// cycles per wave-level load instruction per CU, by load width, by the size of the region the addresses fall in (L1 / L2 /
// Infinity Cache / HBM), at 6 waves per SIMD, 4 independent loads in flight per lane.
//    hipcc --offload-arch=gfx950 -O3 -o gather_rates gather_rates.hip && ./gather_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 512
template <int W, int SAME> __global__ void __launch_bounds__(256) k(const uint4* __restrict__ buf, uint32_t mask, float* out, uint32_t lanes_on) {
    uint32_t x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    if ((threadIdx.x & 63u) >= lanes_on) { out[blockIdx.x * 256 + threadIdx.x] = 0; return; }
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            x = x * 1664525u + 1013904223u;
            const uint32_t line = (x >> 7) & mask;  // a 128-byte line of the region
            const char* p = reinterpret_cast<const char*>(buf) + (size_t)line * 128u;
            if (W == 4) acc += *reinterpret_cast<const float*>(p);
            if (W == 8) { float2 v = *reinterpret_cast<const float2*>(p); acc += v.x + v.y; }
            if (W == 16) { float4 v = *reinterpret_cast<const float4*>(p); acc += v.x + v.w; }
            if (W == 32) { float4 v = *reinterpret_cast<const float4*>(p), w = *reinterpret_cast<const float4*>(p + 16); acc += v.x + w.w; }  // two loads, one 32-byte record
            if (W == 64) { float4 v = *reinterpret_cast<const float4*>(p), w = *reinterpret_cast<const float4*>(p + 16), y = *reinterpret_cast<const float4*>(p + 32), z = *reinterpret_cast<const float4*>(p + 48); acc += v.x + w.w + y.y + z.z; }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int W> void run(const uint4* buf, float* out, double ghz, size_t region_bytes, uint32_t lanes_on) {
    const uint32_t mask = (uint32_t)(region_bytes / 128u) - 1u;
    const int blocks = 256 * 6;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<W, 0>), dim3(blocks), dim3(256), 0, 0, buf, mask, out, lanes_on);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<W, 0>), dim3(blocks), dim3(256), 0, 0, buf, mask, out, lanes_on);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const int per = W <= 16 ? 1 : W / 16;
    const double wave_loads_per_cu = (double)blocks * 4 * ITER * 4 * per / 256.0;
    printf("  %2d B x %2u lanes: %7.1f cyc/load-instr/CU (%5.2f per lane)", W, lanes_on, ms * 1e-3 * ghz * 1e9 / wave_loads_per_cu, ms * 1e-3 * ghz * 1e9 / wave_loads_per_cu / lanes_on);
}
int main() {
    const size_t total = 1ull << 30;
    uint4* buf;
    float* out;
    (void)hipMalloc(&buf, total);
    (void)hipMemset(buf, 0, total);
    (void)hipMalloc(&out, 256 * 6 * 256 * sizeof(float));
    int khz = 0;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz * 1e-6;
    printf("clock %.2f GHz (attribute); cycles per wave-level load instruction per CU (24 waves per CU)\n", ghz);
    for (size_t region : {(size_t)16 << 10, (size_t)2 << 20, (size_t)64 << 20, (size_t)1 << 30}) {
        printf("region %8zu KiB\n", region >> 10);
        for (uint32_t lanes : {64u, 32u, 16u}) {
            run<4>(buf, out, ghz, region, lanes); run<8>(buf, out, ghz, region, lanes); run<16>(buf, out, ghz, region, lanes); printf("\n");
            run<32>(buf, out, ghz, region, lanes); run<64>(buf, out, ghz, region, lanes); printf("\n");
        }
    }
    return 0;
}
