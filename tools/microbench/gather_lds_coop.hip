// developer tool (round 4): two questions the traversal kernels' memory path raises, answered in cycles per wave-level load
// instruction per CU at 6 waves per SIMD (24 waves per CU, 4 independent loads in flight per lane), next to gather_rates.hip:
//   (a) what do random 16 / 32 / 48-byte record reads cost when the records live in the block's LDS (ds_read_b128) instead of
//       behind the texture path (global_load_dwordx4 at one L1 cycle per lane)?
//   (b) does the texture address unit coalesce NEIGHBOURING lanes that name the same line?  Groups of G = 2 / 4 consecutive lanes
//       read consecutive 16-byte pieces of one random record (a 32-byte node by a lane pair, a 64-byte record by a quad): if a
//       group costs one tag lookup, a node fetch shared out over a lane pair halves the L1 cycles of a node step.
//    hipcc --offload-arch=gfx950 -O3 -o gather_lds_coop gather_lds_coop.hip && ./gather_lds_coop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 512

// (b) global loads, 16 bytes per lane, lanes in groups of G on one record of G * 16 bytes (G = 1: every lane its own line)
template <int G>
__global__ void __launch_bounds__(256) k_coop(const uint4* __restrict__ buf, uint32_t mask, float* out) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t x = (blockIdx.x * 256 + (threadIdx.x & ~(uint32_t)(G - 1))) * 2654435761u + 12345u;  // one stream per group
    float acc = 0.0f;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            x = x * 1664525u + 1013904223u;
            const uint32_t line = (x >> 7) & mask;
            const char* p = reinterpret_cast<const char*>(buf) + (size_t)line * 128u + (lane & (uint32_t)(G - 1)) * 16u;
            const float4 v = *reinterpret_cast<const float4*>(p);
            acc += v.x + v.w;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// (a) LDS: records of REC bytes at random record-aligned places of a `bytes`-sized table in the block's dynamic LDS
template <int REC>
__global__ void __launch_bounds__(256) k_lds(uint32_t recs, float* out) {
    extern __shared__ float4 tab[];
    for (uint32_t i = threadIdx.x; i < recs * (REC / 16); i += 256) tab[i] = make_float4((float)i, 1.0f, 2.0f, 3.0f);
    __syncthreads();
    uint32_t x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            x = x * 1664525u + 1013904223u;
            const uint32_t r = __umulhi(x, recs) * (REC / 16);
            const float4 a = tab[r];
            acc += a.x + a.w;
            if (REC >= 32) { const float4 b = tab[r + 1]; acc += b.y; }
            if (REC >= 48) { const float4 c = tab[r + 2]; acc += c.z; }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

static double time_ms(void (*launch)()) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    launch();
    (void)hipEventRecord(e0);
    launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
static const uint4* g_buf;
static float* g_out;
static uint32_t g_mask, g_recs;
static size_t g_lds;
static const int kBlocks = 256 * 6;
template <int G> void launch_coop() { hipLaunchKernelGGL((k_coop<G>), dim3(kBlocks), dim3(256), 0, 0, g_buf, g_mask, g_out); }
template <int REC> void launch_lds() { hipLaunchKernelGGL((k_lds<REC>), dim3(kBlocks), dim3(256), g_lds, 0, g_recs, g_out); }

int main() {
    const size_t total = 1ull << 30;
    uint4* buf;
    (void)hipMalloc(&buf, total);
    (void)hipMemset(buf, 0, total);
    (void)hipMalloc(&g_out, kBlocks * 256 * sizeof(float));
    g_buf = buf;
    int khz = 0;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz * 1e-6;
    const double wave_loads_per_cu = (double)kBlocks * 4 * ITER * 4 / 256.0;
    printf("clock %.2f GHz (attribute); cycles per wave-level load instruction per CU, 24 waves per CU\n", ghz);
    printf("(b) global_load_dwordx4, G consecutive lanes on consecutive 16-byte pieces of one random line\n");
    for (size_t region : {(size_t)16 << 10, (size_t)2 << 20, (size_t)64 << 20, (size_t)1 << 30}) {
        g_mask = (uint32_t)(region / 128u) - 1u;
        const double m1 = time_ms(launch_coop<1>), m2 = time_ms(launch_coop<2>), m4 = time_ms(launch_coop<4>), m8 = time_ms(launch_coop<8>);
        printf("  region %8zu KiB:  G=1 %7.1f   G=2 %7.1f   G=4 %7.1f   G=8 %7.1f\n", region >> 10, m1 * 1e-3 * ghz * 1e9 / wave_loads_per_cu,
               m2 * 1e-3 * ghz * 1e9 / wave_loads_per_cu, m4 * 1e-3 * ghz * 1e9 / wave_loads_per_cu, m8 * 1e-3 * ghz * 1e9 / wave_loads_per_cu);
    }
    printf("(a) ds_read_b128 of random records in the block's LDS (table bytes; cycles per ds_read instruction per CU, and per record)\n");
    for (size_t bytes : {(size_t)2 << 10, (size_t)8 << 10, (size_t)16 << 10}) {
        g_lds = bytes;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 << 10);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 << 10);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds<48>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 << 10);
        g_recs = (uint32_t)(bytes / 16);
        const double a = time_ms(launch_lds<16>);
        g_recs = (uint32_t)(bytes / 32);
        const double b = time_ms(launch_lds<32>);
        g_recs = (uint32_t)(bytes / 48);
        const double c = time_ms(launch_lds<48>);
        printf("  table %3zu KiB:  16 B %6.1f / instr (%6.1f / record)   32 B %6.1f (%6.1f)   48 B %6.1f (%6.1f)\n", bytes >> 10,
               a * 1e-3 * ghz * 1e9 / wave_loads_per_cu, a * 1e-3 * ghz * 1e9 / wave_loads_per_cu, b * 1e-3 * ghz * 1e9 / wave_loads_per_cu / 2,
               b * 1e-3 * ghz * 1e9 / wave_loads_per_cu, c * 1e-3 * ghz * 1e9 / wave_loads_per_cu / 3, c * 1e-3 * ghz * 1e9 / wave_loads_per_cu);
    }
    return 0;
}
