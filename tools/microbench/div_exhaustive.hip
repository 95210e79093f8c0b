// Proof by exhaustion for the box test's quotient (device/traverse.h, slab_rs): for EVERY pair of f32 significands
//     n = 1.mn, d = 1.md   (2^23 x 2^23 pairs)
// the three-instruction sequence   q0 = RN(nn * nr);  e = RN(d * q0 + nn) [exact];  q = RN(e * nr + q0)
// with nn = -n and nr = -RN(1 / d) returns RN(n / d), the IEEE quotient, bit for bit.  Multiplication, fused multiply-add,
// division and rounding to nearest commute with scaling by powers of two as long as nothing overflows or underflows, and with
// the signs of n and d (up to the sign of a zero quotient, which no comparison of the box test sees), so the result holds for
// every n and d whose quotient, reciprocal and remainder stay normal — the guarded range of the division-free box test.
//    hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o div_exhaustive div_exhaustive.hip && ./div_exhaustive [first_md last_md]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
__global__ void __launch_bounds__(256) k(uint32_t md_first, uint32_t mn_first, uint32_t mn_count, unsigned long long* bad, uint32_t* sample) {
    const uint32_t md = md_first + blockIdx.x * 256u + threadIdx.x;
    const float d = __uint_as_float(0x3f800000u | md);
    const float nr = -(1.0f / d);  // correctly rounded: -fhip-fp32-correctly-rounded-divide-sqrt
    unsigned long long mine = 0;
    for (uint32_t i = 0; i < mn_count; ++i) {
        const float n = __uint_as_float(0x3f800000u | (mn_first + i));
        const float nn = -n;
        const float q0 = nn * nr;
        const float e = __builtin_fmaf(d, q0, nn);
        const float q = __builtin_fmaf(e, nr, q0);
        const float want = n / d;
        if (__float_as_uint(q) != __float_as_uint(want)) {
            if (mine == 0) {
                sample[0] = __float_as_uint(n);
                sample[1] = __float_as_uint(d);
            }
            ++mine;
        }
    }
    if (mine) atomicAdd(bad, mine);
}
int main(int argc, char** argv) {
    const uint32_t md0 = argc > 1 ? (uint32_t)strtoul(argv[1], nullptr, 0) : 0u, md1 = argc > 2 ? (uint32_t)strtoul(argv[2], nullptr, 0) : (1u << 23);
    unsigned long long* bad;
    uint32_t* sample;
    (void)hipMalloc(&bad, 8);
    (void)hipMalloc(&sample, 8);
    (void)hipMemset(bad, 0, 8);
    (void)hipMemset(sample, 0, 8);
    const uint32_t slice = 1u << 20;  // divisors per launch
    unsigned long long pairs = 0;
    for (uint32_t md = md0; md < md1; md += slice) {
        const uint32_t n_md = md1 - md < slice ? md1 - md : slice;
        hipLaunchKernelGGL(k, dim3(n_md / 256u), dim3(256), 0, 0, md, 0u, 1u << 23, bad, sample);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
        pairs += (unsigned long long)n_md << 23;
        unsigned long long b = 0;
        (void)hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost);
        printf("divisor significands %u .. %u done, %llu pairs so far, %llu mismatches\n", md, md + n_md - 1, pairs, b);
        fflush(stdout);
    }
    unsigned long long b = 0;
    uint32_t s[2];
    (void)hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(s, sample, 8, hipMemcpyDeviceToHost);
    printf("TOTAL %llu pairs, %llu mismatches", pairs, b);
    if (b) printf(" (one of them: n bits %08x, d bits %08x)", s[0], s[1]);
    printf("\n");
    return b ? 1 : 0;
}
