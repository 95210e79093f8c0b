/* tests/slab_filter_check.c — the conservative box-test filter of include/pbrs_numeric.h against the reference's test
 * (geometry/src/bvh.rs:84-99 with correctly rounded f32 divisions) on random and adversarial boxes and rays inside the guarded
 * range of the division-free test.  Prints: cases, exact passes, filter passes, violations (exact pass without filter pass). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "../include/pbrs_numeric.h"

static uint64_t s = 0x9e3779b97f4a7c15ULL;
static uint32_t u32(void) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (uint32_t)(s >> 32); }
static float uni(void) { return (float)(u32() >> 8) * 5.9604644775390625e-8f; }
static float range(float a, float b) { return a + (b - a) * uni(); }
static float mn(float a, float b) { return a < b ? a : b; }
static float mx(float a, float b) { return a > b ? a : b; }
static int exact(const float* bmin, const float* bmax, const float* o, const float* d, float t_max) {
    float lo = -INFINITY, hi = INFINITY;
    for (int a = 0; a < 3; ++a) {
        float t0 = (bmin[a] - o[a]) / d[a], t1 = (bmax[a] - o[a]) / d[a];
        lo = mx(lo, mn(t0, t1));
        hi = mn(hi, mx(t0, t1));
    }
    return mx(lo, 0.0f) <= mn(hi, t_max);
}
/* A slot of a four-wide node that is not in use holds the inverted box lo = 2^60, hi = -2^60 (device/wide.h): the filter must
 * refuse it for every ray of the guarded range — direction components normal within 2^+-40, origin zero or within
 * [2^-60, 2^40], any extent — the extremes of the exponents included.  Prints: cases, passes (must be 0). */
static float guarded(int lo_exp, int hi_exp, int allow_zero) {
    const uint32_t k = u32() % 8u;
    if (allow_zero && k == 0u) return (u32() & 1u) ? 0.0f : -0.0f;
    const int e = k == 1u ? lo_exp : k == 2u ? hi_exp : lo_exp + (int)(u32() % (uint32_t)(hi_exp - lo_exp + 1));
    float m = k == 3u ? 1.0f : 1.0f + uni();
    if (e == hi_exp) m = 1.0f;
    return ((u32() & 1u) ? -1.0f : 1.0f) * ldexpf(m, e);
}
static int unused_slots(long n) {
    const float plane = 1152921504606846976.0f; /* PBRS_WIDE_UNUSED_PLANE */
    long passes = 0;
    for (long i = 0; i < n; ++i) {
        float o[3], d[3], r[3], nr[3], fr[3];
        for (int a = 0; a < 3; ++a) {
            o[a] = guarded(-60, 40, 1);
            d[a] = guarded(-40, 40, 0);
            r[a] = (float)(1.0 / (double)d[a]);
            nr[a] = d[a] > 0.0f ? plane : -plane;
            fr[a] = d[a] > 0.0f ? -plane : plane;
        }
        const uint32_t k = u32() % 4u;
        const float t_max = k == 0u ? INFINITY : k == 1u ? 0.0f : k == 2u ? 3.4028234663852886e38f : ldexpf(1.0f + uni(), (int)(u32() % 200u) - 100);
        passes += pn_slab_filter(nr[0], nr[1], nr[2], fr[0], fr[1], fr[2], o[0], o[1], o[2], r[0], r[1], r[2], t_max) ? 1 : 0;
    }
    printf("%ld %ld\n", n, passes);
    return passes != 0;
}
int main(int argc, char** argv) {
    long n = argc > 1 ? atol(argv[1]) : 1000000;
    if (argc > 2) return unused_slots(n);
    long passes = 0, fpasses = 0, bad = 0;
    for (long i = 0; i < n; ++i) {
        float bmin[3], bmax[3], o[3], d[3], t_max;
        const int kind = (int)(u32() % 6u);
        const float scale = ldexpf(1.0f, (int)(u32() % 30u) - 15);
        for (int a = 0; a < 3; ++a) {
            float c = range(-1.0f, 1.0f) * scale, e = uni() * scale * 0.5f;
            if (kind == 1 && a == (int)(u32() % 3u)) e = 0.0f; /* a flat box */
            bmin[a] = c - e;
            bmax[a] = c + e;
            if (bmin[a] > bmax[a]) { float t = bmin[a]; bmin[a] = bmax[a]; bmax[a] = t; }
            o[a] = range(-2.0f, 2.0f) * scale;
            d[a] = range(-1.0f, 1.0f);
            if (fabsf(d[a]) < 1e-6f) d[a] = 1e-6f;
        }
        if (kind == 2) { /* a ray through a corner or along an edge: origin moved so that two axes meet the box at the same time */
            float t = range(0.1f, 3.0f) * scale;
            for (int a = 0; a < 3; ++a) o[a] = ((u32() & 1u) ? bmin[a] : bmax[a]) - t * d[a];
        }
        if (kind == 3) { /* origin on a face, or inside the box */
            for (int a = 0; a < 3; ++a) o[a] = (u32() & 1u) ? bmin[a] : range(bmin[a], bmax[a]);
        }
        t_max = (u32() & 3u) ? INFINITY : range(0.0f, 4.0f) * scale;
        if (kind == 4 || kind == 5) { /* the extent equal to a plane distance: the reference's lo <= t_max on the boundary */
            int a = (int)(u32() % 3u);
            float t0 = (bmin[a] - o[a]) / d[a], t1 = (bmax[a] - o[a]) / d[a];
            t_max = kind == 4 ? mn(t0, t1) : mx(t0, t1);
            if (u32() & 1u) t_max = nextafterf(t_max, (u32() & 1u) ? INFINITY : -INFINITY);
        }
        float r[3], nr[3], fr[3];
        for (int a = 0; a < 3; ++a) {
            r[a] = (float)(1.0 / (double)d[a]);
            nr[a] = d[a] > 0.0f ? bmin[a] : bmax[a];
            fr[a] = d[a] > 0.0f ? bmax[a] : bmin[a];
        }
        int e = exact(bmin, bmax, o, d, t_max);
        int f = pn_slab_filter(nr[0], nr[1], nr[2], fr[0], fr[1], fr[2], o[0], o[1], o[2], r[0], r[1], r[2], t_max);
        passes += e;
        fpasses += f;
        if (e && !f) {
            if (bad < 5) printf("violation: kind %d box (%a %a %a)-(%a %a %a) o (%a %a %a) d (%a %a %a) t_max %a\n", kind, bmin[0], bmin[1], bmin[2], bmax[0], bmax[1], bmax[2], o[0], o[1], o[2], d[0], d[1], d[2], t_max);
            ++bad;
        }
    }
    printf("%ld %ld %ld %ld\n", n, passes, fpasses, bad);
    return bad != 0;
}
