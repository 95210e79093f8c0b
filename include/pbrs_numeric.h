/* pbrs_numeric.h — the f32 numeric contract shared by every side of the pbrs_gpu boundary.
 *
 * The reference (plumer/pbrs, Rust) calls the platform libm through
 * f32::{sin_cos,tan,atan,atan2,acos,exp,ln,hypot,powi,signum,max,min,clamp,fract}
 * (math/src/float.rs:267-278, geometry/src/microfacet.rs:17,49,132-150,
 *  geometry/src/bxdf.rs:70-91,196, shape/src/simple.rs:247-249, light/src/sample_shape.rs:187-225).
 * libm's last-ulp behaviour is platform dependent, i.e. unpinned by the reference.  To make
 * "radiance matches at matched seeds" attainable between the host flattener, the HIP kernels
 * and the CPU oracle, all three evaluate transcendentals with the routines below, which use only
 * IEEE-754 f32 + - * / sqrt, integer ops and comparisons.  Every translation unit that includes
 * this header MUST be compiled with -ffp-contract=off (Rust never contracts a*b+c into an fma;
 * hipcc and gcc do by default) and with correctly rounded f32 divide/sqrt (hipcc's default).
 *
 * Algorithms: the classic single-precision Cephes kernels (Moshier, public domain): Cody-Waite
 * 3-part pi/4 reduction + degree-7/8 minimax polynomials for sin/cos, 4-term odd polynomial for
 * atan on [0, tan(pi/8)], asin via 5-term polynomial, exp via 2-part ln2 reduction + degree-5
 * polynomial, log via frexp + degree-8 polynomial.  tests/test_numeric.py pins them to glibc's
 * libm within a stated ulp bound; tests/test_gpu_numeric.py checks gfx950 == x86 bit for bit.
 *
 * Plain C99 / C++ / HIP.  No dependencies.
 */
#ifndef PBRS_NUMERIC_H
#define PBRS_NUMERIC_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PN_FN __host__ __device__ static inline
#else
#define PN_FN static inline
#endif

#define PN_PI 3.14159265358979323846f        /* std::f32::consts::PI        */
#define PN_FRAC_1_PI 0.318309886183790671538f /* std::f32::consts::FRAC_1_PI */
#define PN_FRAC_PI_2 1.57079632679489661923f  /* std::f32::consts::FRAC_PI_2 */
#define PN_FRAC_PI_4 0.785398163397448309616f
#define PN_EPSILON 1.1920928955078125e-7f      /* f32::EPSILON */

PN_FN uint32_t pn_bits(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return u;
}
PN_FN float pn_from_bits(uint32_t u) {
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
PN_FN float pn_inf(void) { return pn_from_bits(0x7f800000u); }
PN_FN float pn_nan(void) { return pn_from_bits(0x7fc00000u); }
PN_FN int pn_isnan(float x) { return x != x; }
PN_FN int pn_isinf(float x) { return (pn_bits(x) & 0x7fffffffu) == 0x7f800000u; }
PN_FN int pn_isfinite(float x) { return (pn_bits(x) & 0x7f800000u) != 0x7f800000u; }
/* f32::is_sign_negative: the sign BIT (true for -0.0 and negative NaN). */
PN_FN int pn_sign_negative(float x) { return (int)(pn_bits(x) >> 31); }
PN_FN float pn_abs(float x) { return pn_from_bits(pn_bits(x) & 0x7fffffffu); }
/* f32::signum: +1 for +0.0 and positives, -1 for -0.0 and negatives, NaN for NaN. */
PN_FN float pn_signum(float x) {
    if (x != x) return x;
    return pn_sign_negative(x) ? -1.0f : 1.0f;
}
/* f32::max / f32::min: IEEE maxNum/minNum — a NaN operand is ignored. */
PN_FN float pn_max(float a, float b) { /* select chain (no branches): same values as the early-return form */
    float m = a > b ? a : b;
    m = (a != a) ? b : m;
    return (b != b) ? a : m;
}
PN_FN float pn_min(float a, float b) {
    float m = a < b ? a : b;
    m = (a != a) ? b : m;
    return (b != b) ? a : m;
}
/* f32::clamp(lo, hi) (NaN stays NaN). */
PN_FN float pn_clamp(float x, float lo, float hi) {
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}
PN_FN float pn_sqrt(float x) { return __builtin_sqrtf(x); }
PN_FN float pn_recip(float x) { return 1.0f / x; }
/* math/src/float.rs:116-122 */
PN_FN float pn_weak_recip(float x) { return x == 0.0f ? 0.0f : 1.0f / x; }
/* trunc for |x| < 2^31; beyond 2^23 every f32 is already integral. */
PN_FN float pn_trunc(float x) {
    if (!(pn_abs(x) < 8388608.0f)) return x;
    float t = (float)(int32_t)x;
    /* keep the sign of zero like f32::trunc does */
    return pn_from_bits(pn_bits(t) | (pn_bits(x) & 0x80000000u));
}
PN_FN float pn_floor(float x) {
    float t = pn_trunc(x);
    return t > x ? t - 1.0f : t;
}
/* f32::fract = self - self.trunc() */
PN_FN float pn_fract(float x) { return x - pn_trunc(x); }
/* `x as i32` of Rust: toward zero, saturating, NaN -> 0 */
PN_FN int32_t pn_f32_to_i32(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (int32_t)(-2147483647 - 1);
    return (int32_t)x;
}

/* compiler-rt __powisf2 / LLVM's constant-exponent expansion: square-and-multiply, LSB first. */
PN_FN float pn_powi(float a, int b) {
    int recip = b < 0;
    float r = 1.0f;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1.0f / r : r;
}
PN_FN float pn_sq(float a) { return a * a; } /* powi(2) */
/* f32::mul_add: ONE rounding (the only fused operation on the path; everything else is compiled -ffp-contract=off) */
PN_FN float pn_mul_add(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

/* 2^n as f32 for n in [-126, 127] */
PN_FN float pn_exp2i(int n) { return pn_from_bits((uint32_t)(n + 127) << 23); }
/* x * 2^n without intermediate overflow for the n range exp() produces. */
PN_FN float pn_ldexp(float x, int n) {
    if (n > 254) n = 254;
    if (n < -252) n = -252;
    int h = n / 2;
    return (x * pn_exp2i(h)) * pn_exp2i(n - h);
}

/* ---- sin / cos ------------------------------------------------------------------------- */
PN_FN void pn_reduce_pio4_(float ax, int* jout, float* rout) {
    const float FOPI = 1.27323954473516f;
    const float DP1 = 0.78515625f;
    const float DP2 = 2.4187564849853515625e-4f;
    const float DP3 = 3.77489497744594108e-8f;
    int j = (int)(FOPI * ax);
    float y = (float)j;
    if (j & 1) {
        j += 1;
        y += 1.0f;
    }
    *jout = j & 7;
    *rout = ((ax - y * DP1) - y * DP2) - y * DP3;
}
PN_FN float pn_sinpoly_(float x, float z) {
    float y = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * x;
    return y + x;
}
PN_FN float pn_cospoly_(float z) {
    float y = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z;
    y = y - 0.5f * z;
    return y + 1.0f;
}
/* Valid for |x| < 8192 (callers pass angles within a few turns). Returns (sin x, cos x). */
PN_FN void pn_sincos(float x, float* s, float* c) {
    if (!pn_isfinite(x)) {
        *s = pn_nan();
        *c = pn_nan();
        return;
    }
    int sneg = x < 0.0f;
    float ax = pn_abs(x);
    int j;
    float r;
    pn_reduce_pio4_(ax, &j, &r);
    int cneg = 0;
    if (j > 3) {
        j -= 4;
        sneg = !sneg;
        cneg = !cneg;
    }
    if (j > 1) cneg = !cneg;
    float z = r * r;
    float ps = pn_sinpoly_(r, z);
    float pc = pn_cospoly_(z);
    float sv, cv;
    if (j == 1 || j == 2) {
        sv = pc;
        cv = ps;
    } else {
        sv = ps;
        cv = pc;
    }
    *s = sneg ? -sv : sv;
    *c = cneg ? -cv : cv;
}
PN_FN float pn_sin(float x) {
    float s, c;
    pn_sincos(x, &s, &c);
    return s;
}
PN_FN float pn_cos(float x) {
    float s, c;
    pn_sincos(x, &s, &c);
    return c;
}
PN_FN float pn_tan(float x) {
    float s, c;
    pn_sincos(x, &s, &c);
    return s / c;
}

/* ---- atan / atan2 / asin / acos ------------------------------------------------------------ */
PN_FN float pn_atan(float xx) {
    if (xx != xx) return xx;
    int neg = pn_sign_negative(xx);
    float x = pn_abs(xx);
    float y;
    if (x > 2.414213562373095f) { /* tan 3pi/8 */
        y = PN_FRAC_PI_2;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) { /* tan pi/8 */
        y = PN_FRAC_PI_4;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
    }
    float z = x * x;
    y += (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
    return neg ? -y : y;
}
/* f32::atan2(self = y, other = x) */
PN_FN float pn_atan2(float y, float x) {
    if (x != x || y != y) return pn_nan();
    int code = 0;
    if (x < 0.0f) code = 2;
    if (y < 0.0f) code |= 1;
    if (x == 0.0f) {
        if (code & 1) return -PN_FRAC_PI_2;
        if (y == 0.0f) return 0.0f;
        return PN_FRAC_PI_2;
    }
    if (y == 0.0f) return (code & 2) ? PN_PI : 0.0f;
    float w = 0.0f;
    if (code == 2) w = PN_PI;
    if (code == 3) w = -PN_PI;
    return w + pn_atan(y / x);
}
PN_FN float pn_asin(float xx) {
    int neg = pn_sign_negative(xx);
    float a = pn_abs(xx);
    if (!(a <= 1.0f)) return pn_nan();
    float z, x;
    int flag = 0;
    if (a < 1.0e-4f) {
        z = a;
        return neg ? -z : z;
    }
    if (a > 0.5f) {
        z = 0.5f * (1.0f - a);
        x = pn_sqrt(z);
        flag = 1;
    } else {
        x = a;
        z = x * x;
    }
    z = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * x + x;
    if (flag) {
        z = z + z;
        z = PN_FRAC_PI_2 - z;
    }
    return neg ? -z : z;
}
PN_FN float pn_acos(float x) {
    if (!(x >= -1.0f && x <= 1.0f)) return pn_nan();
    if (x > 0.5f) return 2.0f * pn_asin(pn_sqrt(0.5f * (1.0f - x)));
    if (x < -0.5f) return PN_PI - 2.0f * pn_asin(pn_sqrt(0.5f * (1.0f + x)));
    return PN_FRAC_PI_2 - pn_asin(x);
}

/* ---- exp / ln -------------------------------------------------------------------------------- */
PN_FN float pn_exp(float x) {
    if (x != x) return x;
    if (x > 88.72283905206835f) return pn_inf();
    if (x < -103.278929903431851103f) return 0.0f;
    float z = pn_floor(1.44269504088896341f * x + 0.5f);
    x = x - z * 0.693359375f;
    x = x - z * -2.12194440e-4f;
    int n = (int)z;
    z = x * x;
    z = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x + 1.6666665459e-1f) * x + 5.0000001201e-1f) * z + x + 1.0f;
    return pn_ldexp(z, n);
}
/* f32::ln */
PN_FN float pn_ln(float x) {
    if (x != x) return x;
    if (x < 0.0f) return pn_nan();
    if (x == 0.0f) return -pn_inf();
    if (pn_isinf(x)) return x;
    /* frexp: x = m * 2^e, m in [0.5, 1) */
    uint32_t u = pn_bits(x);
    int e = 0;
    if ((u & 0x7f800000u) == 0) { /* subnormal: scale up by 2^24 */
        x = x * 16777216.0f;
        u = pn_bits(x);
        e = -24;
    }
    e += (int)((u >> 23) & 0xffu) - 126;
    float m = pn_from_bits((u & 0x807fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float y = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m + 1.4249322787e-1f) * m - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m + 3.3333331174e-1f) * m * z;
    float fe = (float)e;
    if (e) y += -2.12194440e-4f * fe;
    y += -0.5f * z;
    z = m + y;
    if (e) z += 0.693359375f * fe;
    return z;
}
/* f32::hypot for the magnitudes on this path (|x|,|y| <= ~1e18: no overflow handling needed). */
PN_FN float pn_hypot(float x, float y) { return pn_sqrt(x * x + y * y); }
/* f32::to_radians: the reference multiplies by the f32 constant PI/180. */
/* f64::sin_cos for the Newton-bisection of geometry/src/fourier.rs:245-297 (`phi.sin_cos()` on an f64 angle in [0, 2 pi]).
 * The double-precision Cephes kernels: 3-part Cody-Waite pi/4 reduction, degree-6 polynomials in z = x^2.  Only IEEE
 * f64 + - * and an integer conversion; valid for |x| < 2^30 (the callers stay inside [0, 2 pi]). */
PN_FN void pn_sincos_f64(double x, double* s, double* c) {
    int sign_s = 0, sign_c = 0;
    double ax = x;
    if (ax < 0.0) {
        ax = -ax;
        sign_s = 1;
    }
    if (!(ax < 1073741824.0)) { /* out of the reduction's range (and NaN): no caller gets here */
        *s = 0.0;
        *c = 1.0;
        return;
    }
    int64_t j = (int64_t)(ax * 1.27323954473516268615 /* 4 / pi */);
    if (j & 1) j += 1; /* map zeros to the origin */
    double y = (double)j;
    j &= 7;
    if (j > 3) {
        sign_s ^= 1;
        sign_c ^= 1;
        j -= 4;
    }
    if (j > 1) sign_c ^= 1;
    double z = ((ax - y * 7.85398125648498535156e-1) - y * 3.77489470793079817668e-8) - y * 2.69515142907905952645e-15;
    double zz = z * z;
    double ps = 1.58962301576546568060e-10;
    ps = ps * zz + -2.50507477628578072866e-8;
    ps = ps * zz + 2.75573136213857245213e-6;
    ps = ps * zz + -1.98412698295895385996e-4;
    ps = ps * zz + 8.33333333332211858878e-3;
    ps = ps * zz + -1.66666666666666307295e-1;
    double sin_z = z + z * zz * ps;
    double pc = -1.13585365213876817300e-11;
    pc = pc * zz + 2.08757008419747316778e-9;
    pc = pc * zz + -2.75573141792967388112e-7;
    pc = pc * zz + 2.48015872888517045348e-5;
    pc = pc * zz + -1.38888888888730564116e-3;
    pc = pc * zz + 4.16666666666665929218e-2;
    double cos_z = 1.0 - 0.5 * zz + zz * zz * pc;
    double sv = (j == 1 || j == 2) ? cos_z : sin_z;
    double cv = (j == 1 || j == 2) ? sin_z : cos_z;
    *s = sign_s ? -sv : sv;
    *c = sign_c ? -cv : cv;
}

PN_FN float pn_to_radians(float deg) { return deg * (PN_PI / 180.0f); }

/* ---- f64 pieces of radiometry/src/spectrum.rs:3-25 (Planck's law in f64: `powi(2)`, `powi(5)`, `exp_m1`) --------------------
 * f64::powi = compiler-rt __powidf2: square-and-multiply, LSB first (as pn_powi). */
PN_FN double pn_powi_f64(double a, int b) {
    int recip = b < 0;
    double r = 1.0;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1.0 / r : r;
}
/* f64::exp / f64::exp_m1: the double-precision Cephes kernels (2-part ln2 reduction, Pade-type rational in x^2); the
 * platform libm they stand for is unpinned by the reference, as for f32.  Only IEEE f64 + - * / and integer ops. */
PN_FN double pn_exp_f64(double x) {
    if (x != x) return x;
    if (x > 709.782712893384) return (double)pn_inf();
    if (x < -745.13321910194122) return 0.0;
    double fl = 1.4426950408889634073599 * x + 0.5;
    int64_t n = (int64_t)fl;
    if ((double)n > fl) n -= 1; /* floor */
    const double px0 = (double)n;
    x -= px0 * 6.93145751953125e-1;
    x -= px0 * 1.42860682030941723212e-6;
    const double xx = x * x;
    double p = 1.26177193074810590878e-4;
    p = p * xx + 3.02994407707441961300e-2;
    p = p * xx + 9.99999999999999999910e-1;
    p = x * p;
    double q = 3.00198505138664455042e-6;
    q = q * xx + 2.52448340349684104192e-3;
    q = q * xx + 2.27265548208155028766e-1;
    q = q * xx + 2.00000000000000000009e0;
    double r = 1.0 + 2.0 * (p / (q - p));
    /* ldexp in two steps so that results near the ends of the range pass through normal scale factors */
    int64_t n1 = n / 2, n2 = n - n1;
    union { uint64_t u; double d; } s1, s2;
    s1.u = (uint64_t)(n1 + 1023) << 52;
    s2.u = (uint64_t)(n2 + 1023) << 52;
    return r * s1.d * s2.d;
}
PN_FN double pn_expm1_f64(double x) {
    if (x != x) return x;
    if (x < -0.5 || x > 0.5) return pn_exp_f64(x) - 1.0;
    const double xx = x * x;
    double p = 1.2617719307481059087798e-4;
    p = p * xx + 3.0299440770744196129956e-2;
    p = p * xx + 9.9999999999999999991025e-1;
    double r = x * p;
    double q = 3.0019850513866445504159e-6;
    q = q * xx + 2.5244834034968410419224e-3;
    q = q * xx + 2.2726554820815502876593e-1;
    q = q * xx + 2.0000000000000000000897e0;
    r = r / (q - r);
    return r + r;
}

/* ---- conservative filter for the box test of geometry/src/bvh.rs:84-99 ------------------------------------------------------
 * The reference's test passes iff max(0, max_a min(q0a, q1a)) <= min(t_max, min_a max(q0a, q1a)) with the six correctly
 * rounded quotients q = RN((b - o) / d).  For a ray inside the guarded range of the division-free test (device/traverse.h:
 * every direction component normal and within 2^+-40, box and origin coordinates zero or within [2^-60, 2^40]) no quotient is a
 * NaN, an infinity or a subnormal, min(q0a, q1a) is the quotient of the plane the ray meets first on that axis (each rounding
 * step is monotone), and the filter below passes WHENEVER the reference's test passes:
 *   - it takes the same numerators n = RN(b - o) and multiplies by r = RN32(R), R within 1 ulp64 of 1/d: q' = RN(n * r) =
 *     (n / d)(1 + b)(1 + c)(1 + e) with |b| <= 2^-52, |c|, |e| <= 2^-24, while q = (n / d)(1 + a), |a| <= 2^-24; hence
 *     |q' - q| <= eps |q'| with eps < 2^-22;
 *   - x -> x - eps|x| and x -> x + eps|x| are monotone, so they commute with the min / max over the axes:
 *     lo = max_a q_near >= lo' - eps|lo'| and hi = min_a q_far <= hi' + eps|hi'|;
 *   - the two widened bounds are computed with one rounding each (fma) and PN_SLAB_EPS = 2^-21, which leaves more than
 *     2^-22 of margin after that rounding.
 * So lo <= hi, lo <= t_max, 0 <= hi imply the same of the widened bounds.  The filter may also pass boxes the reference's test
 * rejects (none among 7 10^4 hits of random boxes; a third more on rays aimed through box corners or extents placed on a
 * plane distance): callers use it only where passing too often is harmless — inner nodes,
 * whose rejection only prunes (a box inside a rejected box is rejected too), and leaves that get the reference's own test, with
 * the extent of that moment, before anything depends on them.  tests/test_slab_filter.py checks the implication on 10^7
 * adversarial cases (flat boxes, grazing rays, extents equal to the entry distance). */
#define PN_SLAB_EPS 4.76837158203125e-7f /* 2^-21 */
/* near / far: the box planes the ray meets first / last on each axis (min / max plane by the sign of the direction);
 * rx, ry, rz: f32 roundings of the f64 reciprocals of the direction. */
PN_FN int pn_slab_filter(float near_x, float near_y, float near_z, float far_x, float far_y, float far_z, float ox, float oy, float oz,
                         float rx, float ry, float rz, float t_max) {
    const float tnx = (near_x - ox) * rx, tny = (near_y - oy) * ry, tnz = (near_z - oz) * rz;
    const float tfx = (far_x - ox) * rx, tfy = (far_y - oy) * ry, tfz = (far_z - oz) * rz;
    const float lo = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), tnz);
    const float hi = __builtin_fminf(__builtin_fminf(tfx, tfy), tfz);
    const float lo_w = __builtin_fmaf(-__builtin_fabsf(lo), PN_SLAB_EPS, lo);
    const float hi_w = __builtin_fmaf(__builtin_fabsf(hi), PN_SLAB_EPS, hi);
    return __builtin_fmaxf(lo_w, 0.0f) <= __builtin_fminf(hi_w, t_max);
}

/* ---- RNG contract --------------------------------------------------
 * The reference draws from an unseedable thread_rng (ChaCha12); the north-star asks for fixed
 * per-pixel seeds, so the stream is a new contract: one PCG32 (XSH-RR 64/32) generator per camera
 * sample, keyed by (seed, pixel_index = row*W + col, sample_index), consumed in the reference's
 * draw order.  f32 mapping is rand 0.8's `Standard`: (u32 >> 8) * 2^-24 in [0,1).
 */
PN_FN uint64_t pn_mix64(uint64_t z) { /* splitmix64 finaliser */
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
PN_FN uint64_t pn_rng_init(uint64_t seed, uint32_t pixel_index, uint32_t sample_index) {
    uint64_t key = ((uint64_t)pixel_index << 32) | (uint64_t)sample_index;
    return pn_mix64(seed + 0x9e3779b97f4a7c15ULL + pn_mix64(key));
}
PN_FN uint32_t pn_rng_u32(uint64_t* state) {
    uint64_t old = *state;
    *state = old * 6364136223846793005ULL + 1442695040888963407ULL;
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
}
PN_FN float pn_rng_f32(uint64_t* state) { return (float)(pn_rng_u32(state) >> 8) * 5.9604644775390625e-8f; }

#endif /* PBRS_NUMERIC_H */
