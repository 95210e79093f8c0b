// device/fourier.h — the Fourier BSDF: geometry/src/fourier.rs:224-485 (fourier_sum, sample_fourier, FourierBSDF::{eval,
// sample, prob}) and math/src/spline.rs:161-318 (find_interval, catmull_rom_weights, sample_catmull_rom_2d) for one lane.
//
// The reference gathers the interpolated coefficient series a_k of a direction pair into a heap vector (m_max entries per
// channel: up to thousands) and then sums it.  A lane has no such storage, so a_k is recomputed where it is consumed: entry k
// is the sum, in the reference's neighbour order, of weight * series[k] over the (up to) 4 x 4 neighbouring table entries
// whose series reach k — the same f32 products added in the same order, hence the same bits.  The sums over k run in f64
// (Chebyshev recurrence for cos(k phi)) exactly as upstream; f64 +, -, *, / are IEEE on gfx950 and nothing contracts.
//
// Where the reference panics — `todo!()` for a sampled transmitted direction (:423-428), `.unwrap()` of a None from
// sample_catmull_rom_2d (:372) — the lobe returns black with density 0 and the path ends (the oracle counts these).
// The two Newton-bisection loops carry the oracle's bound (FOURIER_MAX_ITERATIONS), which no well-formed table reaches.
#pragma once
#include "dmath.h"

#define PBRS_FOURIER_MAX_ITERATIONS 100

struct FourierView {  // where the tables of the scene are (DevScene): nullptr members in kernels compiled without the lobe
    const pbrs_fourier_table* tables;
    const float* F;
    const uint32_t* W;
    float* ak;        // per-lane LDS column (entry k at ak[k * 256]) for the luminance series of a sampled direction pair, or nullptr
    uint32_t ak_cap;  // ... and its rows
    bool only;  // every lobe the kernel meets is a Fourier lobe (k_shade's PBRS_SHADE_FOURIER_ONLY variants, launched over the Fourier
                // materials' class): a compile-time constant there — the other lobe kinds' code is gone, a lobe's kind is not read
};

// math/src/spline.rs:161-185 with `predicate(i) = nodes[i] <= x`; its asserts are panic sites upstream (NaN operands)
PD uint32_t find_interval_le(const float* nodes, uint32_t size, float x) {
    uint32_t first = 0, len = size;
    while (len > 0) {
        uint32_t half = len >> 1, middle = first + half;
        if (nodes[middle] <= x) {
            first = middle + 1;
            len -= half + 1;
        } else {
            len = half;
        }
    }
    uint32_t left = (first > 1u ? first : 1u) - 1u;
    return left < size - 2u ? left : size - 2u;
}
// math/src/spline.rs:203-247.  false = None (x outside the nodes).  offset = index of the first of the four knots (-1 .. n - 3)
PD bool catmull_rom_weights(const float* nodes, uint32_t n, float x, int& offset, float& w0o, float& w1o, float& w2o, float& w3o) {
    // written so that a NaN x leaves too: upstream a NaN passes `x < nodes[0] || x > nodes[n - 1]` and panics at
    // `assert!(x.inside((x0, x1)))`; here, as at every panic site of the lobe, the answer is None (black, density 0).  A NaN
    // direction is reachable: hat() of a zero vector towards a light sample on the surface.
    if (!(nodes[0] <= x && x <= nodes[n - 1u])) return false;
    const uint32_t i0 = find_interval_le(nodes, n, x), i1 = i0 + 1u;
    const int il = (int)i0 - 1;
    const uint32_t ir = i1 + 1u;
    const float x0 = nodes[i0], x1 = nodes[i1];
    const float t = (x - x0) / (x1 - x0);
    if (!pn_isfinite(t)) return false;  // x1 == x0: pbrs_upload_scene refuses such tables; kept so that no weight can be NaN
    const float t2 = t * t, t3 = t * t * t;
    float w_0 = 0.0f, w_1 = 2.0f * t3 - 3.0f * t2 + 1.0f, w_2 = -2.0f * t3 + 3.0f * t2, w_3 = 0.0f;
    if (il >= 0) {
        const float w0 = (t3 - 2.0f * t2 + t) * (x1 - x0) / (x1 - nodes[il]);
        w_0 = -w0;
        w_2 += w0;
    } else {
        const float w0 = t3 - 2.0f * t2 + t;
        w_0 = 0.0f;
        w_1 -= w0;
        w_2 += w0;
    }
    if (ir < n) {
        const float w3 = (t3 - t2) * (x1 - x0) / (nodes[ir] - x0);
        w_1 -= w3;
        w_3 = w3;
    } else {
        const float w3 = t3 - t2;
        w_1 -= w3;
        w_2 += w3;
        w_3 = 0.0f;
    }
    offset = il;
    w0o = w_0, w1o = w_1, w2o = w_2, w3o = w_3;
    return true;
}

// The 4 x 4 neighbourhood of a direction pair in one table
struct FourierNbrs {
    const float* a;
    const uint32_t* a_offset;
    const uint32_t* m_lookup;
    uint32_t n_mu;
    int offset_i, offset_o;
    float wi[4], wo[4];
    PD float weight(int a_, int b_) const { return wi[a_] * wo[b_]; }
    // a neighbour outside the table (knot -1 or n_mu at an edge interval) has weight exactly 0 whenever the weights are
    // finite; the index test makes that independent of the arithmetic: no lane reads outside the table's arrays
    PD bool inside(int a_, int b_) const { return (uint32_t)(offset_o + b_) < n_mu && (uint32_t)(offset_i + a_) < n_mu; }
    // largest series length among the neighbours with a non-zero weight (the reference's running `m_max`)
    PD uint32_t order() const {
        uint32_t m_max = 0;
#pragma unroll
        for (int b_ = 0; b_ < 4; ++b_)
#pragma unroll
            for (int a_ = 0; a_ < 4; ++a_)
                if (weight(a_, b_) != 0.0f && inside(a_, b_)) {
                    const uint32_t m = m_lookup[(uint32_t)(offset_o + b_) * n_mu + (uint32_t)(offset_i + a_)];
                    m_max = m > m_max ? m : m_max;
                }
        return m_max;
    }
    // a_k[channel * m_max + k] of eval (:331-345) and sample (:396-408): neighbours of mu_o outside, of mu_i inside
    PD float coef_oi(uint32_t channel, uint32_t k) const {
        float acc = 0.0f;
#pragma unroll
        for (int b_ = 0; b_ < 4; ++b_)
#pragma unroll
            for (int a_ = 0; a_ < 4; ++a_) {
                const float w = weight(a_, b_);
                if (w != 0.0f && inside(a_, b_)) {
                    const uint32_t index = (uint32_t)(offset_o + b_) * n_mu + (uint32_t)(offset_i + a_);
                    const uint32_t m = m_lookup[index];
                    if (k < m) acc += w * a[a_offset[index] + channel * m + k];
                }
            }
        return acc;
    }
    // the same sums for several channels of one term in one pass over the neighbours: per channel the additions of coef_oi in
    // their order (the series lengths and offsets are fetched once instead of once per channel)
    template <bool Y>
    PD void coef_oi_yrb(uint32_t k, float& cy, float& cr, float& cb) const {
        float ay = 0.0f, ar = 0.0f, ab = 0.0f;
#pragma unroll
        for (int b_ = 0; b_ < 4; ++b_)
#pragma unroll
            for (int a_ = 0; a_ < 4; ++a_) {
                const float w = weight(a_, b_);
                if (w != 0.0f && inside(a_, b_)) {
                    const uint32_t index = (uint32_t)(offset_o + b_) * n_mu + (uint32_t)(offset_i + a_);
                    const uint32_t m = m_lookup[index];
                    if (k < m) {
                        const float* t = a + (a_offset[index] + k);
                        if (Y) ay += w * t[0];
                        ar += w * t[m];
                        ab += w * t[2u * m];
                    }
                }
            }
        cy = ay, cr = ar, cb = ab;
    }
    // ak[k] of prob (:456-468): neighbours of mu_i outside, of mu_o inside; luminance only
    PD float coef_io(uint32_t k) const {
        float acc = 0.0f;
#pragma unroll
        for (int a_ = 0; a_ < 4; ++a_)
#pragma unroll
            for (int b_ = 0; b_ < 4; ++b_) {
                const float w = weight(a_, b_);
                if (w == 0.0f || !inside(a_, b_)) continue;
                const uint32_t index = (uint32_t)(offset_o + b_) * n_mu + (uint32_t)(offset_i + a_);
                const uint32_t m = m_lookup[index];
                if (k < m) acc += a[a_offset[index] + k] * w;
            }
        return acc;
    }
};
PD bool fourier_nbrs(const FourierView& V, const pbrs_fourier_table& T, float mu_i, float mu_o, FourierNbrs& N) {
    const float* mu = V.F + T.mu;
    N.a = V.F + T.a;
    N.a_offset = V.W + T.a_offset;
    N.m_lookup = V.W + T.m_lookup;
    N.n_mu = T.n_mu;
    return catmull_rom_weights(mu, T.n_mu, mu_i, N.offset_i, N.wi[0], N.wi[1], N.wi[2], N.wi[3]) &&
           catmull_rom_weights(mu, T.n_mu, mu_o, N.offset_o, N.wo[0], N.wo[1], N.wo[2], N.wo[3]);
}

// fourier_sum (:224-237) over the series `coef(k)`, k < n: sum of a_k cos(k phi) with the cosines by Chebyshev's recurrence, in f64
template <typename Coef>
PD float fourier_sum(Coef coef, uint32_t n, float cos_phi) {
    double prev = (double)cos_phi, cur = 1.0, sum = 0.0;
    for (uint32_t k = 0; k < n; ++k) {
        const double next = 2.0 * (double)cos_phi * cur - prev;
        sum += (double)coef(k) * cur;
        prev = cur;
        cur = next;
    }
    return (float)sum;
}
// fourier_sum of the red and blue series (and the luminance series where Y) of one direction pair in one loop: the cosines
// of the three calls are the same numbers, each sum takes its terms in the order of its own call
template <bool Y>
PD void fourier_sum_yrb(const FourierNbrs& N, uint32_t n, float cos_phi, float& y, float& r, float& b) {
    double prev = (double)cos_phi, cur = 1.0, sy = 0.0, sr = 0.0, sb = 0.0;
    for (uint32_t k = 0; k < n; ++k) {
        const double next = 2.0 * (double)cos_phi * cur - prev;
        float cy, cr, cb;
        N.template coef_oi_yrb<Y>(k, cy, cr, cb);
        if (Y) sy += (double)cy * cur;
        sr += (double)cr * cur;
        sb += (double)cb * cur;
        prev = cur;
        cur = next;
    }
    y = (float)sy, r = (float)sr, b = (float)sb;
}
PD float cos_dphi(f3 a, f3 b) {  // bxdf.rs:96-107
    const float res = (a.x * b.x + a.y * b.y) / pn_sqrt((a.x * a.x + a.y * a.y) * (b.x * b.x + b.y * b.y));
    return pn_isfinite(res) ? res : 0.0f;
}

PD f3 fourier_eval(const FourierView& V, const pbrs_fourier_table& T, f3 wo, f3 wi) {  // :300-360
    const float mu_i = -wi.z, mu_o = wo.z;
    const float cos_phi = pn_clamp(cos_dphi(wo, -wi), -1.0f, 1.0f);
    FourierNbrs N;
    if (!fourier_nbrs(V, T, mu_i, mu_o, N)) return gray(0.0f);
    const uint32_t m_max = N.order();
    const float scale = pn_abs(mu_i) == 0.0f ? 0.0f : 1.0f / pn_abs(mu_i);
    if (T.n_channels == 1u) return gray(pn_max(fourier_sum([&](uint32_t k) { return N.coef_oi(0u, k); }, m_max, cos_phi), 0.0f) * scale);
    float y, r, b;
    fourier_sum_yrb<true>(N, m_max, cos_phi, y, r, b);
    y = pn_max(y, 0.0f);
    const float g = 1.39829f * y - 0.100913f * b - 0.297375f * r;
    const f3 c = mk3(r, g, b) * scale;
    return mk3(pn_clamp(c.x, 0.0f, 1.0f), pn_clamp(c.y, 0.0f, 1.0f), pn_clamp(c.z, 0.0f, 1.0f));
}

PD ProbD fourier_prob(const FourierView& V, const pbrs_fourier_table& T, f3 wo, f3 wi) {  // :442-485
    const float mu_i = (-wi).z, mu_o = wo.z;
    const float cos_phi = cos_dphi(wo, -wi);
    FourierNbrs N;
    if (!fourier_nbrs(V, T, mu_i, mu_o, N)) return density(0.0f);
    const uint32_t order_max = N.order();
    const float* cdf = V.F + T.cdf;
    float rho = 0.0f;
#pragma unroll
    for (int o = 0; o < 4; ++o)
        rho += (N.wo[o] == 0.0f || (uint32_t)(N.offset_o + o) >= T.n_mu) ? 0.0f : N.wo[o] * cdf[(uint32_t)(N.offset_o + o) * T.n_mu + T.n_mu - 1u] * 2.0f * PN_PI;
    const float y = pn_max(fourier_sum([&](uint32_t k) { return N.coef_io(k); }, order_max, cos_phi), 0.0f);
    return density(rho == 0.0f ? 0.0f : y / rho);
}

// sample_fourier (:245-297) over the luminance series coef(k), k < n (n >= 1 terms)
template <typename Coef>
PD void sample_fourier_t(Coef coef, uint32_t n, const float* recip, float u, float& f_out, float& phi_out, float& pdf_out) {
    const bool flip = u >= 0.5f;
    u = flip ? 1.0f - 2.0f * (u - 0.5f) : u * 2.0f;
    const double PI64 = 3.14159265358979323846264338327950288, FRAC_1_PI64 = 0.318309886183790671537767526745028724;
    const float ak0 = coef(0u);
    double left = 0.0, right = PI64, phi = 0.5 * PI64, sampled_f = 0.0;
    for (int it = 0; it < PBRS_FOURIER_MAX_ITERATIONS; ++it) {
        double sin_phi, cos_phi;
        pn_sincos_f64(phi, &sin_phi, &cos_phi);
        double prev_cos = cos_phi, cur_cos = 1.0, prev_sin = -sin_phi, cur_sin = 0.0;
        double f_integral = (double)ak0 * phi, f = (double)ak0;
        for (uint32_t k = 1; k < n; ++k) {
            const double next_sin = 2.0 * cos_phi * cur_sin - prev_sin;
            const double next_cos = 2.0 * cos_phi * cur_cos - prev_cos;
            prev_cos = cur_cos, cur_cos = next_cos, prev_sin = cur_sin, cur_sin = next_sin;
            const float akk = coef(k);
            f_integral += (double)(akk * recip[k]) * next_sin;
            f += (double)akk * next_cos;
        }
        f_integral = f_integral - (double)(u * ak0) * PI64;
        if (f_integral > 0.0) right = phi;
        else left = phi;
        sampled_f = f;
        if (__builtin_fabs(f_integral) < 1e-6 || right - left < 1e-6) break;
        phi -= f_integral / f;
        if (!(left < phi && phi < right)) phi = 0.5 * (left + right);
    }
    if (flip) phi = 2.0 * PI64 - phi;
    pdf_out = (float)(sampled_f * FRAC_1_PI64 * 0.5) / ak0;
    f_out = (float)sampled_f;
    phi_out = (float)phi;
}
// The Newton-bisection loop reads the whole series once per iteration: where the kernel has a per-lane LDS column for it (the
// Fourier-only variants of k_shade) and the series fits, a_k is interpolated ONCE — the same f32 sums, stored and read back
// unchanged — instead of once per iteration (16 neighbours x three loads per term).
PD void sample_fourier(const FourierView& V, const FourierNbrs& N, uint32_t n, const float* recip, float u, float& f_out, float& phi_out, float& pdf_out) {
    if (V.ak && n <= V.ak_cap) {
        for (uint32_t k = 0; k < n; ++k) V.ak[k * 256u] = N.coef_oi(0u, k);
        const float* ak = V.ak;
        sample_fourier_t([ak](uint32_t k) { return ak[k * 256u]; }, n, recip, u, f_out, phi_out, pdf_out);
    } else {
        sample_fourier_t([&N](uint32_t k) { return N.coef_oi(0u, k); }, n, recip, u, f_out, phi_out, pdf_out);
    }
}

PD float polynomial4(float x, float c0, float c1, float c2, float c3) { return ((((0.0f * x + c3) * x + c2) * x + c1) * x + c0); }
PD float polynomial5(float x, float c0, float c1, float c2, float c3, float c4) {  // math/src/float.rs:106-110
    return (((((0.0f * x + c4) * x + c3) * x + c2) * x + c1) * x + c0);
}
// sample_catmull_rom_2d(mu, mu, a0, cdf, alpha, u) (spline.rs:249-318).  false = None, or a NaN reached Interval::new
PD bool sample_catmull_rom_2d(const float* nodes, uint32_t n, const float* values, const float* cdf, float alpha, float u, float& fval,
                              float& x, float& pdf) {
    int offset;
    float w[4];
    if (!catmull_rom_weights(nodes, n, alpha, offset, w[0], w[1], w[2], w[3])) return false;
    auto interpolate = [&](const float* array2d, uint32_t col) {
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) sum += (w[i] == 0.0f || (uint32_t)(offset + i) >= n) ? 0.0f : array2d[(uint32_t)(offset + i) * n + col] * w[i];
        return sum;
    };
    const float maximum = interpolate(cdf, n - 1u);
    u = u * maximum;
    uint32_t index;
    {  // find_interval(n, |i| interpolate(cdf, i) <= u)
        uint32_t first = 0, len = n;
        while (len > 0) {
            const uint32_t half = len >> 1, middle = first + half;
            if (interpolate(cdf, middle) <= u) {
                first = middle + 1;
                len -= half + 1;
            } else {
                len = half;
            }
        }
        const uint32_t left = (first > 1u ? first : 1u) - 1u;
        index = left < n - 2u ? left : n - 2u;
    }
    const float f0 = interpolate(values, index), f1 = interpolate(values, index + 1u);
    const float x0 = nodes[index], x1 = nodes[index + 1u];
    const float width = x1 - x0;
    u = (u - interpolate(cdf, index)) / width;
    const float d0 = index > 0u ? width * (f1 - interpolate(values, index - 1u)) / (x1 - nodes[index - 1u]) : f1 - f0;
    const float d1 = index + 2u < n ? width * (interpolate(values, index + 2u) - f0) / (nodes[index + 2u] - x0) : f1 - f0;
    const float diff = f0 - f1;
    float t = diff == 0.0f ? u / f0 : (f0 - pn_sqrt(pn_max(f0 * f0 + 2.0f * u * -diff, 0.0f))) / diff;
    float lo = 0.0f, hi = 1.0f, fhat = 0.0f;
    for (int it = 0; it < PBRS_FOURIER_MAX_ITERATIONS; ++it) {
        if (!(t >= lo && t <= hi)) t = (lo + hi) * 0.5f;
        const float integral_hat =
            polynomial5(t, 0.0f, f0, 0.5f * d0, 1.0f / 3.0f * (-2.0f * d0 - d1) + f1 - f0, 0.25f * (d0 + d1) + 0.5f * (f0 - f1));
        fhat = polynomial4(t, f0, d0, -2.0f * d0 - d1 + 3.0f * (f1 - f0), d0 + d1 + 2.0f * (f0 - f1));
        if (pn_abs(integral_hat - u) < 1e-6f || hi - lo < 1e-6f) break;
        const float a = integral_hat - u < 0.0f ? t : lo, b = integral_hat - u < 0.0f ? hi : t;
        if (a != a || b != b) return false;
        lo = a < b ? a : b;
        hi = a < b ? b : a;
        t -= (integral_hat - u) / fhat;
    }
    fval = fhat;
    x = x0 + width * t;
    pdf = fhat / maximum;
    return true;
}

PD void fourier_sample(const FourierView& V, const pbrs_fourier_table& T, f3 wo, float u, float v, f3& f, f3& wi_out, ProbD& pr) {  // :362-440
    f = gray(0.0f);
    wi_out = mk3(0.0f, 0.0f, 1.0f);
    pr = density(0.0f);
    const float mu_o = wo.z;
    float f_mu, mu_i, pdf_mu;
    if (!sample_catmull_rom_2d(V.F + T.mu, T.n_mu, V.F + T.a0, V.F + T.cdf, mu_o, v, f_mu, mu_i, pdf_mu)) return;
    FourierNbrs N;
    if (!fourier_nbrs(V, T, mu_i, mu_o, N)) return;
    const uint32_t m_max = N.order();
    float y, phi, pdf_phi;
    if (m_max == 0u) {
        y = 0.0f;
        phi = u * 2.0f * PN_PI;
        pdf_phi = PN_FRAC_1_PI;
    } else {
        sample_fourier(V, N, m_max, V.F + T.recip, u, y, phi, pdf_phi);
    }
    const float pdf = pn_max(pdf_phi * pdf_mu, 0.0f);
    const float sin2_theta_i = pn_max(1.0f - mu_i * mu_i, 0.0f);
    float norm = pn_sqrt(sin2_theta_i / (1.0f - pn_sq(wo.z)));
    if (pn_isinf(norm)) norm = 0.0f;
    float sin_phi, cos_phi;
    pn_sincos(phi, &sin_phi, &cos_phi);
    const f3 wi = -hat(mk3(norm * (cos_phi * wo.x - sin_phi * wo.y), norm * (sin_phi * wo.x + cos_phi * wo.y), mu_i));
    const float scale = pn_abs(mu_i) == 0.0f ? 0.0f : 1.0f / pn_abs(mu_i);
    if (mu_i * mu_o > 0.0f) return;  // `todo!()` upstream
    if (T.n_channels == 1u) {
        f = gray(y * scale);
    } else {
        float y_unused, r, b;
        fourier_sum_yrb<false>(N, m_max, cos_phi, y_unused, r, b);
        const float g = 1.39829f * y - 0.100913f * b - 0.297375f * r;
        f = mk3(r * scale, g * scale, b * scale);
    }
    wi_out = wi;
    pr = density(pdf);
}
