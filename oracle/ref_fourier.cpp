// oracle/ref_fourier.cpp — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the Fourier BSDF: geometry/src/fourier.rs:99-485 (FourierTable, fourier_sum, sample_fourier,
// FourierBSDF::{eval, sample, prob}) and the spline helpers it uses, math/src/spline.rs:161-335 (find_interval,
// catmull_rom_weights, sample_catmull_rom_2d).  Operation order, precision (f32 / f64) and iteration structure follow the
// Rust source line by line; f64 sin_cos is include/pbrs_numeric.h's (the platform libm is unpinned, as for f32).
//
// Positions taken where the reference panics (counted by ref_panic, the sample contributes black):
//  * FourierBSDF::sample draws a transmitted direction (`mu_i * mu_o > 0` -> `todo!()`, fourier.rs:423-428);
//  * sample_catmull_rom_2d returns None (`.unwrap()` at :372, wo outside the table's elevations);
//  * the asserts of find_interval / Interval::new / catmull_rom_weights fail (NaN operands).
// Not in the reference: the two Newton-bisection loops stop after FOURIER_MAX_ITERATIONS rounds (they have no bound
// upstream); reaching the bound counts as a panic too, so a test that meets it notices.
#include <cmath>

#include "ref_scene.h"

namespace ref {

static const int FOURIER_MAX_ITERATIONS = 100;
static const double PI64 = 3.14159265358979323846264338327950288;       // std::f64::consts::PI
static const double FRAC_1_PI64 = 0.318309886183790671537767526745028724;  // std::f64::consts::FRAC_1_PI

std::shared_ptr<FourierTable> FourierTable::build(const pbrs_fourier_table_spec& t) {  // :115-151 (+ from_file :167-221)
    auto ft = std::make_shared<FourierTable>();
    const size_t n = t.n_mu;
    REF_ASSERT(t.n_channels == 1 || t.n_channels == 3);
    ft->n_channels = t.n_channels;
    ft->mu.assign(t.mu, t.mu + n);
    for (size_t i = 0; i + 1 < n; ++i) REF_ASSERT(ft->mu[i] <= ft->mu[i + 1]);  // :198-200
    ft->cdf.assign(t.cdf, t.cdf + n * n);
    ft->a.assign(t.a, t.a + t.n_coeffs);
    ft->a_offset.resize(n * n);
    ft->m_lookup.resize(n * n);
    for (size_t i = 0; i < n * n; ++i) {  // :205-212
        ft->a_offset[i] = t.offset_and_length[2 * i];
        ft->m_lookup[i] = t.offset_and_length[2 * i + 1];
    }
    int32_t m_max = 0;
    for (int32_t m : ft->m_lookup) m_max = m > m_max ? m : m_max;
    ft->m_max = (size_t)m_max;
    ft->a0.resize(n * n);
    for (size_t i = 0; i < n * n; ++i) {  // :125-137
        size_t end = (size_t)ft->a_offset[i] + (size_t)ft->m_lookup[i] * ft->n_channels;
        REF_ASSERT(end <= ft->a.size());
        ft->a0[i] = ft->m_lookup[i] > 0 ? ft->a[(size_t)ft->a_offset[i]] : 0.0f;
    }
    ft->recip.resize(ft->m_max);
    for (size_t i = 0; i < ft->m_max; ++i) ft->recip[i] = pn_recip((float)i);  // :138 (recip[0] = inf, never read)
    return ft;
}
const float* FourierTable::get_ak(size_t offset_i, size_t offset_o, size_t* m) const {  // :160-165
    size_t index = offset_o * mu.size() + offset_i;
    *m = (size_t)m_lookup[index];
    return a.data() + (size_t)a_offset[index];
}

// ---- math/src/spline.rs ------------------------------------------------------------------------------------------
size_t find_interval(size_t size, const std::function<bool(size_t)>& predicate) {  // :161-185
    size_t first = 0, len = size;
    while (len > 0) {
        size_t half = len >> 1, middle = first + half;
        if (predicate(middle)) {
            first = middle + 1;
            len -= half + 1;
        } else {
            len = half;
        }
    }
    size_t left = (first > 1 ? first : 1) - 1;
    left = left < size - 2 ? left : size - 2;
    if (left > 0) REF_ASSERT(predicate(left));
    if (left < size - 2) REF_ASSERT(!predicate(left + 1));
    return left;
}
bool catmull_rom_weights(const std::vector<float>& nodes, float x, long* offset, float w[4]) {  // :203-247
    REF_ASSERT(nodes.size() >= 3);
    if (x < nodes[0] || x > nodes.back()) return false;
    if (x != x) {  // a NaN passes the test above and trips `assert!(x.inside((x0, x1)))` (:214): a panic site, answered with None
        ref_panic();
        return false;
    }
    size_t i0 = find_interval(nodes.size(), [&](size_t i) { return nodes[i] <= x; });
    size_t i1 = i0 + 1;
    long il = (long)i0 - 1;
    size_t ir = i1 + 1;
    float x0 = nodes[i0], x1 = nodes[i1];
    REF_ASSERT(x0 <= x && x <= x1);
    float t = (x - x0) / (x1 - x0);
    if (!pn_isfinite(t)) {  // x1 == x0 (a repeated node): NaN weights upstream, then an index of -1; the scene builders refuse such tables
        ref_panic();
        return false;
    }
    float t2 = t * t, t3 = t * t * t;
    w[0] = 0.0f;
    w[1] = 2.0f * t3 - 3.0f * t2 + 1.0f;
    w[2] = -2.0f * t3 + 3.0f * t2;
    w[3] = 0.0f;
    if (il >= 0) {
        float w0 = (t3 - 2.0f * t2 + t) * (x1 - x0) / (x1 - nodes[(size_t)il]);
        w[0] = -w0;
        w[2] += w0;
    } else {
        float w0 = t3 - 2.0f * t2 + t;
        w[0] = 0.0f;
        w[1] -= w0;
        w[2] += w0;
    }
    if (ir < nodes.size()) {
        float w3 = (t3 - t2) * (x1 - x0) / (nodes[ir] - x0);
        w[1] -= w3;
        w[3] = w3;
    } else {
        float w3 = t3 - t2;
        w[1] -= w3;
        w[2] += w3;
        w[3] = 0.0f;
    }
    *offset = il;
    return true;
}
static float polynomial(float x, const float* c, int n) {  // math/src/float.rs:106-110
    float d = 0.0f;
    for (int i = n - 1; i >= 0; --i) d = d * x + c[i];
    return d;
}
bool sample_catmull_rom_2d(const std::vector<float>& nodes_v, const std::vector<float>& nodes_h, const std::vector<float>& values,
                           const std::vector<float>& cdf, float alpha, float u, float* fval, float* x, float* pdf) {  // :249-318
    long offset;
    float weights[4];
    if (!catmull_rom_weights(nodes_v, alpha, &offset, weights)) return false;
    const size_t nh = nodes_h.size();
    auto interpolate = [&](const std::vector<float>& array2d, size_t col) {
        float sum = 0.0f;
        for (long i = 0; i < 4; ++i)
            sum += (weights[i] == 0.0f || (size_t)(offset + i) >= nodes_v.size()) ? 0.0f : array2d[(size_t)(offset + i) * nh + col] * weights[i];
        return sum;
    };
    float maximum = interpolate(cdf, nh - 1);
    u = u * maximum;
    size_t index = find_interval(nh, [&](size_t i) { return interpolate(cdf, i) <= u; });
    float f0 = interpolate(values, index), f1 = interpolate(values, index + 1);
    float x0 = nodes_h[index], x1 = nodes_h[index + 1];
    float width = x1 - x0;
    u = (u - interpolate(cdf, index)) / width;
    float d0 = index > 0 ? width * (f1 - interpolate(values, index - 1)) / (x1 - nodes_h[index - 1]) : f1 - f0;
    float d1 = index + 2 < nh ? width * (interpolate(values, index + 2) - f0) / (nodes_h[index + 2] - x0) : f1 - f0;
    float diff = f0 - f1;
    float t = diff == 0.0f ? u / f0 : (f0 - pn_sqrt(pn_max(f0 * f0 + 2.0f * u * -diff, 0.0f))) / diff;
    float lo = 0.0f, hi = 1.0f;  // Interval::new(0.0, 1.0)
    float fhat = 0.0f;
    int it = 0;
    for (;; ++it) {
        if (it == FOURIER_MAX_ITERATIONS) {
            ref_panic();
            break;
        }
        if (!(t >= lo && t <= hi)) t = (lo + hi) * 0.5f;  // filter_or(contains, midpoint)
        const float ci[5] = {0.0f, f0, 0.5f * d0, 1.0f / 3.0f * (-2.0f * d0 - d1) + f1 - f0, 0.25f * (d0 + d1) + 0.5f * (f0 - f1)};
        const float cf[4] = {f0, d0, -2.0f * d0 - d1 + 3.0f * (f1 - f0), d0 + d1 + 2.0f * (f0 - f1)};
        float integral_hat = polynomial(t, ci, 5);
        fhat = polynomial(t, cf, 4);
        if (pn_abs(integral_hat - u) < 1e-6f || hi - lo < 1e-6f) break;
        float a = integral_hat - u < 0.0f ? t : lo, b = integral_hat - u < 0.0f ? hi : t;
        if (a != a || b != b) {  // Interval::new asserts neither end is NaN (float.rs:163-164)
            ref_panic();
            return false;
        }
        lo = a < b ? a : b;  // min_max (float.rs:197-203)
        hi = a < b ? b : a;
        t -= (integral_hat - u) / fhat;
    }
    *fval = fhat;
    *x = x0 + width * t;
    *pdf = fhat / maximum;
    return true;
}

// ---- geometry/src/fourier.rs ---------------------------------------------------------------------------------------
float fourier_sum(const float* a, size_t n, float cos_phi) {  // :224-237
    double prev = (double)cos_phi, cur = 1.0, sum = 0.0;
    for (size_t k = 0; k < n; ++k) {
        double next = 2.0 * (double)cos_phi * cur - prev;
        sum += (double)a[k] * cur;
        prev = cur;
        cur = next;
    }
    return (float)sum;
}
void sample_fourier(const float* ak, size_t n, const float* recip, float u, float* f_out, float* phi_out, float* pdf_out) {  // :245-297
    const bool flip = u >= 0.5f;
    u = flip ? 1.0f - 2.0f * (u - 0.5f) : u * 2.0f;
    double left = 0.0, right = PI64, phi = 0.5 * PI64, sampled_f = 0.0;
    for (int it = 0;; ++it) {
        if (it == FOURIER_MAX_ITERATIONS) {
            ref_panic();
            break;
        }
        double sin_phi, cos_phi;
        pn_sincos_f64(phi, &sin_phi, &cos_phi);
        double prev_cos = cos_phi, cur_cos = 1.0, prev_sin = -sin_phi, cur_sin = 0.0;
        double f_integral = (double)ak[0] * phi, f = (double)ak[0];
        for (size_t k = 1; k < n; ++k) {
            double next_sin = 2.0 * cos_phi * cur_sin - prev_sin;
            double next_cos = 2.0 * cos_phi * cur_cos - prev_cos;
            prev_cos = cur_cos, cur_cos = next_cos, prev_sin = cur_sin, cur_sin = next_sin;
            f_integral += (double)(ak[k] * recip[k]) * next_sin;
            f += (double)ak[k] * next_cos;
        }
        f_integral = f_integral - (double)(u * ak[0]) * PI64;
        if (f_integral > 0.0) right = phi;
        else left = phi;
        sampled_f = f;
        if (std::fabs(f_integral) < 1e-6 || right - left < 1e-6) break;
        phi -= f_integral / f;
        if (!(left < phi && phi < right)) phi = 0.5 * (left + right);
    }
    if (flip) phi = 2.0 * PI64 - phi;
    *pdf_out = (float)(sampled_f * FRAC_1_PI64 * 0.5) / ak[0];
    *f_out = (float)sampled_f;
    *phi_out = (float)phi;
}

static float cos_dphi(Omega a, Omega b) {  // bxdf.rs:96-107
    float res = (a.x * b.x + a.y * b.y) / pn_sqrt((a.x * a.x + a.y * a.y) * (b.x * b.x + b.y * b.y));
    return pn_isfinite(res) ? res : 0.0f;
}
// The weighted sum of the coefficient series around (mu_i, mu_o), into a_k[channel * m_max + k]; outer loop over the
// mu_o neighbours, inner over the mu_i ones (eval :331-345, sample :396-408); `channels`: how many are accumulated.
static size_t gather_ak(const FourierTable& T, long offset_i, const float* wi4, long offset_o, const float* wo4, size_t channels,
                        std::vector<float>& a_k) {
    a_k.assign(T.m_max * channels, 0.0f);
    size_t m_max = 0;
    for (long b = 0; b < 4; ++b)
        for (long a = 0; a < 4; ++a) {
            float weight = wi4[a] * wo4[b];
            // a knot outside the table (edge intervals) has weight exactly 0; the index test does not rely on it
            if (weight != 0.0f && (size_t)(offset_i + a) < T.mu.size() && (size_t)(offset_o + b) < T.mu.size()) {
                size_t m;
                const float* ap = T.get_ak((size_t)(offset_i + a), (size_t)(offset_o + b), &m);
                m_max = m > m_max ? m : m_max;
                for (size_t c = 0; c < channels; ++c)
                    for (size_t k = 0; k < m; ++k) a_k[c * T.m_max + k] += weight * ap[c * m + k];
            }
        }
    return m_max;
}

Color fourier_eval(const FourierTable& T, Omega wo, Omega wi) {  // :300-360
    float mu_i = -wi.z, mu_o = wo.z;
    float cos_phi = pn_clamp(cos_dphi(wo, -wi), -1.0f, 1.0f);
    REF_ASSERT(cos_phi >= -1.0f && cos_phi <= 1.0f);
    long offset_i, offset_o;
    float weights_i[4], weights_o[4];
    if (!catmull_rom_weights(T.mu, mu_i, &offset_i, weights_i) || !catmull_rom_weights(T.mu, mu_o, &offset_o, weights_o)) return black();
    REF_ASSERT(pn_abs(weights_i[0] + weights_i[1] + weights_i[2] + weights_i[3] - 1.0f) < 1e-3f);
    REF_ASSERT(pn_abs(weights_o[0] + weights_o[1] + weights_o[2] + weights_o[3] - 1.0f) < 1e-3f);
    std::vector<float> a_k;
    size_t m_max = gather_ak(T, offset_i, weights_i, offset_o, weights_o, T.n_channels, a_k);
    float y = pn_max(fourier_sum(a_k.data(), m_max, cos_phi), 0.0f);
    float scale = pn_abs(mu_i) == 0.0f ? 0.0f : 1.0f / pn_abs(mu_i);
    if (T.n_channels == 1) return gray(y * scale);
    float r = fourier_sum(a_k.data() + T.m_max, m_max, cos_phi);
    float b = fourier_sum(a_k.data() + 2 * T.m_max, m_max, cos_phi);
    float g = 1.39829f * y - 0.100913f * b - 0.297375f * r;
    Color c = Color{r, g, b} * scale;
    return Color{pn_clamp(c.r, 0.0f, 1.0f), pn_clamp(c.g, 0.0f, 1.0f), pn_clamp(c.b, 0.0f, 1.0f)};
}

void fourier_sample(const FourierTable& T, Omega wo, float u, float v, Color* f, Omega* wi_out, Prob* pr) {  // :362-440
    *f = black();
    *wi_out = Omega{0, 0, 1};
    *pr = Prob::Density(0.0f);
    float mu_o = wo.z, f_mu, mu_i, pdf_mu;
    if (!sample_catmull_rom_2d(T.mu, T.mu, T.a0, T.cdf, mu_o, v, &f_mu, &mu_i, &pdf_mu)) {
        ref_panic();  // `.unwrap()` on None, :372
        return;
    }
    long offset_i, offset_o;
    float weights_i[4], weights_o[4];
    if (!catmull_rom_weights(T.mu, mu_i, &offset_i, weights_i) || !catmull_rom_weights(T.mu, mu_o, &offset_o, weights_o)) return;
    std::vector<float> a_k;
    size_t m_max = gather_ak(T, offset_i, weights_i, offset_o, weights_o, T.n_channels, a_k);
    float y, phi, pdf_phi;
    if (m_max == 0) {
        y = 0.0f;
        phi = u * 2.0f * (float)PI64;
        pdf_phi = (float)FRAC_1_PI64;
    } else {
        sample_fourier(a_k.data(), m_max, T.recip.data(), u, &y, &phi, &pdf_phi);
    }
    float pdf = pn_max(pdf_phi * pdf_mu, 0.0f);
    float sin2_theta_i = pn_max(1.0f - mu_i * mu_i, 0.0f);
    float norm = pn_sqrt(sin2_theta_i / (1.0f - pn_sq(wo.z)));
    if (pn_isinf(norm)) norm = 0.0f;
    float sin_phi, cos_phi;
    pn_sincos(phi, &sin_phi, &cos_phi);
    Omega wi = -hat(Vec3{norm * (cos_phi * wo.x - sin_phi * wo.y), norm * (sin_phi * wo.x + cos_phi * wo.y), mu_i});
    float scale = pn_abs(mu_i) == 0.0f ? 0.0f : 1.0f / pn_abs(mu_i);
    if (mu_i * mu_o > 0.0f) {
        ref_panic();  // `todo!()`, :423-428: the transmitted direction's radiance scaling is not written upstream
        return;
    }
    if (T.n_channels == 1) {
        *f = gray(y * scale);
    } else {
        float r = fourier_sum(a_k.data() + T.m_max, m_max, cos_phi);
        float b = fourier_sum(a_k.data() + 2 * T.m_max, m_max, cos_phi);
        float g = 1.39829f * y - 0.100913f * b - 0.297375f * r;
        *f = Color{r * scale, g * scale, b * scale};
    }
    *wi_out = wi;
    *pr = Prob::Density(pdf);
}

Prob fourier_prob(const FourierTable& T, Omega wo, Omega wi) {  // :442-485
    float mu_i = (-wi).z, mu_o = wo.z;
    float cos_phi = cos_dphi(wo, -wi);
    long offset_i, offset_o;
    float weights_i[4], weights_o[4];
    if (!catmull_rom_weights(T.mu, mu_i, &offset_i, weights_i) || !catmull_rom_weights(T.mu, mu_o, &offset_o, weights_o))
        return Prob::Density(0.0f);
    std::vector<float> ak(T.m_max, 0.0f);
    size_t order_max = 0;
    for (long i = 0; i < 4; ++i)  // here the mu_i neighbours are the outer loop (:458)
        for (long o = 0; o < 4; ++o) {
            float weight = weights_i[i] * weights_o[o];
            if (weight == 0.0f || (size_t)(offset_i + i) >= T.mu.size() || (size_t)(offset_o + o) >= T.mu.size()) continue;
            size_t order;
            const float* coeffs = T.get_ak((size_t)(offset_i + i), (size_t)(offset_o + o), &order);
            order_max = order > order_max ? order : order_max;
            for (size_t k = 0; k < order; ++k) ak[k] += coeffs[k] * weight;
        }
    float rho = 0.0f;
    for (long o = 0; o < 4; ++o)
        rho += (weights_o[o] == 0.0f || (size_t)(offset_o + o) >= T.mu.size()) ? 0.0f : weights_o[o] * T.cdf[(size_t)(offset_o + o) * T.mu.size() + T.mu.size() - 1] * 2.0f * (float)PI64;
    float y = pn_max(fourier_sum(ak.data(), order_max, cos_phi), 0.0f);
    return Prob::Density(rho == 0.0f ? 0.0f : y / rho);
}

}  // namespace ref
