// pbrs_amd/csrc/host/spectrum.h — spectra to RGB at scene-load time: radiometry/src/spectrum.rs:3-70 (`blackbody`,
// `blackbody_normalized`, `temperature_to_color`, `sampled_spectrum_to_color`) over math/src/spline.rs:11-158 (`CubicSpline`,
// `cubic_spline_zero_hess`, `tridiagonal`) and the CIE tables (:72-466, include/pbrs_cie_tables.h).  Used by the pbrt front-end for
// `"blackbody L" [T scale]` colours (scene/src/loader.rs:763) and metal `eta` / `k` from `.spd` files (:548-570, :858-879).
// f32 throughout except Planck's law, which the reference evaluates in f64; every operation in the reference's order;
// f64 `exp_m1` / `powi` from include/pbrs_numeric.h (the platform libm is unpinned upstream).  Where the reference would panic
// (fewer than four samples: `tridiagonal` indexes an empty vector or fails its length assert; a NaN wavelength: `partial_cmp().unwrap()`)
// SpectrumError is thrown.
#pragma once
#include <algorithm>
#include <stdexcept>
#include <utility>
#include <vector>

#include "../../../include/pbrs_cie_tables.h"
#include "../../../include/pbrs_numeric.h"

namespace spectrum {

struct SpectrumError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct Rgb {
    float r, g, b;
};
// Color::from_xyz, radiometry/src/color.rs:30-36
inline Rgb from_xyz(float x, float y, float z) {
    return {3.240479f * x - 1.537150f * y - 0.498535f * z, -0.969256f * x + 1.875991f * y + 0.041556f * z,
            0.055648f * x - 0.204043f * y + 1.057311f * z};
}
// `iter().sum::<f32>()`: a left fold
inline float cie_y_sum() {
    float s = 0.0f;
    for (int i = 0; i < PBRS_CIE_SAMPLES; ++i) s += pbrs_cie_y[i];
    return s;
}

// spectrum.rs:3-25: exitant radiance of a black body at `kelvin` for a wavelength in nanometres
inline float blackbody(float kelvin, float lambda_nm) {
    if (kelvin < 0.0f) return 0.0f;
    const double c = 299792458.0, h = 6.62606957e-34, kb = 1.3806488e-23;
    const double lambda_m = (double)(lambda_nm * 1e-9f);  // `(nm * 1e-9) as f64`: the product is an f32 one
    const double numerator = 2.0 * h * pn_powi_f64(c, 2);
    const double denominator = pn_powi_f64(lambda_m, 5) * pn_expm1_f64((h * c) / (lambda_m * kb * (double)kelvin));
    return (float)(numerator / denominator);
}
// :38-55 with :27-36 inlined: radiances normalised by the one at Wien's peak, integrated against the CIE curves
inline Rgb temperature_to_color(float kelvin) {
    const float lambda_max = 2.8977721e-3f / kelvin * 1e9f;
    const float max_radiance = blackbody(kelvin, lambda_max);
    float x = 0.0f, y = 0.0f, z = 0.0f;
    std::vector<float> radiance(PBRS_CIE_SAMPLES);
    for (int i = 0; i < PBRS_CIE_SAMPLES; ++i) radiance[i] = blackbody(kelvin, (float)(PBRS_CIE_LAMBDA_MIN + i)) / max_radiance;
    for (int i = 0; i < PBRS_CIE_SAMPLES; ++i) x += radiance[i] * pbrs_cie_x[i];
    for (int i = 0; i < PBRS_CIE_SAMPLES; ++i) y += radiance[i] * pbrs_cie_y[i];
    for (int i = 0; i < PBRS_CIE_SAMPLES; ++i) z += radiance[i] * pbrs_cie_z[i];
    const float scale = 1.0f / cie_y_sum();
    return from_xyz(x * scale, y * scale, z * scale);
}

// spline.rs:117-141: Thomas' algorithm as the reference writes it (a[0] and c[n - 1] are never read)
inline std::vector<float> tridiagonal(const std::vector<float>& a, const std::vector<float>& b, const std::vector<float>& c, const std::vector<float>& rhs) {
    const size_t n = a.size();
    if (n < 2) throw SpectrumError("a cubic spline needs at least four samples");  // `c[0]` of an empty vector / the length assert at n = 1
    std::vector<float> betas{c[0] / b[0]};
    for (size_t i = 1; i + 1 < n; ++i) betas.push_back(c[i] / (b[i] - betas[i - 1] * a[i]));
    std::vector<float> ys{rhs[0] / b[0]};
    for (size_t i = 1; i < n; ++i) ys.push_back((rhs[i] - a[i] * ys[i - 1]) / (b[i] - a[i] * betas[i - 1]));
    std::vector<float> xs = ys;
    for (size_t i = n - 1; i-- > 0;) xs[i] = ys[i] - betas[i] * xs[i + 1];
    return xs;
}
// spline.rs:83-104: the second derivatives M_1 .. M_{n-1} of the natural spline through the samples
inline std::vector<float> cubic_spline_zero_hess(const std::vector<std::pair<float, float>>& p) {
    std::vector<float> dx, dydx;
    for (size_t i = 0; i + 1 < p.size(); ++i) {
        dx.push_back(p[i + 1].first - p[i].first);
        dydx.push_back((p[i + 1].second - p[i].second) / (p[i + 1].first - p[i].first));
    }
    std::vector<float> mus, lambdas, ds;
    for (size_t i = 0; i + 2 < p.size(); ++i) {
        mus.push_back(dx[i] / (dx[i] + dx[i + 1]));
        lambdas.push_back(1.0f - dx[i] / (dx[i] + dx[i + 1]));
        ds.push_back(6.0f * (dydx[i + 1] - dydx[i]) / (p[i + 2].first - p[i].first));
    }
    return tridiagonal(mus, std::vector<float>(ds.size(), 2.0f), lambdas, ds);
}
struct CubicSpline {  // spline.rs:11-60
    std::vector<float> m, xs, ys;
    explicit CubicSpline(const std::vector<std::pair<float, float>>& p) {
        m.push_back(0.0f);
        for (float v : cubic_spline_zero_hess(p)) m.push_back(v);
        m.push_back(0.0f);
        for (const auto& s : p) {
            xs.push_back(s.first);
            ys.push_back(s.second);
        }
    }
    float evaluate(float at) const {
        size_t i1 = 0;  // partition_point(|&x| x < at)
        while (i1 < xs.size() && xs[i1] < at) ++i1;
        if (i1 == 0) return ys.front();
        if (i1 >= ys.size()) return ys.back();
        const float x0 = xs[i1 - 1], x1 = xs[i1], y0 = ys[i1 - 1], y1 = ys[i1], m0 = m[i1 - 1], m1 = m[i1];
        const float h = x1 - x0;
        const float frac_1_6h = 1.0f / (6.0f * h);
        return 0.0f + m0 * pn_powi(x1 - at, 3) * frac_1_6h + m1 * pn_powi(at - x0, 3) * frac_1_6h + (y0 - m0 * h * h / 6.0f) * (x1 - at) / h +
               (y1 - m1 * h * h / 6.0f) * (at - x0) / h;
    }
};
// spectrum.rs:57-70: samples sorted by wavelength (a stable sort), a natural cubic spline through them, its values at the CIE
// wavelengths integrated against the curves
inline Rgb sampled_spectrum_to_color(std::vector<std::pair<float, float>> samples) {
    for (const auto& s : samples)
        if (s.first != s.first) throw SpectrumError("a spectrum sample has a NaN wavelength");  // `partial_cmp().unwrap()`
    std::stable_sort(samples.begin(), samples.end(), [](const std::pair<float, float>& a, const std::pair<float, float>& b) { return a.first < b.first; });
    const CubicSpline spline(samples);
    float x = 0.0f, y = 0.0f, z = 0.0f;
    std::vector<float> radiance(PBRS_CIE_SAMPLES);
    for (int i = 0; i < PBRS_CIE_SAMPLES; ++i) radiance[i] = spline.evaluate((float)(PBRS_CIE_LAMBDA_MIN + i));
    for (int i = 0; i < PBRS_CIE_SAMPLES; ++i) x += radiance[i] * pbrs_cie_x[i];
    for (int i = 0; i < PBRS_CIE_SAMPLES; ++i) y += radiance[i] * pbrs_cie_y[i];
    for (int i = 0; i < PBRS_CIE_SAMPLES; ++i) z += radiance[i] * pbrs_cie_z[i];
    const Rgb c = from_xyz(x, y, z);
    const float scale = 1.0f / cie_y_sum();
    return {c.r * scale, c.g * scale, c.b * scale};
}

}  // namespace spectrum
