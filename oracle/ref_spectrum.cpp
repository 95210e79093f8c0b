// oracle/ref_spectrum.cpp — TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's load-time colour conversions:
// radiometry/src/spectrum.rs:3-70 (blackbody, blackbody_normalized, temperature_to_color, sampled_spectrum_to_color) and
// math/src/spline.rs:11-158 (CubicSpline, cubic_spline_zero_hess, tridiagonal).  Structure follows the Rust source — vectors in,
// vectors out, iterator sums as left folds — so that it can be read against it line by line; f64 exp_m1 / powi are
// include/pbrs_numeric.h's (the platform libm is unpinned, as everywhere).  Pinned by the reference's own vectors in
// oracle/selftest.cpp: test_temperature_to_color (spectrum.rs:471-496), tridiagonal_test, cubic_spline_solve_test and
// cubic_spline_eval_test (spline.rs:314-360).  Panic sites (`c[0]` of an empty system, the length assert of `tridiagonal`,
// `partial_cmp().unwrap()` on a NaN wavelength) count through ref_panic and return black.
#include <algorithm>

#include "../include/pbrs_cie_tables.h"
#include "ref_scene.h"

namespace ref {

static Color color_from_xyz(float x, float y, float z) {  // radiometry/src/color.rs:30-36
    return Color{3.240479f * x - 1.537150f * y - 0.498535f * z, -0.969256f * x + 1.875991f * y + 0.041556f * z,
                 0.055648f * x - 0.204043f * y + 1.057311f * z};
}
static float sum_f32(const std::vector<float>& v) {  // Iterator::sum::<f32>()
    float s = 0.0f;
    for (float x : v) s += x;
    return s;
}
static float dot_sum(const std::vector<float>& s, const float* curve) {  // radiances.iter().zip(CIE_*.iter()).map(|(s, x)| s * x).sum()
    float acc = 0.0f;
    for (size_t i = 0; i < s.size(); ++i) acc += s[i] * curve[i];
    return acc;
}

std::vector<float> blackbody(float kelvin, const std::vector<float>& lambdas_nm) {  // spectrum.rs:3-25
    std::vector<float> out;
    if (kelvin < 0.0f) {
        out.assign(lambdas_nm.size(), 0.0f);
        return out;
    }
    const double LIGHT_SPEED = 299792458.0, PLANCK = 6.62606957e-34, BOLTZMANN = 1.3806488e-23;
    for (float nm : lambdas_nm) {
        double lambda_m = (double)(nm * 1e-9f);
        double numerator = 2.0 * PLANCK * pn_powi_f64(LIGHT_SPEED, 2);
        double denominator = pn_powi_f64(lambda_m, 5) * pn_expm1_f64((PLANCK * LIGHT_SPEED) / (lambda_m * BOLTZMANN * (double)kelvin));
        out.push_back((float)(numerator / denominator));
    }
    return out;
}
std::vector<float> blackbody_normalized(float kelvin, const std::vector<float>& lambdas_nm) {  // :27-36
    const float WIEN_DISPLACEMENT = 2.8977721e-3f;
    std::vector<float> exitant_radiance = blackbody(kelvin, lambdas_nm);
    float lambda_max = WIEN_DISPLACEMENT / kelvin * 1e9f;
    float max_radiance = blackbody(kelvin, std::vector<float>{lambda_max})[0];
    for (float& r : exitant_radiance) r = r / max_radiance;
    return exitant_radiance;
}
Color temperature_to_color(float kelvin) {  // :38-55
    std::vector<float> lambdas;
    for (int i = 0; i < PBRS_CIE_SAMPLES; ++i) lambdas.push_back((float)(PBRS_CIE_LAMBDA_MIN + i));
    std::vector<float> radiances = blackbody_normalized(kelvin, lambdas);
    float xyz[3] = {dot_sum(radiances, pbrs_cie_x), dot_sum(radiances, pbrs_cie_y), dot_sum(radiances, pbrs_cie_z)};
    float scale = 1.0f / sum_f32(std::vector<float>(pbrs_cie_y, pbrs_cie_y + PBRS_CIE_SAMPLES));
    return color_from_xyz(xyz[0] * scale, xyz[1] * scale, xyz[2] * scale);
}

bool tridiagonal(const std::vector<float>& a, const std::vector<float>& b, const std::vector<float>& c, const std::vector<float>& rhs,
                 std::vector<float>* x_out) {  // spline.rs:117-141; false = the reference panics
    if (a.empty()) return false;  // `c[0]`
    std::vector<float> betas;
    betas.push_back(c[0] / b[0]);
    for (size_t i = 1; i + 1 < a.size(); ++i) betas.push_back(c[i] / (b[i] - betas[i - 1] * a[i]));
    if (betas.size() != b.size() - 1) return false;  // assert_eq!(betas.len(), b.len() - 1): a system of one equation
    std::vector<float> ys;
    ys.push_back(rhs[0] / b[0]);
    for (size_t i = 1; i < a.size(); ++i) ys.push_back((rhs[i] - a[i] * ys[i - 1]) / (b[i] - a[i] * betas[i - 1]));
    std::vector<float> xs = ys;
    for (size_t i = a.size() - 1; i-- > 0;) xs[i] = ys[i] - betas[i] * xs[i + 1];
    *x_out = xs;
    return true;
}
bool cubic_spline_zero_hess(const std::vector<std::pair<float, float>>& xs_and_ys, std::vector<float>* m_out) {  // :83-104
    std::vector<std::pair<float, float>> dx_and_dydx;
    for (size_t i = 0; i + 1 < xs_and_ys.size(); ++i) {
        float x0 = xs_and_ys[i].first, y0 = xs_and_ys[i].second, x1 = xs_and_ys[i + 1].first, y1 = xs_and_ys[i + 1].second;
        dx_and_dydx.push_back({x1 - x0, (y1 - y0) / (x1 - x0)});
    }
    std::vector<float> mus_1n_1, lambdas_1n_1, ds_1n_1;
    for (size_t i = 0; i + 1 < dx_and_dydx.size(); ++i) {
        float dx0 = dx_and_dydx[i].first, dy0 = dx_and_dydx[i].second, dx1 = dx_and_dydx[i + 1].first, dy1 = dx_and_dydx[i + 1].second;
        float x_1 = xs_and_ys[i].first, x1 = xs_and_ys[i + 2].first;
        mus_1n_1.push_back(dx0 / (dx0 + dx1));
        lambdas_1n_1.push_back(1.0f - dx0 / (dx0 + dx1));
        ds_1n_1.push_back(6.0f * (dy1 - dy0) / (x1 - x_1));
    }
    std::vector<float> diag(ds_1n_1.size(), 2.0f);
    return tridiagonal(mus_1n_1, diag, lambdas_1n_1, ds_1n_1, m_out);
}
bool CubicSpline::from_samples(const std::vector<std::pair<float, float>>& xs_and_ys, CubicSpline* out) {  // :21-36
    std::vector<float> inner;
    if (!cubic_spline_zero_hess(xs_and_ys, &inner)) return false;
    out->m.assign(1, 0.0f);
    out->m.insert(out->m.end(), inner.begin(), inner.end());
    out->m.push_back(0.0f);
    out->xs.clear();
    out->ys.clear();
    for (const auto& p : xs_and_ys) {
        out->xs.push_back(p.first);
        out->ys.push_back(p.second);
    }
    return out->m.size() == out->xs.size();
}
float CubicSpline::evaluate(float at) const {  // :40-59
    size_t i1 = (size_t)(std::partition_point(xs.begin(), xs.end(), [&](float x) { return x < at; }) - xs.begin());
    if (i1 == 0) return ys[0];
    if (i1 >= ys.size()) return ys.back();
    float x0 = xs[i1 - 1], x1 = xs[i1];
    float y0 = ys[i1 - 1], y1 = ys[i1];
    float m0 = m[i1 - 1], m1 = m[i1];
    REF_ASSERT(at >= x0 && at <= x1);
    float h = x1 - x0;
    float frac_1_6h = 1.0f / (6.0f * h);
    return 0.0f + m0 * pn_powi(x1 - at, 3) * frac_1_6h + m1 * pn_powi(at - x0, 3) * frac_1_6h + (y0 - m0 * h * h / 6.0f) * (x1 - at) / h +
           (y1 - m1 * h * h / 6.0f) * (at - x0) / h;
}
Color sampled_spectrum_to_color(std::vector<std::pair<float, float>> lambdas_and_values) {  // spectrum.rs:57-70
    for (const auto& p : lambdas_and_values)
        if (p.first != p.first) {  // partial_cmp(..).unwrap()
            ref_panic();
            return black();
        }
    std::stable_sort(lambdas_and_values.begin(), lambdas_and_values.end(),
                     [](const std::pair<float, float>& a, const std::pair<float, float>& b) { return a.first < b.first; });
    CubicSpline spline;
    if (!CubicSpline::from_samples(lambdas_and_values, &spline)) {
        ref_panic();
        return black();
    }
    std::vector<float> radiances;
    for (int i = 0; i < PBRS_CIE_SAMPLES; ++i) radiances.push_back(spline.evaluate((float)(PBRS_CIE_LAMBDA_MIN + i)));
    Color c = color_from_xyz(dot_sum(radiances, pbrs_cie_x), dot_sum(radiances, pbrs_cie_y), dot_sum(radiances, pbrs_cie_z));
    return c * (1.0f / sum_f32(std::vector<float>(pbrs_cie_y, pbrs_cie_y + PBRS_CIE_SAMPLES)));
}

}  // namespace ref
