// oracle/ref_shading.cpp — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// Restates geometry/src/bxdf.rs, geometry/src/microfacet.rs, src/bsdf.rs, material/src/lib.rs
// (`bxdfs_at` / `emission`), light/src/lib.rs and light/src/sample_shape.rs.
#include "ref_scene.h"

namespace ref {

// ---- Omega helpers (geometry/src/bxdf.rs:42-155) -------------------------------------------------
static inline float cos_theta(Omega w) { return w.z; }                       // :52-54
static inline float cos2_theta(Omega w) { return pn_sq(w.z); }               // :55-57
static inline float sin2_theta(Omega w) { return 1.0f - cos2_theta(w); }     // :61-63
static inline float sin_theta(Omega w) { return pn_sqrt(pn_max(sin2_theta(w), 0.0f)); }  // :58-60
static inline float tan2_theta(Omega w) { return sin2_theta(w) / cos2_theta(w); }        // :64-66
static inline float cos2_phi(Omega w) {  // :76-79
    float den = w.x * w.x + w.y * w.y;
    return den == 0.0f ? 1.0f : (w.x * w.x) / den;
}
static inline float sin2_phi(Omega w) {  // :80-83
    float den = w.x * w.x + w.y * w.y;
    return den == 0.0f ? 0.0f : (w.y * w.y) / den;
}
static inline void sin_cos_phi(Omega w, float* a, float* b) {  // :85-93 (Q5: returns (x/h, y/h))
    float xy_hypot = pn_hypot(w.x, w.y);
    if (xy_hypot == 0.0f) {
        *a = 0.0f;
        *b = 1.0f;
    } else {
        *a = w.x / xy_hypot;
        *b = w.y / xy_hypot;
    }
}
static inline bool same_hemisphere(Omega w0, Omega w1) { return cos_theta(w0) * cos_theta(w1) >= 0.0f; }  // :111-113
static inline bool bisector(Omega w0, Omega w1, Omega* out) { return try_hat(w0 + w1, out); }             // :143-146
static inline Omega face_forward(Omega self, Omega normal) {                                              // :149-155
    return pn_sign_negative(dot(self, normal)) ? -self : self;
}

void concentric_sample_disk(float u, float v, float* ox, float* oy) {  // :187-200
    float x = u * 2.0f - 1.0f;
    float y = v * 2.0f - 1.0f;
    if (x == 0.0f && y == 0.0f) {
        *ox = 0.0f;
        *oy = 0.0f;
        return;
    }
    float r = pn_abs(pn_abs(x) > pn_abs(y) ? x : y);
    float hypot = pn_hypot(x, y);
    float cos_t = x / hypot, sin_t = y / hypot;
    *ox = r * cos_t;
    *oy = r * sin_t;
}
Omega cos_sample_hemisphere(float u, float v) {  // :202-206
    float x, y;
    concentric_sample_disk(u, v, &x, &y);
    float z = pn_sqrt(pn_max(1.0f - x * x - y * y, 0.0f));
    return Omega{x, y, z};
}
static inline float cos_hemisphere_pdf(Omega w) { return cos_theta(w) * PN_FRAC_1_PI; }  // :208-210

// ---- Fresnel (:284-393) ------------------------------------------------------------------------------
Fresnel fresnel_nop() {
    Fresnel f{};
    f.kind = Fresnel::Nop;
    return f;
}
Fresnel fresnel_dielectric(float eta_front, float eta_back) {
    Fresnel f{};
    f.kind = Fresnel::Dielectric;
    f.eta_front = eta_front;
    f.eta_back = eta_back;
    return f;
}
Fresnel fresnel_conductor(Color eta_real, Color eta_imag) {
    Fresnel f{};
    f.kind = Fresnel::Conductor;
    f.eta_i = gray(1.0f);
    f.eta_t = eta_real;
    f.k = eta_imag;
    return f;
}
float Fresnel::refl_coeff(float cos_theta_i) const {  // :308-342
    if (kind == Nop) return 1.0f;
    if (kind == Conductor) {
        ref_panic();
        return 0.0f;
    }
    cos_theta_i = pn_clamp(cos_theta_i, -1.0f, 1.0f);
    float eta_i, eta_t;
    if (cos_theta_i > 0.0f) {
        eta_i = eta_front;
        eta_t = eta_back;
    } else {
        eta_i = eta_back;
        eta_t = eta_front;
        cos_theta_i = -cos_theta_i;
    }
    float sin_theta_i = pn_sqrt(pn_max(1.0f - pn_sq(cos_theta_i), 0.0f));
    float sin_theta_t = eta_i / eta_t * sin_theta_i;
    if (sin_theta_t >= 1.0f) return 1.0f;
    float cos_theta_t = pn_sqrt(pn_max(1.0f - pn_sq(sin_theta_t), 0.0f));
    float r_perpendicular = (eta_i * cos_theta_i - eta_t * cos_theta_t) / (eta_i * cos_theta_i + eta_t * cos_theta_t);
    float r_parallel = (eta_t * cos_theta_i - eta_i * cos_theta_t) / (eta_t * cos_theta_i + eta_i * cos_theta_t);
    return (pn_sq(r_parallel) + pn_sq(r_perpendicular)) * 0.5f;
}
Color Fresnel::eval(float cos_theta_i) const {  // :344-392
    if (kind != Conductor) return gray(refl_coeff(cos_theta_i));
    Color eta = cw_div(eta_t, eta_i);
    Color eta2 = eta * eta;
    Color etak = cw_div(k, eta_i);
    Color etak2 = etak * etak;
    float cos2_theta_i = pn_sq(pn_clamp(cos_theta_i, -1.0f, 1.0f));
    float sin2_theta_i = 1.0f - cos2_theta_i;
    Color t0 = eta2 - etak2 - gray(sin2_theta_i);
    Color a2_plus_b2 = cw_sqrt(t0 * t0 + 4.0f * eta2 * etak2);
    Color t1 = a2_plus_b2 + gray(cos2_theta_i);
    Color a = cw_sqrt((a2_plus_b2 + t0) * 0.5f);
    Color t2 = 2.0f * a * cos_theta_i;
    Color ratio_s = cw_div(t1 - t2, t1 + t2);
    REF_ASSERT(is_finite(ratio_s));
    Color t3 = cos2_theta_i * a2_plus_b2 + gray(pn_sq(sin2_theta_i));
    Color t4 = t2 * sin2_theta_i;
    Color ratio_p = ratio_s * cw_div(t3 - t4, t3 + t4);
    REF_ASSERT(is_finite(ratio_p));
    return cw_max((ratio_s + ratio_p) * 0.5f, 0.0f);
}

// ---- MicrofacetDistrib (geometry/src/microfacet.rs) ------------------------------------------------------
float roughness_to_alpha(float roughness) {  // :16-23
    float x = pn_max(pn_ln(roughness), -8.0f);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}
float MicrofacetDistrib::d(Omega wh) const {  // :36-60
    float tan2 = tan2_theta(wh);
    float cos4_theta = pn_sq(cos2_theta(wh));
    REF_ASSERT(!pn_isnan(tan2));
    REF_ASSERT(!pn_isnan(cos4_theta));
    if (pn_isinf(tan2)) return 0.0f;
    if (kind == Beckmann) {
        float x = cos2_phi(wh) / pn_sq(alpha_x) + sin2_phi(wh) / pn_sq(alpha_y);
        return pn_exp(x * -tan2) / (PN_PI * alpha_x * alpha_y * cos4_theta);
    }
    float e = cos2_phi(wh) / pn_sq(alpha_x) + sin2_phi(wh) / pn_sq(alpha_y);
    return pn_recip(pn_sq(1.0f + e * tan2) * (PN_PI * alpha_x * alpha_y * cos4_theta));
}
float MicrofacetDistrib::lambda(Omega w) const {  // :65-88
    float abs_tan_theta = pn_abs(pn_sqrt(tan2_theta(w)));
    if (pn_isinf(abs_tan_theta)) return 0.0f;
    if (kind == Beckmann) {
        float alpha = pn_sqrt(cos2_phi(w) * pn_sq(alpha_x) + sin2_phi(w) * pn_sq(alpha_y));
        float a = pn_recip(alpha * abs_tan_theta);
        if (a >= 1.6f) return 0.0f;
        return (1.0f - 1.259f * a + 0.396f * a * a) / (3.535f * a + 2.181f * a * a);
    }
    float alpha2 = cos2_phi(w) * pn_sq(alpha_x) + sin2_phi(w) * pn_sq(alpha_y);
    float alpha2_tan2_theta = alpha2 * tan2_theta(w);
    return (-1.0f + pn_sqrt(1.0f + alpha2_tan2_theta)) * 0.5f;
}
float MicrofacetDistrib::g1(Omega w) const { return pn_recip(1.0f + lambda(w)); }                       // :99-101
float MicrofacetDistrib::g(Omega wo, Omega wi) const { return pn_recip(1.0f + lambda(wo) + lambda(wi)); }  // :106-108
float MicrofacetDistrib::pdf(Omega, Omega wh) const {  // :110-122 (cfg(not(sample_visible_area)))
    float x = d(wh);
    float y = pn_abs(cos_theta(wh));
    REF_ASSERT(!pn_isnan(x * y));
    return d(wh) * pn_abs(cos_theta(wh));
}
Omega MicrofacetDistrib::sample_wh(Omega wo, float u, float v) const {  // :124-159
    if (kind != Beckmann) {  // Q19: TrowbridgeReitz::sample_wh is todo!()
        ref_panic();
        return Omega{0, 0, 1};
    }
    float tan2, phi;
    if (alpha_x == alpha_y) {
        float log_sample = pn_ln(1.0f - u);
        REF_ASSERT(pn_isfinite(log_sample));
        tan2 = -pn_sq(alpha_x) * log_sample;
        phi = v * 2.0f * PN_PI;
    } else {
        float log_sample = pn_ln(1.0f - u);
        REF_ASSERT(pn_isfinite(log_sample));
        phi = pn_atan(alpha_y / alpha_x * pn_tan(2.0f * PN_PI * v + PN_FRAC_PI_2));
        if (v >= 0.5f) phi += PN_PI;
        float sin_phi, cos_phi;
        pn_sincos(phi, &sin_phi, &cos_phi);
        float alpha2 = pn_sq(cos_phi / alpha_x) + pn_sq(sin_phi / alpha_y);
        tan2 = -log_sample / alpha2;
    }
    float cos_t = pn_recip(pn_sqrt(1.0f + tan2));
    float sin_t = cos_t * pn_sqrt(tan2);
    Omega wh = spherical_direction(sin_t, cos_t, phi);
    return face_forward(wh, wo);
}

// ---- BxDF constructors ------------------------------------------------------------------------------
BXDF bxdf_mirror(Color albedo) {
    BXDF b{};
    b.kind = BXDF::Specular;
    b.fresnel = fresnel_nop();
    b.albedo = albedo;
    b.intrusion = BXDF::Reflection;
    return b;
}
BXDF bxdf_dielectric(Color albedo, float eta_outer, float eta_inner) {
    BXDF b{};
    b.kind = BXDF::Specular;
    b.fresnel = fresnel_dielectric(eta_outer, eta_inner);
    b.albedo = albedo;
    b.intrusion = BXDF::Hybrid;
    return b;
}
BXDF bxdf_transmit(Color albedo, float eta_outer, float eta_inner) {
    BXDF b{};
    b.kind = BXDF::Specular;
    b.fresnel = fresnel_dielectric(eta_outer, eta_inner);
    b.albedo = albedo;
    b.intrusion = BXDF::Transmission;
    return b;
}
BXDF bxdf_lambertian(Color albedo) {
    BXDF b{};
    b.kind = BXDF::DiffuseReflect;
    b.albedo = albedo;
    b.oren_nayar = false;
    return b;
}
BXDF bxdf_oren_nayar(Color albedo, float sigma_rad) {  // :528-536
    BXDF b{};
    b.kind = BXDF::DiffuseReflect;
    b.albedo = albedo;
    b.oren_nayar = true;
    float sigma_sqr = pn_sq(sigma_rad);
    b.coeff_a = 1.0f - (sigma_sqr / (2.0f * (sigma_sqr + 0.33f)));
    b.coeff_b = 0.45f * sigma_sqr / (sigma_sqr + 0.09f);
    return b;
}
BXDF bxdf_microfacet(Color albedo, MicrofacetDistrib d, Fresnel f) {
    BXDF b{};
    b.kind = BXDF::MicrofacetReflect;
    b.albedo = albedo;
    b.distrib = d;
    b.fresnel = f;
    return b;
}
BXDF bxdf_fresnel_blend(Color diffuse, Color specular, MicrofacetDistrib d) {
    BXDF b{};
    b.kind = BXDF::FresnelBlend;
    b.diffuse = diffuse;
    b.specular = specular;
    b.distrib = d;
    return b;
}

// ---- Specular (:395-506) ----------------------------------------------------------------------------
static void specular_reflect(const BXDF& s, Omega wo, Omega* wi, Color* f) {  // :427-434
    *wi = Omega{-wo.x, -wo.y, wo.z};
    Color fr_refl = s.fresnel.eval(cos_theta(*wi));
    *f = fr_refl * s.albedo * pn_weak_recip(pn_abs(cos_theta(*wi)));
}
static void specular_refract(const BXDF& s, Omega wo, float eta_front, float eta_back, Omega* wi, Color* f) {  // :436-454
    float eta_i, eta_t;
    Vec3 normal;
    if (cos_theta(wo) > 0.0f) {
        eta_i = eta_front;
        eta_t = eta_back;
        normal = Vec3{0, 0, 1};
    } else {
        eta_i = eta_back;
        eta_t = eta_front;
        normal = -Vec3{0, 0, 1};
    }
    Vec3 t;
    if (!refract(normal, wo, eta_i / eta_t, &t)) {
        *wi = Omega{0, 0, 0};
        *f = black();
        return;
    }
    *wi = t;
    float f_tr = 1.0f - s.fresnel.refl_coeff(cos_theta(t));
    *f = (f_tr / pn_abs(cos_theta(t))) * s.albedo;
}

// Schlick term of FresnelBlend (:663-665)
static Color schlick_fresnel(const BXDF& b, float cos_t) {
    return b.specular + pn_powi(1.0f - cos_t, 5) * (gray(1.0f) - b.specular);
}

Color fourier_eval(const FourierTable& T, Omega wo, Omega wi);  // ref_fourier.cpp
void fourier_sample(const FourierTable& T, Omega wo, float u, float v, Color* f, Omega* wi_out, Prob* pr);
Prob fourier_prob(const FourierTable& T, Omega wo, Omega wi);

Color BXDF::eval(Omega wo, Omega wi) const {
    if (kind == Fourier) return fourier_eval(*table, wo, wi);  // fourier.rs:300-360
    switch (kind) {
        case Specular: return black();  // :458-460
        case DiffuseReflect: {           // :540-559
            if (!oren_nayar) return albedo * PN_FRAC_1_PI;
            float sin_theta_i = sin_theta(wi);
            float sin_theta_o = sin_theta(wo);
            float sin_phi_i, cos_phi_i, sin_phi_o, cos_phi_o;
            sin_cos_phi(wi, &sin_phi_i, &cos_phi_i);
            sin_cos_phi(wo, &sin_phi_o, &cos_phi_o);
            float delta_cos_phi = pn_max(cos_phi_i * cos_phi_o + sin_phi_i * sin_phi_o, 0.0f);
            float abs_cos_theta_i = pn_abs(cos_theta(wi));
            float abs_cos_theta_o = pn_abs(cos_theta(wo));
            float sin_alpha, tan_beta;
            if (abs_cos_theta_i > abs_cos_theta_o) {
                sin_alpha = sin_theta_o;
                tan_beta = sin_theta_i / abs_cos_theta_i;
            } else {
                sin_alpha = sin_theta_i;
                tan_beta = sin_theta_o / abs_cos_theta_o;
            }
            return albedo * PN_FRAC_1_PI * (coeff_a + coeff_b * delta_cos_phi * sin_alpha * tan_beta);
        }
        case MicrofacetReflect: {  // :594-609
            float cos_theta_o = pn_abs(cos_theta(wo));
            float cos_theta_i = pn_abs(cos_theta(wi));
            Omega wh;
            bool has_wh = bisector(wo, wi, &wh);
            if (cos_theta_o == 0.0f || cos_theta_i == 0.0f || !has_wh) return black();
            wh = face_forward(wh, Omega{0, 0, 1});
            Color refl = fresnel.eval(dot(wi, wh));
            return albedo * distrib.d(wh) * distrib.g(wo, wi) * refl * pn_weak_recip(4.0f * cos_theta_o * cos_theta_i);
        }
        default: {  // FresnelBlend :669-686
            Omega wh;
            if (!bisector(wo, wi, &wh)) return black();
            Color diff = (28.0f / 23.0f * PN_FRAC_1_PI) * diffuse * (gray(1.0f) - specular) *
                         (1.0f - pn_powi(1.0f - 0.5f * pn_abs(cos_theta(wi)), 5)) *
                         (1.0f - pn_powi(1.0f - 0.5f * pn_abs(cos_theta(wo)), 5));
            Color spec = distrib.d(wh) / (4.0f * pn_abs(dot(wi, wh)) * pn_max(pn_abs(cos_theta(wi)), pn_abs(cos_theta(wo)))) *
                         schlick_fresnel(*this, dot(wi, wh));
            return diff + spec;
        }
    }
}

Prob BXDF::prob(Omega wo, Omega wi) const {
    if (kind == Fourier) return fourier_prob(*table, wo, wi);  // fourier.rs:442-485
    switch (kind) {
        case Specular: return Prob::Mass(0.0f);  // :503-505
        case DiffuseReflect:                     // :566-572
            if (wo.z * wi.z >= 0.0f) return Prob::Density(cos_hemisphere_pdf(wi));
            return Prob::Density(0.0f);
        case MicrofacetReflect: {  // :628-638
            if (!same_hemisphere(wo, wi)) return Prob::Density(0.0f);
            Omega wh;
            if (bisector(wo, wi, &wh)) return Prob::Density(distrib.pdf(wo, wh) / (4.0f * dot(wo, wh)));
            return Prob::Density(0.0f);
        }
        default: {  // FresnelBlend :708-716
            if (same_hemisphere(wo, wi)) return Prob::Density(0.0f);
            Omega wh;
            if (bisector(wo, wi, &wh))
                return Prob::Density(0.5f * (pn_abs(cos_theta(wi)) + distrib.d(wh) / (4.0f * dot(wo, wh))));
            return Prob::Density(0.0f);
        }
    }
}

void BXDF::sample(Omega wo, float r0, float r1, Color* f, Omega* wi, Prob* pr) const {
    if (kind == Fourier) return fourier_sample(*table, wo, r0, r1, f, wi, pr);  // fourier.rs:362-440, rnd2 = (u, v)
    switch (kind) {
        case Specular: {  // :462-501
            if (intrusion == Reflection) {
                specular_reflect(*this, wo, wi, f);
                *pr = Prob::Mass(1.0f);
            } else if (intrusion == Transmission && fresnel.kind == Fresnel::Dielectric) {
                specular_refract(*this, wo, fresnel.eta_front, fresnel.eta_back, wi, f);
                *pr = Prob::Mass(1.0f);
            } else if (intrusion == Hybrid && fresnel.kind == Fresnel::Dielectric) {
                float refl_coeff = fresnel.refl_coeff(cos_theta(wo));
                if (r0 < refl_coeff) {
                    specular_reflect(*this, wo, wi, f);
                    *pr = Prob::Mass(refl_coeff);
                } else {
                    specular_refract(*this, wo, fresnel.eta_front, fresnel.eta_back, wi, f);
                    *pr = Prob::Mass(1.0f - refl_coeff);
                }
            } else {
                ref_panic();
                *f = black();
                *wi = Omega{0, 0, 0};
                *pr = Prob::Mass(0.0f);
            }
            return;
        }
        case DiffuseReflect: {  // :560-564
            REF_ASSERT(cos_theta(wo) >= 0.0f);
            *wi = cos_sample_hemisphere(r0, r1);
            *f = eval(wo, *wi);
            *pr = prob(wo, *wi);
            return;
        }
        case MicrofacetReflect: {  // :611-626
            Omega wh = distrib.sample_wh(wo, r0, r1);
            Omega w = reflect(wh, wo);
            if (!same_hemisphere(wo, w)) {
                *f = black();
                *wi = Omega{0, 0, 1};
                *pr = Prob::Density(0.0f);
                return;
            }
            float pdf = distrib.pdf(wo, wh) / (4.0f * dot(wo, wh));
            *f = eval(wo, w);
            *wi = w;
            *pr = Prob::Density(pdf);
            return;
        }
        default: {  // FresnelBlend :688-706 (constructed by no material; kept for completeness)
            float u = r0, v = r1;
            Omega w;
            if (u < 0.5f) {
                float u_remapped = pn_min(u * 2.0f, 1.0f - PN_EPSILON);
                w = cos_sample_hemisphere(u_remapped, v);
                REF_ASSERT(same_hemisphere(wo, w));
            } else {
                float u_remapped = pn_fract(u * 2.0f);
                Omega wh = distrib.sample_wh(wo, u_remapped, v);
                w = reflect(wh, wo);
                if (same_hemisphere(wo, w)) {
                    *f = black();
                    *wi = Omega{0, 0, 1};
                    *pr = Prob::Mass(0.0f);
                    return;
                }
            }
            *f = eval(wo, w);
            *wi = w;
            *pr = prob(wo, w);
            return;
        }
    }
}

// ---- src/bsdf.rs ---------------------------------------------------------------------------------------
// src/bsdf.rs:104-113: the first Specular lobe, sampled with rnd2 = (0.0, 0.0); None when the material has none.
bool BSDF::sample_specular(Vec3 wo_world, Color* f, Vec3* wi, Prob* pr) const {
    Omega wo = world_to_local(wo_world);
    for (const BXDF& bxdf : *bxdfs) {
        if (bxdf.kind == BXDF::Specular) {
            Omega wi_local;
            bxdf.sample(wo, 0.0f, 0.0f, f, &wi_local, pr);
            *wi = local_to_world(wi_local);
            return true;
        }
    }
    return false;
}

BSDF bsdf_new_frame(const Interaction& isect) {  // :18-31
    Vec3 normal = hat(isect.normal);
    Vec3 bitangent = hat(cross(isect.normal, tangent(isect)));
    Vec3 tan = cross(bitangent, normal);
    REF_ASSERT(pn_abs(dot(normal, bitangent)) < 1e-4f);
    REF_ASSERT(pn_abs(dot(normal, tan)) < 1e-4f);
    REF_ASSERT(pn_abs(dot(tan, bitangent)) < 1e-4f);
    BSDF b;
    b.frame = mat3_cols(tan, bitangent, normal);
    b.bxdfs = nullptr;
    float det = dot(cross(b.frame.cols[0], b.frame.cols[1]), b.frame.cols[2]);
    REF_ASSERT(pn_abs(det - 1.0f) < 1e-4f);
    return b;
}
Omega BSDF::world_to_local(Vec3 w) const {  // :114-118 (Omega::normalize = hat)
    return hat(Vec3{dot(frame.cols[0], w), dot(frame.cols[1], w), dot(frame.cols[2], w)});
}
Vec3 BSDF::local_to_world(Omega l) const {  // :120-124
    return l.x * frame.cols[0] + l.y * frame.cols[1] + l.z * frame.cols[2];
}
Color BSDF::eval(Vec3 wo_w, Vec3 wi_w) const {  // :43-51
    Omega wi = world_to_local(wi_w);
    Omega wo = world_to_local(wo_w);
    if (wo.z == 0.0f) return black();
    Color sum = black();
    for (const BXDF& b : *bxdfs) sum = sum + b.eval(wo, wi);
    return sum;
}
float BSDF::pdf(Vec3 wo_w, Vec3 wi_w) const {  // :53-57 (Q7: a sum, not an average)
    Omega wi = world_to_local(wi_w);
    Omega wo = world_to_local(wo_w);
    float sum = 0.0f;
    for (const BXDF& b : *bxdfs) sum += b.prob(wo, wi).density();
    return sum;
}
void BSDF::sample(Vec3 wo_world, float u, float v, Color* f, Vec3* wi_out, Prob* pr) const {  // :59-103
    REF_ASSERT(u < 1.0f);
    Omega wo = world_to_local(wo_world);
    std::vector<const BXDF*> list;
    for (const BXDF& b : *bxdfs) list.push_back(&b);
    if (list.empty()) {
        *f = black();
        *wi_out = Vec3{0, 0, 0};
        *pr = Prob::Mass(0.0f);
        return;
    }
    float n = (float)list.size();
    size_t chosen_index = (size_t)(u * n);
    float remapped_u = pn_fract(u * n);
    // Q8: `let rnd2 = (v, remapped_u);`
    const BXDF* chosen = list[chosen_index];  // swap_remove: the last element takes the hole
    list[chosen_index] = list.back();
    list.pop_back();
    Color bsdf_value;
    Omega wi;
    Prob prob{};
    chosen->sample(wo, v, remapped_u, &bsdf_value, &wi, &prob);
    if (prob.is_mass) {
        *f = bsdf_value;
        *wi_out = local_to_world(wi);
        *pr = prob;
        return;
    }
    size_t other_pdf_count = 0;
    float other_pdf_sum = 0.0f;
    for (const BXDF* b : list) {
        Prob p = b->prob(wo, wi);
        if (p.is_density()) {
            other_pdf_count += 1;
            other_pdf_sum += p.density();
        }
    }
    float overall_pdf = (prob.density() + other_pdf_sum) / (float)(1 + other_pdf_count);
    Color others = black();
    for (const BXDF* b : list) others = others + b->eval(wo, wi);
    *f = bsdf_value + others;
    *wi_out = local_to_world(wi);
    *pr = Prob::Density(overall_pdf);
}

// ---- material/src/lib.rs -----------------------------------------------------------------------------
static Color c3(const float* p) { return Color{p[0], p[1], p[2]}; }
// ---- texture/src/lib.rs ------------------------------------------------------------------------------------
float Texture::noise(Point3 p) const {  // :97-137
    auto split = [](float f, int* i, float* u) {
        float fl = pn_floor(f);
        *i = (int)fl;  // `f.floor() as i32`
        *u = f - fl;
    };
    int i, j, k;
    float u, v, w;
    split(p.x * freq, &i, &u);
    split(p.y * freq, &j, &v);
    split(p.z * freq, &k, &w);
    u = u * u * (3.0f - 2.0f * u);
    v = v * v * (3.0f - 2.0f * v);
    w = w * w * (3.0f - 2.0f * w);
    Vec3 c[2][2][2];
    for (int di = 0; di < 2; ++di)
        for (int dj = 0; dj < 2; ++dj)
            for (int dk = 0; dk < 2; ++dk) {
                uint32_t index = perm_x[(size_t)((i + di) & 255)] ^ perm_y[(size_t)((j + dj) & 255)] ^ perm_z[(size_t)((k + dk) & 255)];
                c[di][dj][dk] = rand_vec[index];
            }
    float accum = 0.0f;
    for (int di = 0; di < 2; ++di)
        for (int dj = 0; dj < 2; ++dj)
            for (int dk = 0; dk < 2; ++dk) {
                Vec3 weight_v{u - (float)di, v - (float)dj, w - (float)dk};
                float dot_product = dot(c[di][dj][dk], weight_v);
                accum += ((float)di * u + (float)(1 - di) * (1.0f - u)) * ((float)dj * v + (float)(1 - dj) * (1.0f - v)) *
                         ((float)dk * w + (float)(1 - dk) * (1.0f - w)) * dot_product;
            }
    REF_ASSERT(accum >= -1.0f);
    REF_ASSERT(accum <= 1.0f);
    return accum;
}
float Texture::turbulance(Point3 p) const {  // :139-147
    float acc = 0.0f;
    for (int i = 0; i < 7; ++i) {
        float scale = pn_powi(2.0f, i);
        acc = acc + pn_powi(0.5f, i) * noise(Point3{p.x * scale, p.y * scale, p.z * scale});
    }
    return pn_abs(acc);
}
Color Texture::value(float u, float v, Point3 p) const {
    switch (kind) {
        case PBRS_TEX_CHECKER: {  // :40-49
            float sines = pn_sin(10.0f * p.x) * pn_sin(10.0f * p.y) * pn_sin(10.0f * p.z);
            return sines < 0.0f ? odd : even;
        }
        case PBRS_TEX_PERLIN:  // :150-160: a marble-like texture
            return pn_mul_add(pn_sin(freq * p.z + 10.0f * turbulance(p)), 0.5f, 0.5f) * gray(1.0f);
        default: {  // Image :211-223
            float uc = pn_clamp(u, 0.0f, 1.0f), vc = pn_clamp(v, 0.0f, 1.0f);
            // `(u * w as f32) as usize % w`: Rust's float -> int cast saturates and sends NaN to 0
            auto to_usize = [](float x) -> uint64_t { return x != x ? 0u : (x <= 0.0f ? 0u : (x >= 1.8446744e19f ? ~0ull : (uint64_t)x)); };
            uint64_t col = to_usize(uc * (float)width) % width;
            uint64_t row = to_usize(vc * (float)height) % height;
            return data[row * width + col];
        }
    }
}

Color Material::emission() const {  // :24-26, :294-296
    if (spec.kind == PBRS_MTL_DIFFUSE_LIGHT) return c3(spec.p);
    return black();
}
std::vector<BXDF> Material::bxdfs_at(const Interaction& isect) const {
    std::vector<BXDF> out;
    const float* p = spec.p;
    // `self.kd.value(isect.uv, isect.pos)`: a Solid texture is the colour in p[]
    auto colour = [&](int slot, const float* solid) { return tex[slot] ? tex[slot]->value(isect.u, isect.v, isect.pos) : c3(solid); };
    switch (spec.kind) {
        case PBRS_MTL_LAMBERTIAN:  // :180-184
            out.push_back(bxdf_lambertian(colour(0, p)));
            break;
        case PBRS_MTL_METAL: {  // :200-206
            float alpha = roughness_to_alpha(p[6]);
            MicrofacetDistrib d{MicrofacetDistrib::Beckmann, alpha, alpha};
            out.push_back(bxdf_microfacet(gray(1.0f), d, fresnel_conductor(c3(p), c3(p + 3))));
            break;
        }
        case PBRS_MTL_GLOSSY: {  // :71-78, :216-218
            float alpha = roughness_to_alpha(p[3]);
            MicrofacetDistrib d{MicrofacetDistrib::Beckmann, alpha, alpha};
            out.push_back(bxdf_microfacet(c3(p), d, fresnel_nop()));
            break;
        }
        case PBRS_MTL_MIRROR:  // :229-232
            out.push_back(bxdf_mirror(c3(p)));
            break;
        case PBRS_MTL_PLASTIC: {  // :433-445
            float alpha = (spec.flags & PBRS_MTL_FLAG_REMAP_ROUGHNESS) ? roughness_to_alpha(p[6]) : p[6];
            MicrofacetDistrib d{MicrofacetDistrib::Beckmann, alpha, alpha};
            out.push_back(bxdf_microfacet(c3(p + 3), d, fresnel_nop()));
            out.push_back(bxdf_lambertian(c3(p)));
            break;
        }
        case PBRS_MTL_DIELECTRIC:  // :265-268 (Q18: `reflect` colour on both lobes)
            out.push_back(bxdf_dielectric(c3(p + 1), 1.0f, p[0]));
            break;
        case PBRS_MTL_DIFFUSE_LIGHT:  // :291-293
            break;
        case PBRS_MTL_UBER: {  // :317-365
            float opacity = p[15], eta = p[14];
            Color transmission = gray(pn_clamp(1.0f - opacity, 0.0f, 1.0f));
            if (!is_black(transmission)) out.push_back(bxdf_transmit(transmission, 1.0f, eta));
            Color kd = colour(0, p);
            if (!is_black(kd)) out.push_back(bxdf_lambertian(kd));
            Color ks = colour(1, p + 3);
            if (!is_black(ks)) {
                float ru = p[12], rv = p[13];
                float au = ru, av = rv;
                if (spec.flags & PBRS_MTL_FLAG_REMAP_ROUGHNESS) {
                    au = roughness_to_alpha(ru);
                    av = roughness_to_alpha(rv);
                }
                MicrofacetDistrib d{MicrofacetDistrib::Beckmann, au, av};
                out.push_back(bxdf_microfacet(ks, d, fresnel_dielectric(1.0f, eta)));
            }
            if (spec.flags & PBRS_MTL_FLAG_HAS_KR) {
                Color kr = colour(2, p + 6);
                if (!is_black(kr)) out.push_back(bxdf_dielectric(kr, 1.0f, eta));
            }
            if (spec.flags & PBRS_MTL_FLAG_HAS_KT) {
                Color kt = colour(3, p + 9);
                if (!is_black(kt)) out.push_back(bxdf_transmit(kt, 1.0f, eta));
            }
            break;
        }
        case PBRS_MTL_SUBSTRATE: {  // :393-420 (Q18: degenerates to Lambert)
            Color diff = c3(p), specular = c3(p + 3);
            if (!(is_black(diff) && is_black(specular))) out.push_back(bxdf_lambertian(diff));
            break;
        }
        case PBRS_MTL_FOURIER: {  // :467-470
            BXDF b{};
            b.kind = BXDF::Fourier;
            b.table = fourier.get();
            out.push_back(b);
            break;
        }
        default: ref_panic();
    }
    return out;
}

// ---- light/src/sample_shape.rs -----------------------------------------------------------------------------
static float sphere_area(const Sphere& s) { return pn_sq(s.radius) * 4.0f * PN_PI; }  // :252-254
static Interaction sphere_sample(const Sphere& s, float u, float v) {                 // :185-195
    float theta = 2.0f * PN_PI * u;
    float phi = pn_acos(2.0f * v - 1.0f);
    Vec3 dir{pn_sin(phi) * pn_cos(theta), pn_sin(phi) * pn_sin(theta), 2.0f * v - 1.0f};
    return isect_rayless(s.center + s.radius * dir, u, v, dir);
}
static Interaction sphere_sample_towards(const Sphere& s, const Interaction& target, float u, float v) {  // :197-236
    Vec3 wc = s.center - target.pos;
    if (norm_squared(wc) < pn_sq(s.radius)) return sphere_sample(s, u, v);
    float sin_theta_max_2 = pn_sq(s.radius) / norm_squared(wc);
    float cos_theta_max = pn_sqrt(pn_max(1.0f - sin_theta_max_2, 0.0f));
    float cos_t = (1.0f - u) + u * cos_theta_max;
    float sin_theta_2 = pn_max(1.0f - pn_sq(cos_t), 0.0f);
    float phi = v * 2.0f * PN_PI;
    float dc = norm(wc);
    float ds = dc * cos_t - pn_sqrt(pn_max(pn_sq(s.radius) - norm_squared(wc) * sin_theta_2, 0.0f));
    float cos_alpha = (norm_squared(wc) + pn_sq(s.radius) - pn_sq(ds)) / (2.0f * dc * s.radius);
    float sin_alpha = pn_sqrt(pn_max(1.0f - pn_sq(cos_alpha), 0.0f));
    Vec3 normal_object_space = spherical_direction(sin_alpha, cos_alpha, phi);
    Vec3 wcx, wcy;
    make_coord_system(-hat(wc), &wcx, &wcy);
    Vec3 normal_world_space = mat3_cols(wcx, wcy, -hat(wc)) * normal_object_space;
    Point3 point_on_sphere = normal_world_space * s.radius + s.center;
    return isect_rayless(point_on_sphere, u, v, normal_world_space);
}
static bool sphere_pdf_at(const Sphere& s, const Interaction& reference, Vec3 wi, float* pdf) {  // :238-250
    Vec3 ref_to_center = s.center - reference.pos;
    if (norm_squared(ref_to_center) < pn_sq(s.radius)) {
        *pdf = 1.0f / sphere_area(s);
        return true;
    }
    float sin_theta_max_2 = pn_sq(s.radius) / norm_squared(ref_to_center);
    float cos_theta_max = pn_sqrt(pn_max(1.0f - sin_theta_max_2, 0.0f));
    float cos_t = dot(ref_to_center, wi) / (norm(ref_to_center) * norm(wi));
    if (cos_t > cos_theta_max) {
        *pdf = 1.0f / (2.0f * PN_PI * (1.0f - cos_theta_max));
        return true;
    }
    return false;
}

float SamplableShape::area() const {
    switch (kind) {
        case PBRS_SHAPE_SPHERE: return sphere_area(sphere);
        case PBRS_SHAPE_DISK: return norm_squared(disk.radial) * PN_PI;                          // :271-273
        case PBRS_SHAPE_TRIANGLE: return norm(cross(tri.p0 - tri.p1, tri.p2 - tri.p1)) * 0.5f;  // :291-293
        default: return norm(cross(quad.side_u, quad.side_v));                                   // :306-308
    }
}
bool SamplableShape::intersect(const Ray& r, Interaction* out) const {
    Shape sh{};
    sh.kind = kind;
    sh.sphere = sphere;
    sh.disk = disk;
    sh.tri = tri;
    sh.quad = quad;
    // Light-shape self tests are not scene rays: keep them out of the work counters.
    Counters* saved = g_cnt;
    g_cnt = nullptr;
    bool hit = sh.intersect(r, out);
    g_cnt = saved;
    return hit;
}
Interaction SamplableShape::sample(float u, float v) const {
    switch (kind) {
        case PBRS_SHAPE_SPHERE: return sphere_sample(sphere, u, v);
        case PBRS_SHAPE_DISK: {  // :258-264
            float cos_t, sin_t;
            concentric_sample_disk(u, v, &cos_t, &sin_t);
            Vec3 radial2 = cross(disk.normal, disk.radial);
            Vec3 cp = disk.radial * cos_t + radial2 * sin_t;
            return isect_rayless(disk.center + cp, u, v, disk.normal);
        }
        case PBRS_SHAPE_TRIANGLE: {  // :277-287
            if (u + v > 1.0f) {
                float nu = 1.0f - v, nv = 1.0f - u;
                u = nu;
                v = nv;
            }
            Point3 position = tri.p0 + (tri.p1 - tri.p0) * u + (tri.p2 - tri.p0) * v;
            Vec3 normal = hat(cross(tri.p0 - tri.p1, tri.p2 - tri.p1));
            return isect_rayless(position, u, v, normal);
        }
        default: {  // :297-302
            Point3 position = quad.origin + u * quad.side_u + v * quad.side_v;
            Vec3 normal = cross(quad.side_u, quad.side_v);
            return isect_rayless(position, u, v, normal);
        }
    }
}
Interaction SamplableShape::sample_towards(const Interaction& target, float u, float v) const {
    switch (kind) {
        case PBRS_SHAPE_SPHERE: return sphere_sample_towards(sphere, target, u, v);
        case PBRS_SHAPE_DISK: {  // :265-269
            Interaction res = sample(u, v);
            res.normal = facing(res.normal, target.normal);
            return res;
        }
        default: return sample(u, v);  // :288-290, :303-305
    }
}
bool SamplableShape::pdf_at(const Interaction& reference, Vec3 wi, float* pdf) const {
    if (kind == PBRS_SHAPE_SPHERE) return sphere_pdf_at(sphere, reference, wi, pdf);
    // default impl, sample_shape.rs:28-33 (Q4: distance, not distance squared)
    Ray ray = spawn_ray(reference, wi);
    Interaction hit;
    if (!intersect(ray, &hit)) return false;
    *pdf = distance_to(reference.pos, hit.pos) / (pn_abs(dot(hit.normal, -wi)) * area());
    return true;
}

// ---- light/src/lib.rs ------------------------------------------------------------------------------------------
Color DiffuseAreaLight::radiance_from(const Interaction& from, Vec3 wo) const {  // :127-133
    return !pn_sign_negative(dot(from.normal, wo)) ? emit_radiance : black();
}
bool DiffuseAreaLight::radiance_to(const Interaction& target, Vec3 wi, Color* le, float* pdf, Ray* vis) const {  // :141-146
    Interaction light_hit;
    if (!shape.intersect(spawn_ray(target, wi), &light_hit)) return false;
    if (!shape.pdf_at(target, wi, pdf)) return false;
    *vis = spawn_limited_ray_to(target, light_hit.pos);
    *le = emit_radiance;
    return true;
}
void DiffuseAreaLight::sample_incident_radiance(const Interaction& target, float u, float v, Color* li, Vec3* wi, Prob* pr,
                                                Ray* vis) const {  // :158-172
    Interaction point_on_light = shape.sample_towards(target, u, v);
    Vec3 w = hat(point_on_light.pos - target.pos);
    *li = radiance_from(point_on_light, -w);
    float pdf = 0.0f;
    if (!shape.pdf_at(target, w, &pdf)) pdf = 0.0f;
    *wi = w;
    *pr = Prob::Density(pdf);
    *vis = spawn_limited_ray_to(target, point_on_light.pos);
}
void DeltaLight::sample_incident_radiance(const Interaction& target, Color* li, Vec3* wi, Prob* pr, Ray* vis) const {  // :67-92
    if (kind == PBRS_DELTA_POINT) {
        *li = color * pn_weak_recip(squared_distance_to(v, target.pos));
        *wi = hat(v - target.pos);
        *vis = spawn_limited_ray_to(target, v);
        *pr = Prob::Mass(1.0f);
        return;
    }
    REF_ASSERT(world_radius > 0.0f);
    Point3 outside_world = target.pos - world_radius * 2.0f * v;
    *vis = spawn_limited_ray_to(target, outside_world);
    Point3 dummy = position_at(*vis, vis->t_max);
    REF_ASSERT(distance_to(dummy, outside_world) < norm(v) * world_radius * 0.01f);
    *li = color;
    *wi = -v;
    *pr = Prob::Mass(1.0f);
}

}  // namespace ref
