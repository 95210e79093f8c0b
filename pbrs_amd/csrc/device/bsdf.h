// device/bsdf.h — BxDF lobes and the world-space BSDF wrapper for the `shade` stage.
//
// Restates geometry/src/bxdf.rs (Omega :42-155, sampling helpers :187-210, Fresnel :284-393,
// Specular :395-506, DiffuseReflect :509-573, MicrofacetReflection :577-639),
// geometry/src/microfacet.rs (Beckmann d/lambda/g/pdf/sample_wh :36-159) and src/bsdf.rs:18-124.
// The reference builds a heap Vec<BXDF> per hit (material/src/lib.rs `bxdfs_at`); with Solid
// textures that list is constant per material, so it is a flat table in HBM (pbrs_bxdf) indexed by
// the material record.  FresnelBlend is constructed by no material and is not compiled in; the
// Fourier lobe (device/fourier.h) is compiled into the kernels that scenes with such a material run.
#pragma once
#include "shapes.h"

struct ProbD {  // math/src/prob.rs:5-8
    bool is_mass;
    float v;
};
PD ProbD density(float v) { return ProbD{false, v}; }
PD ProbD mass(float v) { return ProbD{true, v}; }
PD float dens_of(ProbD p) { return p.is_mass ? 0.0f : p.v; }
#include "fourier.h"

// ---- Omega (bxdf.rs:42-155) -------------------------------------------------------------------------------
PD float cos2_theta(f3 w) { return pn_sq(w.z); }
PD float sin2_theta(f3 w) { return 1.0f - cos2_theta(w); }
PD float sin_theta(f3 w) { return pn_sqrt(pn_max(sin2_theta(w), 0.0f)); }
PD float tan2_theta(f3 w) { return sin2_theta(w) / cos2_theta(w); }
PD float cos2_phi(f3 w) {
    float den = w.x * w.x + w.y * w.y;
    return den == 0.0f ? 1.0f : (w.x * w.x) / den;
}
PD float sin2_phi(f3 w) {
    float den = w.x * w.x + w.y * w.y;
    return den == 0.0f ? 0.0f : (w.y * w.y) / den;
}
PD void sin_cos_phi(f3 w, float& a, float& b) {  // :85-93
    float h = pn_hypot(w.x, w.y);
    if (h == 0.0f) {
        a = 0.0f;
        b = 1.0f;
    } else {
        a = w.x / h;
        b = w.y / h;
    }
}
PD bool same_hemisphere(f3 a, f3 b) { return a.z * b.z >= 0.0f; }
PD f3 face_forward(f3 self, f3 normal) { return pn_sign_negative(dot(self, normal)) ? -self : self; }

PD void concentric_sample_disk(float u, float v, float& ox, float& oy) {  // :187-200
    float x = u * 2.0f - 1.0f;
    float y = v * 2.0f - 1.0f;
    if (x == 0.0f && y == 0.0f) {
        ox = 0.0f;
        oy = 0.0f;
        return;
    }
    float r = pn_abs(pn_abs(x) > pn_abs(y) ? x : y);
    float h = pn_hypot(x, y);
    float cos_t = x / h, sin_t = y / h;
    ox = r * cos_t;
    oy = r * sin_t;
}
PD f3 cos_sample_hemisphere(float u, float v) {  // :202-206
    float x, y;
    concentric_sample_disk(u, v, x, y);
    float z = pn_sqrt(pn_max(1.0f - x * x - y * y, 0.0f));
    return mk3(x, y, z);
}

// ---- Fresnel (:308-392) ---------------------------------------------------------------------------------------
PD float fresnel_dielectric_coeff(float eta_front, float eta_back, float cos_theta_i) {
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
PD float fresnel_refl_coeff(const pbrs_bxdf& b, float cos_theta_i) {
    if (b.fresnel == PBRS_FRESNEL_NOP) return 1.0f;
    return fresnel_dielectric_coeff(b.eta[0], b.eta[1], cos_theta_i);  // Conductor: panics upstream, unreachable
}
PD f3 fresnel_eval(const pbrs_bxdf& b, float cos_theta_i) {
    if (b.fresnel != PBRS_FRESNEL_CONDUCTOR) return gray(fresnel_refl_coeff(b, cos_theta_i));
    f3 eta_i = gray(1.0f);  // Fresnel::conductor sets eta_i = white (:299-305)
    f3 eta = cdiv(ld3(b.eta), eta_i);
    f3 eta2 = cmul(eta, eta);
    f3 etak = cdiv(ld3(b.k), eta_i);
    f3 etak2 = cmul(etak, etak);
    float cos2_theta_i = pn_sq(pn_clamp(cos_theta_i, -1.0f, 1.0f));
    float sin2_theta_i = 1.0f - cos2_theta_i;
    f3 t0 = eta2 - etak2 - gray(sin2_theta_i);
    f3 a2_plus_b2 = csqrt(cmul(t0, t0) + cmul(4.0f * eta2, etak2));
    f3 t1 = a2_plus_b2 + gray(cos2_theta_i);
    f3 a = csqrt((a2_plus_b2 + t0) * 0.5f);
    f3 t2 = (2.0f * a) * cos_theta_i;
    f3 ratio_s = cdiv(t1 - t2, t1 + t2);
    f3 t3 = cos2_theta_i * a2_plus_b2 + gray(pn_sq(sin2_theta_i));
    f3 t4 = t2 * sin2_theta_i;
    f3 ratio_p = cmul(ratio_s, cdiv(t3 - t4, t3 + t4));
    return cmaxs((ratio_s + ratio_p) * 0.5f, 0.0f);
}

// ---- Beckmann (microfacet.rs:36-159); TrowbridgeReitz is built by no material ---------------------------------
PD float beckmann_d(float ax, float ay, f3 wh) {
    float tan2 = tan2_theta(wh);
    float cos4 = pn_sq(cos2_theta(wh));
    if (pn_isinf(tan2)) return 0.0f;
    float x = cos2_phi(wh) / pn_sq(ax) + sin2_phi(wh) / pn_sq(ay);
    return pn_exp(x * -tan2) / (PN_PI * ax * ay * cos4);
}
PD float beckmann_lambda(float ax, float ay, f3 w) {
    float abs_tan_theta = pn_abs(pn_sqrt(tan2_theta(w)));
    if (pn_isinf(abs_tan_theta)) return 0.0f;
    float alpha = pn_sqrt(cos2_phi(w) * pn_sq(ax) + sin2_phi(w) * pn_sq(ay));
    float a = pn_recip(alpha * abs_tan_theta);
    if (a >= 1.6f) return 0.0f;
    return (1.0f - 1.259f * a + 0.396f * a * a) / (3.535f * a + 2.181f * a * a);
}
PD float beckmann_g(float ax, float ay, f3 wo, f3 wi) { return pn_recip(1.0f + beckmann_lambda(ax, ay, wo) + beckmann_lambda(ax, ay, wi)); }
PD float beckmann_pdf(float ax, float ay, f3 wh) { return beckmann_d(ax, ay, wh) * pn_abs(wh.z); }  // :110-122
PD f3 beckmann_sample_wh(float ax, float ay, f3 wo, float u, float v) {                             // :124-154
    float tan2, phi;
    if (ax == ay) {
        float log_sample = pn_ln(1.0f - u);
        tan2 = -pn_sq(ax) * log_sample;
        phi = v * 2.0f * PN_PI;
    } else {
        float log_sample = pn_ln(1.0f - u);
        phi = pn_atan(ay / ax * pn_tan(2.0f * PN_PI * v + PN_FRAC_PI_2));
        if (v >= 0.5f) phi += PN_PI;
        float sin_phi, cos_phi;
        pn_sincos(phi, &sin_phi, &cos_phi);
        float alpha2 = pn_sq(cos_phi / ax) + pn_sq(sin_phi / ay);
        tan2 = -log_sample / alpha2;
    }
    float cos_t = pn_recip(pn_sqrt(1.0f + tan2));
    float sin_t = cos_t * pn_sqrt(tan2);
    f3 wh = spherical_direction(sin_t, cos_t, phi);
    return face_forward(wh, wo);
}

// ---- lobes ----------------------------------------------------------------------------------------------------------
// `albedo` is the lobe's colour at this hit: pbrs_bxdf::albedo, or the value of its texture (Bsdf::albedo_at).
// `lam` (a compile-time constant in the kernels that pass true: k_shade's PBRS_SHADE_LAMBERT variants, chosen at upload when
// every lobe of the scene is a Lambertian DiffuseReflect): the lobe's kind need not be read, the other kinds' code is gone.
// `fv`: the scene's Fourier tables in the kernels that carry the lobe (k_shade's PBRS_SHADE_FOURIER variants), nullptr — a
// compile-time constant — in all others.
PD f3 bxdf_eval(const pbrs_bxdf& b, f3 albedo, f3 wo, f3 wi, bool lam, const FourierView* fv = nullptr) {
    if (lam) return albedo * PN_FRAC_1_PI;
    if (fv && (fv->only || b.kind == PBRS_BXDF_FOURIER)) return fourier_eval(*fv, fv->tables[b.intrusion], wo, wi);  // fourier.rs:300-360
    if (b.kind == PBRS_BXDF_SPECULAR) return gray(0.0f);  // :458-460
    if (b.kind == PBRS_BXDF_DIFFUSE) {                    // :540-559
        if (!b.oren_nayar) return albedo * PN_FRAC_1_PI;
        float sin_theta_i = sin_theta(wi), sin_theta_o = sin_theta(wo);
        float sin_phi_i, cos_phi_i, sin_phi_o, cos_phi_o;
        sin_cos_phi(wi, sin_phi_i, cos_phi_i);
        sin_cos_phi(wo, sin_phi_o, cos_phi_o);
        float delta_cos_phi = pn_max(cos_phi_i * cos_phi_o + sin_phi_i * sin_phi_o, 0.0f);
        float aci = pn_abs(wi.z), aco = pn_abs(wo.z);
        float sin_alpha, tan_beta;
        if (aci > aco) {
            sin_alpha = sin_theta_o;
            tan_beta = sin_theta_i / aci;
        } else {
            sin_alpha = sin_theta_i;
            tan_beta = sin_theta_o / aco;
        }
        return albedo * PN_FRAC_1_PI * (b.k[0] + b.k[1] * delta_cos_phi * sin_alpha * tan_beta);
    }
    // MicrofacetReflection :594-609
    float cos_theta_o = pn_abs(wo.z), cos_theta_i = pn_abs(wi.z);
    f3 wh;
    bool has_wh = try_hat(wo + wi, wh);
    if (cos_theta_o == 0.0f || cos_theta_i == 0.0f || !has_wh) return gray(0.0f);
    wh = face_forward(wh, mk3(0.0f, 0.0f, 1.0f));
    f3 refl = fresnel_eval(b, dot(wi, wh));
    return cmul(albedo * beckmann_d(b.alpha_x, b.alpha_y, wh) * beckmann_g(b.alpha_x, b.alpha_y, wo, wi), refl) *
           pn_weak_recip(4.0f * cos_theta_o * cos_theta_i);
}
PD ProbD bxdf_prob(const pbrs_bxdf& b, f3 wo, f3 wi, bool lam, const FourierView* fv = nullptr) {
    if (fv && (fv->only || b.kind == PBRS_BXDF_FOURIER)) return fourier_prob(*fv, fv->tables[b.intrusion], wo, wi);  // fourier.rs:442-485
    if (!lam && b.kind == PBRS_BXDF_SPECULAR) return mass(0.0f);  // :503-505
    if (lam || b.kind == PBRS_BXDF_DIFFUSE) {                    // :566-572
        if (wo.z * wi.z >= 0.0f) return density(wi.z * PN_FRAC_1_PI);
        return density(0.0f);
    }
    if (!same_hemisphere(wo, wi)) return density(0.0f);  // :628-638
    f3 wh;
    if (try_hat(wo + wi, wh)) return density(beckmann_pdf(b.alpha_x, b.alpha_y, wh) / (4.0f * dot(wo, wh)));
    return density(0.0f);
}
PD void specular_reflect(const pbrs_bxdf& b, f3 albedo, f3 wo, f3& wi, f3& f) {  // :427-434
    wi = mk3(-wo.x, -wo.y, wo.z);
    f3 fr_refl = fresnel_eval(b, wi.z);
    f = cmul(fr_refl, albedo) * pn_weak_recip(pn_abs(wi.z));
}
PD void specular_refract(const pbrs_bxdf& b, f3 albedo, f3 wo, f3& wi, f3& f) {  // :436-454
    float eta_front = b.eta[0], eta_back = b.eta[1];
    float eta_i, eta_t;
    f3 normal;
    if (wo.z > 0.0f) {
        eta_i = eta_front;
        eta_t = eta_back;
        normal = mk3(0.0f, 0.0f, 1.0f);
    } else {
        eta_i = eta_back;
        eta_t = eta_front;
        normal = -mk3(0.0f, 0.0f, 1.0f);
    }
    f3 t;
    if (!refract3(normal, wo, eta_i / eta_t, t)) {
        wi = mk3(0.0f, 0.0f, 0.0f);
        f = gray(0.0f);
        return;
    }
    wi = t;
    float f_tr = 1.0f - fresnel_refl_coeff(b, t.z);
    f = (f_tr / pn_abs(t.z)) * albedo;
}
PD void bxdf_sample(const pbrs_bxdf& b, f3 albedo, f3 wo, float r0, float r1, f3& f, f3& wi, ProbD& pr, bool lam,
                    const FourierView* fv = nullptr) {
    if (fv && (fv->only || b.kind == PBRS_BXDF_FOURIER)) {  // fourier.rs:362-440, rnd2 = (u, v)
        fourier_sample(*fv, fv->tables[b.intrusion], wo, r0, r1, f, wi, pr);
        return;
    }
    if (!lam && b.kind == PBRS_BXDF_SPECULAR) {  // :462-501
        if (b.intrusion == PBRS_REFLECTION) {
            specular_reflect(b, albedo, wo, wi, f);
            pr = mass(1.0f);
        } else if (b.intrusion == PBRS_TRANSMISSION) {
            specular_refract(b, albedo, wo, wi, f);
            pr = mass(1.0f);
        } else {
            float refl_coeff = fresnel_refl_coeff(b, wo.z);
            if (r0 < refl_coeff) {
                specular_reflect(b, albedo, wo, wi, f);
                pr = mass(refl_coeff);
            } else {
                specular_refract(b, albedo, wo, wi, f);
                pr = mass(1.0f - refl_coeff);
            }
        }
        return;
    }
    if (lam || b.kind == PBRS_BXDF_DIFFUSE) {  // :560-564
        wi = cos_sample_hemisphere(r0, r1);
        f = bxdf_eval(b, albedo, wo, wi, lam);
        pr = bxdf_prob(b, wo, wi, lam);
        return;
    }
    f3 wh = beckmann_sample_wh(b.alpha_x, b.alpha_y, wo, r0, r1);  // :611-626
    f3 w = reflect3(wh, wo);
    if (!same_hemisphere(wo, w)) {
        f = gray(0.0f);
        wi = mk3(0.0f, 0.0f, 1.0f);
        pr = density(0.0f);
        return;
    }
    float pdf = beckmann_pdf(b.alpha_x, b.alpha_y, wh) / (4.0f * dot(wo, wh));
    f = bxdf_eval(b, albedo, wo, w, false);
    wi = w;
    pr = density(pdf);
}

// ---- src/bsdf.rs ------------------------------------------------------------------------------------------------------
struct Bsdf {
    f3 c0, c1, c2;  // frame columns: tangent, bitangent, normal
    const pbrs_bxdf* lobes;
    uint32_t n;
    // Textured materials (material/src/lib.rs: colours from `tex.value(isect.uv, isect.pos)`): the lobes pushed for THIS
    // hit and their colours wait in the block's LDS, entry k of a lane at [k * 256] (lobe index) and [(3 k + c) * 256]
    // (colour), filled once per vertex by bsdf_bind_textures.  nullptr: every lobe with its own constant colour.
    const uint32_t* hit_lobe;
    const float* hit_albedo;
    bool lam;  // every lobe is a Lambertian DiffuseReflect and a material has at most one (see bxdf_eval)
    f3 a0;     // ... whose colour is read once per vertex then (k_shade), not at each of the five places that ask for it
    const FourierView* fourier;  // the scene's Fourier tables, or nullptr in kernels without the lobe (see bxdf_eval)
    PD const pbrs_bxdf& lobe(uint32_t k) const { return lobes[hit_lobe ? hit_lobe[k * 256u] : k]; }
    PD f3 albedo_at(uint32_t k) const {
        if (lam) return a0;
        if (hit_albedo) return mk3(hit_albedo[(3u * k) * 256u], hit_albedo[(3u * k + 1u) * 256u], hit_albedo[(3u * k + 2u) * 256u]);
        return ld3(lobes[k].albedo);
    }
};
PD Bsdf bsdf_new_frame(const Isect& is, const pbrs_bxdf* lobes, uint32_t n) {  // :18-41
    Bsdf b;
    f3 normal = hat(is.normal);
    f3 bitangent = hat(cross(is.normal, is.tangent));
    b.c0 = cross(bitangent, normal);
    b.c1 = bitangent;
    b.c2 = normal;
    b.lobes = lobes;
    b.n = n;
    b.hit_lobe = nullptr;
    b.hit_albedo = nullptr;
    b.lam = false;
    b.a0 = gray(0.0f);
    b.fourier = nullptr;
    return b;
}
PD f3 world_to_local(const Bsdf& b, f3 w) { return hat(mk3(dot(b.c0, w), dot(b.c1, w), dot(b.c2, w))); }  // :114-118
PD f3 local_to_world(const Bsdf& b, f3 l) { return l.x * b.c0 + l.y * b.c1 + l.z * b.c2; }                 // :120-124
// src/bsdf.rs:104-113: the first Specular lobe sampled with rnd2 = (0.0, 0.0); false when the material has none
// A kernel specialised on materials of ONE lobe that is not Specular (k_shade's PBRS_SHADE_LAMBERT and PBRS_SHADE_FOURIER_ONLY variants)
PD bool one_plain_lobe(const Bsdf& b) { return b.lam || (b.fourier && b.fourier->only); }
PD bool bsdf_sample_specular(const Bsdf& b, f3 wo_world, f3& f, f3& wi_out, ProbD& pr) {
    if (one_plain_lobe(b)) return false;
    f3 wo = world_to_local(b, wo_world);
    for (uint32_t i = 0; i < b.n; ++i) {
        if (b.lobe(i).kind == PBRS_BXDF_SPECULAR) {
            f3 wi;
            bxdf_sample(b.lobe(i), b.albedo_at(i), wo, 0.0f, 0.0f, f, wi, pr, false);
            wi_out = local_to_world(b, wi);
            return true;
        }
    }
    return false;
}
// The *_l forms take wo already in the local frame (world_to_local of the same vector gives the same bits wherever it is
// evaluated; k_shade needs hit.wo's three times per vertex).
PD f3 bsdf_eval_l(const Bsdf& b, f3 wo, f3 wi_w) {                                                          // :43-51
    f3 wi = world_to_local(b, wi_w);
    if (wo.z == 0.0f) return gray(0.0f);
    f3 sum = gray(0.0f);
    for (uint32_t i = 0; i < b.n; ++i) sum = sum + bxdf_eval(b.lobe(i), b.albedo_at(i), wo, wi, b.lam, b.fourier);
    return sum;
}
PD f3 bsdf_eval(const Bsdf& b, f3 wo_w, f3 wi_w) { return bsdf_eval_l(b, world_to_local(b, wo_w), wi_w); }
PD float bsdf_pdf_l(const Bsdf& b, f3 wo, f3 wi_w) {  // :53-57 (Q7)
    f3 wi = world_to_local(b, wi_w);
    float sum = 0.0f;
    for (uint32_t i = 0; i < b.n; ++i) sum += dens_of(bxdf_prob(b.lobe(i), wo, wi, b.lam, b.fourier));
    return sum;
}
PD float bsdf_pdf(const Bsdf& b, f3 wo_w, f3 wi_w) { return bsdf_pdf_l(b, world_to_local(b, wo_w), wi_w); }
PD void bsdf_sample_l(const Bsdf& b, f3 wo, float u, float v, f3& f, f3& wi_out, ProbD& pr) {  // :59-103
    if (b.n == 0) {
        f = gray(0.0f);
        wi_out = mk3(0.0f, 0.0f, 0.0f);
        pr = mass(0.0f);
        return;
    }
    float n = (float)b.n;
    uint32_t chosen = (uint32_t)(u * n);
    float remapped_u = pn_fract(u * n);
    f3 bsdf_value, wi;
    ProbD prob;
    bxdf_sample(b.lobe(chosen), b.albedo_at(chosen), wo, v, remapped_u, bsdf_value, wi, prob, b.lam, b.fourier);  // Q8: (v, remapped_u)
    if (prob.is_mass) {
        f = bsdf_value;
        wi_out = local_to_world(b, wi);
        pr = prob;
        return;
    }
    // `swap_remove(chosen)`: the others are visited as [0 .. chosen-1, last, chosen+1 .. n-2].
    uint32_t others = one_plain_lobe(b) ? 0u : b.n - 1;  // one lobe at most
    uint32_t count = 0;
    float other_pdf_sum = 0.0f;
    f3 other_f = gray(0.0f);
    for (uint32_t k = 0; k < others; ++k) {
        uint32_t idx = (k == chosen) ? (b.n - 1) : k;
        ProbD p = bxdf_prob(b.lobe(idx), wo, wi, false, b.fourier);
        if (!p.is_mass) {
            count += 1;
            other_pdf_sum += p.v;
        }
    }
    for (uint32_t k = 0; k < others; ++k) {
        uint32_t idx = (k == chosen) ? (b.n - 1) : k;
        other_f = other_f + bxdf_eval(b.lobe(idx), b.albedo_at(idx), wo, wi, false, b.fourier);
    }
    float overall_pdf = (prob.v + other_pdf_sum) / (float)(1 + count);
    f = bsdf_value + other_f;
    wi_out = local_to_world(b, wi);
    pr = density(overall_pdf);
}
PD void bsdf_sample(const Bsdf& b, f3 wo_world, float u, float v, f3& f, f3& wi_out, ProbD& pr) {
    bsdf_sample_l(b, world_to_local(b, wo_world), u, v, f, wi_out, pr);
}
