// oracle/selftest.cpp — TEST INFRASTRUCTURE ONLY.
//
// The reference's own known-answer / property tests for the layers under the path integrator,
// transcribed with their literals.  They pin the oracle: the Rust reference
// cannot run here, so these vectors are the only outputs of it that exist.  Tests that draw from
// the unseeded thread_rng in the reference use the repo's PCG32 contract with a fixed seed here.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

#include "oracle_api.h"
#include "ref_scene.h"

using namespace ref;

namespace {

struct Log {
    std::string text;
    int failures = 0;
};
Log* g_log = nullptr;

void fail(const char* file, int line, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    char head[128];
    snprintf(head, sizeof head, "%s:%d: ", file, line);
    g_log->text += head;
    g_log->text += buf;
    g_log->text += "\n";
    g_log->failures++;
}
#define CHECK(cond, ...)                                \
    do {                                                \
        if (!(cond)) fail(__FILE__, __LINE__, __VA_ARGS__); \
    } while (0)

// math::float::linspace (float.rs:140-155)
std::vector<float> linspace(float a, float b, int count, float* spacing_out) {
    float spacing = (b - a) * (1.0f / (float)count);
    std::vector<float> v;
    for (int i = 0; i < count; ++i) v.push_back(spacing * ((float)i + 0.5f) + a);
    if (spacing_out) *spacing_out = spacing;
    return v;
}
// Omega::tesselate_hemi (bxdf.rs:159-176)
std::vector<Omega> tesselate_hemi(int count, float* d_theta, float* d_phi) {
    auto thetas = linspace(0.0f, PN_FRAC_PI_2, count, d_theta);
    auto phis = linspace(0.0f, PN_PI * 2.0f, count * 4, d_phi);
    std::vector<Omega> out;
    for (float theta : thetas)
        for (float phi : phis) {
            float s, c;
            pn_sincos(theta, &s, &c);
            out.push_back(spherical_direction(s, c, phi));
        }
    return out;
}
float o_sin_theta(Omega w) { return pn_sqrt(pn_max(1.0f - pn_sq(w.z), 0.0f)); }

bool f32_close(float a, float b) { return b / a > 0.999f && b / a < 1.001f; }  // bxdf_test.rs:10-12

// geometry/tests/bxdf_test.rs:14-26
void local_trigonometry_test() {
    Omega w{0.64f, 0.48f, 0.6f};
    CHECK(w.z == 0.6f, "cos_theta");
    CHECK(pn_sq(w.z) == 0.36f, "cos2_theta = %.9g", pn_sq(w.z));
    CHECK(1.0f - pn_sq(w.z) == 0.64f, "sin2_theta = %.9g", 1.0f - pn_sq(w.z));
    CHECK(o_sin_theta(w) == 0.8f, "sin_theta = %.9g", o_sin_theta(w));
    float h = pn_hypot(w.x, w.y);  // bxdf.rs:68-75
    CHECK(f32_close(w.x / h, 0.8f), "cos_phi = %.9g", w.x / h);
    CHECK(f32_close(w.y / h, 0.6f), "sin_phi = %.9g", w.y / h);
    CHECK(f32_close((w.x * w.x) / (w.x * w.x + w.y * w.y), 0.64f), "cos2_phi");
    CHECK(f32_close((w.y * w.y) / (w.x * w.x + w.y * w.y), 0.36f), "sin2_phi");
}

// geometry/tests/bxdf_test.rs:28-49 — bit-exact f32 goldens
void fresnel_test() {
    Fresnel glass = fresnel_dielectric(1.0f, 2.0f);
    Fresnel invert_glass = fresnel_dielectric(2.0f, 1.0f);
    const float cos_thetas[2] = {0.3f, 0.9f};
    const float expected_forward_values[2] = {0.26872247f, 0.112083375f};
    const float expected_inverted_values[2] = {1.0f, 0.1645631f};
    for (int i = 0; i < 2; ++i) {
        float af = glass.refl_coeff(cos_thetas[i]);
        float ai = invert_glass.refl_coeff(cos_thetas[i]);
        CHECK(af == expected_forward_values[i], "forward[%d] = %.9g, expected %.9g", i, af, expected_forward_values[i]);
        CHECK(ai == expected_inverted_values[i], "inverted[%d] = %.9g, expected %.9g", i, ai, expected_inverted_values[i]);
        CHECK(af == invert_glass.refl_coeff(-cos_thetas[i]), "forward symmetry %d", i);
        CHECK(ai == glass.refl_coeff(-cos_thetas[i]), "inverted symmetry %d", i);
    }
}

// geometry/tests/bxdf_test.rs:51-61
void specular_refl_test() {
    BXDF glass = bxdf_dielectric(gray(1.0f), 1.0f, 2.0f);
    Color f;
    Omega wi;
    Prob pr{};
    glass.sample(Omega{0.8f, 0.0f, 0.6f}, 0.0f, 0.0f, &f, &wi, &pr);
    CHECK(wi.x == -0.8f, "wi.x = %.9g", wi.x);
    CHECK(wi.y == -0.0f, "wi.y = %.9g", wi.y);
    CHECK(wi.z == 0.6f, "wi.z = %.9g", wi.z);
    CHECK(pr.is_mass, "pdf is not a mass");
}

float riemann_integral_hemi_pdf_i(const BXDF& bsdf, Omega wo, int count) {  // bxdf_test.rs:144-156
    float d_theta, d_phi;
    auto wis = tesselate_hemi(count, &d_theta, &d_phi);
    float sum = 0.0f;
    for (Omega wi : wis) {
        Prob pr = bsdf.prob(wo, wi);
        CHECK(pr.is_density(), "bxdf returns a mass value");
        sum += pr.density() * o_sin_theta(wi) * d_theta * d_phi;
    }
    return sum;
}
float riemann_integral_pdf_2d(const BXDF& bsdf) {  // bxdf_test.rs:158-179
    float pdf_integral = 0.0f;
    const int N = 20;
    float d_theta, d_phi;
    auto thetas = linspace(0.0f, PN_FRAC_PI_2, N, &d_theta);
    auto phis = linspace(0.0f, PN_PI * 2.0f, N * 4, &d_phi);
    for (float theta : thetas)
        for (float phi : phis) {
            float s, c;
            pn_sincos(theta, &s, &c);
            Omega wo = spherical_direction(s, c, phi);
            float marginal = riemann_integral_hemi_pdf_i(bsdf, wo, N);
            pdf_integral += marginal * s * d_theta * d_phi;
        }
    return pdf_integral * PN_FRAC_1_PI * 0.5f;
}
bool color_is_close(Color c0, Color c1) {  // bxdf_test.rs:105-114
    Vec3 v0{c0.r, c0.g, c0.b}, v1{c1.r, c1.g, c1.b};
    if (norm_squared(v0) == 0.0f || norm_squared(v1) == 0.0f) return norm_squared(v0 - v1) < 1e-6f;
    float longer = pn_max(norm_squared(v0), norm_squared(v1));
    return norm_squared(v0 - v1) / longer < 1e-3f;
}
Color montecarlo_integrate_rho(const BXDF& bsdf) {  // bxdf_test.rs:181-200
    uint64_t rng = pn_rng_init(7, 0, 0);
    const int TRIALS = 800;
    Color acc = black();
    for (int t = 0; t < TRIALS; ++t) {
        float u = pn_rng_f32(&rng), v = pn_rng_f32(&rng);
        Omega wo = hat(Vec3{0.2f, -0.1f, 0.9f});
        Color f;
        Omega wi;
        Prob pr{};
        bsdf.sample(wo, u, v, &f, &wi, &pr);
        CHECK(!has_nan(wi), "wi has nan");
        CHECK(pr.is_density(), "mass from a diffuse brdf");
        acc = acc + f * pn_abs(wi.z) * pn_weak_recip(pr.v);
    }
    return (1.0f / (float)TRIALS) * acc;
}
// geometry/tests/bxdf_test.rs:63-70, :116-138
void diffuse_refl_test() {
    Color albedo = rgb(1.0f, 2.0f, 5.0f);
    BXDF brdfs[2] = {bxdf_lambertian(albedo), bxdf_oren_nayar(albedo, pn_to_radians(0.0f))};
    for (const BXDF& brdf : brdfs) {
        float i1 = riemann_integral_hemi_pdf_i(brdf, hat(Vec3{0.48f, 0.64f, 0.6f}), 25);
        CHECK(pn_abs(i1 - 1.0f) < 1e-3f, "hemisphere pdf integrates to %.6f", i1);
        float i2 = riemann_integral_pdf_2d(brdf);
        CHECK(pn_abs(i2 - 1.0f) < 4e-3f, "2D hemisphere pdf integrates to %.6f", i2);
        Color rho = montecarlo_integrate_rho(brdf);
        CHECK(color_is_close(albedo, rho), "MC rho = (%.4f %.4f %.4f)", rho.r, rho.g, rho.b);
    }
}

// geometry/tests/bxdf_test.rs:202-231
void play_with_mf_brdf() {
    Color albedo = rgb(3.0f, 3.4f, 2.9f);
    MicrofacetDistrib mf{MicrofacetDistrib::Beckmann, 0.2f, 0.3f};
    BXDF brdf = bxdf_microfacet(albedo, mf, fresnel_nop());
    Omega wo = hat(Vec3{0.6f, 0.8f, 0.3f});
    CHECK(pn_abs(norm_squared(wo) - 1.0f) < 1e-3f, "wo not unit");
    uint64_t rng = pn_rng_init(11, 0, 0);
    for (int t = 0; t < 64; ++t) {
        float u = pn_rng_f32(&rng), v = pn_rng_f32(&rng);
        Omega wh_from_mf = mf.sample_wh(wo, u, v);
        Color f;
        Omega wi;
        Prob pr{};
        brdf.sample(wo, u, v, &f, &wi, &pr);
        if (is_black(f)) continue;
        Omega wh;
        if (try_hat(wo + wi, &wh)) {
            float d2 = norm_squared(wh - wh_from_mf);
            CHECK(d2 < 1e-3f, "wh mismatch, dist^2 = %g", d2);
        }
    }
}

// geometry/tests/microfacet_test.rs:103-136
float integrate_differental_area(const MicrofacetDistrib& mf) {
    const int N = 50;
    float integral = 0.0f, d_theta, d_phi;
    auto thetas = linspace(0.0f, PN_FRAC_PI_2, N, &d_theta);
    auto phis = linspace(0.0f, PN_PI * 2.0f, N * 4, &d_phi);
    for (float theta : thetas)
        for (float phi : phis) {
            float s, c;
            pn_sincos(theta, &s, &c);
            Omega wh = spherical_direction(s, c, phi);
            float da = mf.d(wh);
            CHECK(!pn_isinf(da), "d() infinite");
            integral += da * c * (s * d_theta * d_phi);
        }
    return integral;
}
float integrate_masking(const MicrofacetDistrib& mf, Omega w) {
    const int N = 100;
    float integral = 0.0f, d_theta, d_phi;
    auto thetas = linspace(0.0f, PN_FRAC_PI_2, N, &d_theta);
    auto phis = linspace(0.0f, PN_PI * 2.0f, N * 4, &d_phi);
    for (float theta : thetas)
        for (float phi : phis) {
            float s, c;
            pn_sincos(theta, &s, &c);
            Omega wh = spherical_direction(s, c, phi);
            float masked = mf.d(wh) * mf.g1(w) * pn_max(dot(w, wh), 0.0f);
            integral += masked * (s * d_theta * d_phi);
        }
    return integral;
}
// geometry/tests/microfacet_test.rs:12-25
void diff_area_validate() {
    MicrofacetDistrib mfs[2] = {{MicrofacetDistrib::Beckmann, 0.2f, 0.2f}, {MicrofacetDistrib::TrowbridgeReitz, 0.2f, 0.2f}};
    for (auto& mf : mfs) {
        float pa = integrate_differental_area(mf);
        CHECK(pn_abs(pa - 1.0f) < 4e-3f, "projected area = %.6f", pa);
        Omega w{0.48f, 0.64f, 0.6f};
        float ma = integrate_masking(mf, w);
        CHECK(pn_abs(ma - w.z) < 1e-3f, "masked area = %.6f", ma);
    }
}
// geometry/tests/microfacet_test.rs:27-49, :68-82
void pdf_integral_validate() {
    auto alphas = linspace(0.1f, 0.9f, 8, nullptr);
    float d_theta, d_phi;
    auto whs = tesselate_hemi(70, &d_theta, &d_phi);
    auto wos = tesselate_hemi(3, nullptr, nullptr);
    for (float alpha : alphas) {
        MicrofacetDistrib mfs[2] = {{MicrofacetDistrib::Beckmann, alpha, alpha}, {MicrofacetDistrib::TrowbridgeReitz, alpha, alpha}};
        for (auto& mf : mfs) {
            float full = 0.0f;
            for (Omega wo : wos) {
                float marginal = 0.0f;
                for (Omega wh : whs) marginal += mf.pdf(wo, wh) * o_sin_theta(wh) * d_theta * d_phi;
                full += marginal;
            }
            float integral = full / (float)wos.size();
            CHECK(pn_abs(integral - 1.0f) < 2e-3f, "alpha %.3f kind %d: pdf integral %.6f", alpha, (int)mf.kind, integral);
        }
    }
}
// geometry/tests/microfacet_test.rs:165-194 (seeded here)
void beckmann_rho() {
    uint64_t rng = pn_rng_init(13, 0, 0);
    const float alphas[4] = {0.05f, 0.1f, 0.3f, 0.6f};
    for (float alpha : alphas) {
        MicrofacetDistrib mf{MicrofacetDistrib::Beckmann, alpha, alpha};
        BXDF refl = bxdf_microfacet(gray(1.0f), mf, fresnel_nop());
        Omega wo = hat(Vec3{0.2f, 0.6f, 0.5f});
        std::vector<float> norms;
        for (int t = 0; t < 500; ++t) {
            float u = pn_rng_f32(&rng), v = pn_rng_f32(&rng);
            Color color;
            Omega wi;
            Prob pr{};
            refl.sample(wo, u, v, &color, &wi, &pr);
            color = color * pn_abs(wi.z);
            float n = norm(Vec3{color.r, color.g, color.b});
            if (pr.density() != 0.0f) norms.push_back(n / pr.density());
        }
        float mean = 0.0f;
        for (float n : norms) mean += n;
        mean = mean / (float)norms.size();
        float var = 0.0f;
        for (float n : norms) var += pn_sq(n - mean);
        var = var / (float)norms.size();
        float s3 = pn_sqrt(3.0f);
        CHECK(s3 >= mean - 2.0f * var && s3 <= mean + 2.0f * var, "alpha %.2f: mean %.4f var %.4f", alpha, mean, var);
        CHECK(var / alpha >= 0.0f && var / alpha <= 2.0f, "alpha %.2f: var/alpha = %.4f", alpha, var / alpha);
    }
}

// shape/tests/frame_test.rs:9-15
void quad_frame_test() {
    Shape q{};
    q.kind = PBRS_SHAPE_QUAD;
    q.quad = ParallelQuad{Vec3{-1.0f, -1.0f, 0.0f}, Vec3{2.0f, 0.0f, 0.0f}, Vec3{0.0f, 2.0f, 0.0f}};  // new_xy((-1,1),(-1,1),0)
    Ray ray = ray_new(Vec3{0.5f, 0.5f, -1.0f}, Vec3{-0.2f, -0.2f, 1.0f});
    Interaction isect;
    bool hit = q.intersect(ray, &isect);
    CHECK(hit, "quad not hit");
    if (hit) CHECK(has_valid_frame(isect), "invalid frame");
}
// shape/tests/frame_test.rs:17-52 (+ math/src/hcm.rs:585-594 doctest)
void custom_frame_test() {
    Vec3 normal = hat(Vec3{-0.3f, 0.5f, 1.0f});
    Vec3 dpdu, dpdv;
    make_coord_system(normal, &dpdu, &dpdv);
    CHECK(pn_abs(dot(normal, dpdu)) < 1e-4f, "normal/tangent not perp");
    CHECK(pn_abs(dot(normal, dpdv)) < 1e-4f, "normal/bitangent not perp");
    Vec3 c[3] = {dpdu, dpdv, normal};
    float frob = 0.0f;  // ||F F^T - I||_F^2
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float e = c[0][i] * c[0][j] + c[1][i] * c[1][j] + c[2][i] * c[2][j] - (i == j ? 1.0f : 0.0f);
            frob += e * e;
        }
    CHECK(frob < 1e-6f, "frame not orthonormal: %g", frob);
    Interaction isect = with_dpdu(isect_rayless(Vec3{3.0f, 2.5f, 2.0f}, 0.2f, 0.8f, normal), dpdu);
    CHECK(norm_squared(tangent(isect) - dpdu) < 1e-6f, "tangent mismatch");
    CHECK(norm_squared(isect.normal - normal) < 1e-6f, "normal mismatch");

    Vec3 v0 = hat(Vec3{0.3f, 0.4f, -0.6f});  // hcm.rs:585-594
    Vec3 v1, v2;
    make_coord_system(v0, &v1, &v2);
    Vec3 d[3] = {v0, v1, v2};
    frob = 0.0f;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float e = d[0][i] * d[0][j] + d[1][i] * d[1][j] + d[2][i] * d[2][j] - (i == j ? 1.0f : 0.0f);
            frob += e * e;
        }
    CHECK(frob < PN_EPSILON, "doctest basis not orthonormal: %g", frob);
}
// shape/tests/frame_test.rs:54-85
void sphere_test() {
    Shape sp{};
    sp.kind = PBRS_SHAPE_SPHERE;
    sp.sphere = Sphere{Vec3{3.0f, 4.0f, 5.0f}, 1.6f};
    const float scales[7] = {0.001f, 0.01f, 0.1f, 1.0f, 10.0f, 100.0f, 1000.0f};
    Vec3 dir_0{1.5f, 2.0f, 2.5f};
    for (float s : scales) {
        Ray r = ray_new(Vec3{0.1f, 0.2f, 0.1f}, dir_0 * s);
        r.t_max = 1.0f / s;
        Interaction i;
        CHECK(!sp.intersect(r, &i), "dir_0 scale %g intersects", s);
        CHECK(!sp.occludes(r), "dir_0 scale %g occludes", s);
    }
    // dir_1's segment ends INSIDE the sphere (far root t = 1.194/s > t_max = 1/s).  The reference test
    // asserts `sphere.occludes(&r)` for it (:78), but Sphere::occludes (shape/src/simple.rs:287) needs
    // BOTH roots inside (eps, t_max) (Q13), so that assertion cannot hold against the reference's own
    // code: the test is failing upstream.  The oracle follows the code; the expectation here is what
    // the code returns.  dir_2's segment passes through the sphere and occludes as the test says.
    Vec3 dirs[2] = {Vec3{3.0f, 4.0f, 5.0f}, Vec3{4.8f, 6.4f, 8.0f}};
    const bool expect_occluded[2] = {false, true};
    for (int k = 0; k < 2; ++k)
        for (float s : scales) {
            Vec3 dir = dirs[k];
            Ray r = ray_new(Vec3{0.1f, 0.2f, 0.1f}, dir * s);
            r.t_max = 1.0f / s;
            Interaction i;
            bool hit = sp.intersect(r, &i);
            CHECK(hit, "scale %g misses", s);
            CHECK(sp.occludes(r) == expect_occluded[k], "dir_%d scale %g: occludes != %d", k + 1, s, (int)expect_occluded[k]);
            if (hit) {
                float dist2 = squared_distance_to(i.pos, sp.sphere.center);
                CHECK(pn_abs(dist2 - pn_sq(sp.sphere.radius)) <= 1e-4f, "scale %g: |p-c|^2 - r^2 = %g", s, dist2 - pn_sq(sp.sphere.radius));
            }
        }
}

// shape/src/blas.rs:497-522 — must not trip any assert
void tricky_triangle() {
    float pos[9] = {10.3457699f, 27.3706398f, -21.2291069f, 10.3457699f, 13.3905125f, -21.1700611f, 7.22226286f, 13.3905125f, -21.1700611f};
    float nrm[9] = {0.0f, 0.00419999985f, 1.0f, 0.0f, 0.00419999985f, 1.0f, 0.0f, 0.00419999985f, 1.0f};
    float uvs[6] = {0, 0, 0, 0, 0, 0};
    uint32_t idx[3] = {0, 1, 2};
    pbrs_mesh_spec m{3, 1, pos, nrm, uvs, idx};
    Diag d;
    g_diag = &d;
    Shape sh{};
    sh.kind = PBRS_SHAPE_MESH;
    sh.mesh = std::shared_ptr<TriangleMesh>(mesh_from_soa(m));
    Ray ray = ray_new(Vec3{0.0f, 23.0f, 30.0f}, Vec3{0.219424784f, -0.0887561888f, -1.08688462f});
    Interaction i;
    sh.intersect(ray, &i);
    g_diag = nullptr;
    CHECK(d.panics == 0, "%llu assert sites tripped", (unsigned long long)d.panics);
}

// math/src/hcm.rs:671-705 (+ doctest :133-143)
void reflect_refract_test() {
    Vec3 r = reflect(Vec3{0, 1, 0}, Vec3{2.0f, 1.0f, 0.5f});
    CHECK(norm_squared(r - Vec3{-2.0f, 1.0f, -0.5f}) < PN_EPSILON, "reflect = (%g %g %g)", r.x, r.y, r.z);
    Vec3 normal = Vec3{0, 1, 0} * 6.0f;
    Vec3 wi = hat(Vec3{1.0f, 1.0f, 0.0f});
    Vec3 wo{-0.5f, -0.5f * pn_sqrt(3.0f), 0.0f};
    Vec3 out;
    bool transmit = refract(normal, wi, pn_sqrt(0.5f), &out);
    CHECK(transmit, "full reflection at n = sqrt(.5)");
    CHECK(norm_squared(wo - out) < PN_EPSILON, "refract = (%g %g %g)", out.x, out.y, out.z);
    CHECK(!refract(normal, hat(Vec3{0.51f, pn_sqrt(0.75f), 0.0f}), 2.0f, &out), "0.51 should fully reflect");
    CHECK(refract(normal, hat(Vec3{0.49f, pn_sqrt(0.75f), 0.0f}), 2.0f, &out), "0.49 should transmit");
    Vec3 a{1.0f, 2.5f, 0.0f}, b{0.6f, 0.0f, 0.0f};
    Vec3 c = b - projected_onto(b, a);
    CHECK(pn_abs(dot(c, a)) < PN_EPSILON, "projected_onto #1: %g", dot(c, a));
    a = Vec3{0.19f, -0.00f, 0.98f};
    b = Vec3{-9762.44f, -17.83f, 1851.39f};
    c = b - projected_onto(b, a);
    CHECK(pn_abs(dot(c, a)) < PN_EPSILON, "projected_onto #2: %g", dot(c, a));
}

// math/src/float.rs doctests :73-112, :135-139
void float_doctests() {
    auto cathetus = [](float s, float o) { return pn_sqrt(pn_max(pn_sq(s) - pn_sq(o), 0.0f)); };
    CHECK(cathetus(1.0f, 0.6f) == 0.8f, "cathetus(1, .6) = %.9g", cathetus(1.0f, 0.6f));
    CHECK(cathetus(1.0f, -0.6f) == 0.8f, "cathetus(1, -.6)");
    CHECK(0.75f / 2.5f == 0.3f, "try_divide(0.75, 2.5)");
    CHECK(pn_weak_recip(0.0f) == 0.0f && pn_weak_recip(4.0f) == 0.25f, "weak_recip");
    float spacing;
    auto nums = linspace(0.0f, 12.0f, 4, &spacing);
    CHECK(spacing == 3.0f, "linspace spacing");
    CHECK(nums[0] == 1.5f && nums[1] == 4.5f && nums[2] == 7.5f && nums[3] == 10.5f, "linspace values");
}

// geometry/src/transform.rs:343-357 restated on AffineTransform (the instance transform actually used,
// transform.rs:197); Mat4::rotater per math/src/hcm.rs:508-520, translater :489-493.
void bbox_transform_test() {
    Vec3 axis{0.6f, 0.8f, 0.0f};
    float s, c;
    pn_sincos(0.3f, &s, &c);
    Mat4 rot{{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}};
    for (int i = 0; i < 3; ++i) {
        Vec3 base{0, 0, 0};
        base.at(i) = 1.0f;
        Vec3 vc = dot(base, axis) * axis / dot(axis, axis);
        Vec3 v1 = base - vc;
        Vec3 v2 = cross(v1, hat(axis));
        Vec3 col = vc + v1 * c + v2 * s;
        rot.cols[i] = Vec4{col.x, col.y, col.z, 0.0f};
    }
    // forward = R * T(7,8,-13): fourth column = R * t
    Vec3 t{7.0f, 8.0f, -13.0f};
    Vec3 rt = mul_vec3(rot, t);
    Mat4 fwd = rot;
    fwd.cols[3] = Vec4{rt.x, rt.y, rt.z, 1.0f};
    Mat4 inv = transpose(rot);  // only forward is used by bbox/point application
    Instance inst;
    inst.forward = fwd;
    inst.inverse = inv;
    auto sh = std::make_shared<Shape>();
    sh->kind = PBRS_SHAPE_CUBOID;
    sh->cuboid = Cuboid{Vec3{-0.3f, 0.4f, 0.8f}, Vec3{3.4f, 2.3f, 4.4f}};
    inst.shape = sh;
    BBox tb = inst.bbox();
    for (int i = 0; i < 8; ++i) {
        Vec3 corner{(i & 1) ? 3.4f : -0.3f, (i & 2) ? 2.3f : 0.4f, (i & 4) ? 4.4f : 0.8f};
        Vec4 p = mul(fwd, Vec4{corner.x, corner.y, corner.z, 1.0f});
        bool inside = tb.min.x <= p.x && p.x <= tb.max.x && tb.min.y <= p.y && p.y <= tb.max.y && tb.min.z <= p.z && p.z <= tb.max.z;
        CHECK(inside, "corner %d outside the transformed bbox", i);
    }
}

// light/tests/shape_sample_test.rs:9-20, :69-90
void sphere_sample_pdf_integrate() {
    SamplableShape s{};
    s.kind = PBRS_SHAPE_SPHERE;
    s.sphere = Sphere{Vec3{5.0f, 6.0f, 12.0f}, 2.0f};
    Interaction p = isect_rayless(Vec3{0, 0, 0}, 0.0f, 0.0f, Vec3{0, 0, 1});
    auto uvec = linspace(0.0f, 1.0f, 20, nullptr);
    float sin2_t = pn_sq(s.sphere.radius) / squared_distance_to(p.pos, s.sphere.center);
    float cos_t = pn_sqrt(pn_max(1.0f - sin2_t, 0.0f));
    float cone = 2.0f * PN_PI * (1.0f - cos_t);
    float integral = 0.0f;
    for (float u : uvec)
        for (float v : uvec) {
            Interaction pt = s.sample_towards(p, u, v);
            CHECK(pn_abs(distance_to(pt.pos, s.sphere.center) - s.sphere.radius) < 1e-5f, "sampled radius off");
            Vec3 wi = pt.pos - p.pos;
            float pdf = 0.0f;
            bool ok = s.pdf_at(p, wi, &pdf);
            CHECK(ok, "pdf_at None for a sampled direction (u=%g v=%g)", u, v);
            integral += pdf * cone;
        }
    integral = integral / (float)(uvec.size() * uvec.size());
    CHECK(pn_abs(integral - 1.0f) < 1e-2f, "Int(pdf) = %.6f", integral);
}
// light/tests/shape_sample_test.rs:22-66
void observe_sphere_sample_towards() {
    {  // literal regression input, :28-39 — must not trip an assert
        SamplableShape s{};
        s.kind = PBRS_SHAPE_SPHERE;
        s.sphere = Sphere{Vec3{9.44999981f, 20.0f, -8.0f}, 0.2f};
        Diag d;
        g_diag = &d;
        Interaction target = with_dpdu(isect_new(Vec3{-18.3287563f, 19.3762169f, 0.0f}, 56.0f, 0.0417810902f, 0.968810856f, -Vec3{0, 0, 1},
                                                 Vec3{0.327299207f, -0.203146726f, -1.0f}),
                                       Vec3{0.0f, -1.0f, 0.0f});
        s.sample_towards(target, 0.9898101f, 0.724872649f);
        g_diag = nullptr;
        CHECK(d.panics == 0, "regression input tripped %llu asserts", (unsigned long long)d.panics);
    }
    SamplableShape s{};
    s.kind = PBRS_SHAPE_SPHERE;
    s.sphere = Sphere{Vec3{0, 0, 0}, 1.5f};
    Interaction target = isect_rayless(Vec3{0.0f, 3.0f, 0.0f}, 0.0f, 0.0f, Vec3{0.6f, -0.8f, 0.0f});
    auto uvec = linspace(0.0f, 1.0f, 10, nullptr);
    for (float u : uvec)
        for (float v : uvec) {
            Interaction pt = s.sample_towards(target, u, v);
            Vec3 radial = pt.pos - s.sphere.center;
            CHECK(pn_abs(norm_squared(radial) - pn_sq(s.sphere.radius)) < 1e-3f, "not on the sphere");
            CHECK(norm_squared(cross(pt.normal, radial)) < 1e-3f, "normal not radial");
            Ray r = spawn_ray(target, pt.pos - target.pos);
            Interaction hit;
            bool ok = s.intersect(r, &hit);
            CHECK(ok, "sampled point not re-intersected (u=%g v=%g)", u, v);
            if (ok) CHECK(squared_distance_to(hit.pos, pt.pos) < 1e-1f, "re-intersection far from the sample");
        }
}

// material/src/lib.rs:513-547, second half: Lambert f*cos/pdf == albedo for rnd2 = (0.8, 0.5)
void lambertian_test() {
    Interaction isect = with_dpdu(isect_rayless(Vec3{0.4f, 0.5f, 3.0f}, 0.3f, 0.8f, Vec3{0.36f, 0.48f, 0.8f}), Vec3{-0.8f, 0.6f, 0.0f});
    CHECK(has_valid_frame(isect), "invalid frame");
    Material m{};
    m.spec.kind = PBRS_MTL_LAMBERTIAN;
    m.spec.p[0] = m.spec.p[1] = m.spec.p[2] = 1.0f;
    auto bxdfs = m.bxdfs_at(isect);
    CHECK(bxdfs.size() == 1, "expected one bxdf");
    Color f;
    Omega wi;
    Prob p{};
    bxdfs[0].sample(hat(Vec3{0.7f, 0.5f, 0.3f}), 0.8f, 0.5f, &f, &wi, &p);
    Color c1 = f * pn_abs(wi.z) * pn_weak_recip(p.density());
    Vec3 d{c1.r - 1.0f, c1.g - 1.0f, c1.b - 1.0f};
    CHECK(norm_squared(d) < 1e-5f, "f*cos/pdf = (%.6f %.6f %.6f)", c1.r, c1.g, c1.b);
}


// src/bsdf.rs:159-226 (`mf_refl_test`): BSDF::new_frame reproduces the frame the hit was built with, world_to_local
// inverts `frame * wo_local`, and a Beckmann (roughness 0.2, Fresnel::Nop) reflection lobe sampled on a 10 x 10 grid
// returns densities (never a probability mass) whose responses f |wi . n| / pdf are not NaN.
void mf_refl_test() {
    const float alpha = roughness_to_alpha(0.2f);
    MicrofacetDistrib distrib{MicrofacetDistrib::Beckmann, alpha, alpha};
    std::vector<BXDF> bxdfs{bxdf_microfacet(rgb(1.0f, 1.0f, 1.0f), distrib, fresnel_nop())};
    Vec3 normal = hat(Vec3{-0.6f, 0.5f, 0.2f});
    Vec3 dpdu, dpdv;
    make_coord_system(normal, &dpdu, &dpdv);
    Mat3 frame = mat3_cols(dpdu, dpdv, normal);
    auto frob_diff = [](const Mat3& a, const Mat3& b) {  // (a - b).frobenius_norm_squared(), hcm.rs:431-433
        float sum = 0.0f;
        for (int i = 0; i < 3; ++i) sum += norm_squared(a.cols[i] - b.cols[i]);
        return sum;
    };
    {  // frame * frame^T - I
        Mat3 t = mat3_cols(Vec3{frame.cols[0][0], frame.cols[1][0], frame.cols[2][0]}, Vec3{frame.cols[0][1], frame.cols[1][1], frame.cols[2][1]},
                           Vec3{frame.cols[0][2], frame.cols[1][2], frame.cols[2][2]});
        Mat3 prod = mat3_cols(frame * t.cols[0], frame * t.cols[1], frame * t.cols[2]);  // hcm.rs:436-446
        Mat3 eye = mat3_cols(Vec3{1, 0, 0}, Vec3{0, 1, 0}, Vec3{0, 0, 1});
        CHECK(frob_diff(prod, eye) < 1e-6f, "frame is not orthonormal: %g", frob_diff(prod, eye));
    }
    Interaction isect = with_dpdu(isect_rayless(Vec3{3.0f, 2.5f, 2.0f}, 0.2f, 0.8f, normal), dpdu);
    BSDF bsdf = bsdf_new_frame(isect);
    bsdf.bxdfs = &bxdfs;
    CHECK(frob_diff(frame, bsdf.frame) < 1e-6f, "BSDF frame differs from the hit's frame: %g", frob_diff(frame, bsdf.frame));
    Vec3 wo_local = hat(Vec3{0.6f, 0.0f, 0.8f});
    Vec3 wo_world = frame * wo_local;
    CHECK(pn_abs(dot(wo_world, normal) - wo_local[2]) < 1e-3f, "wo_world . n = %g", dot(wo_world, normal));
    {
        Omega actual = bsdf.world_to_local(wo_world);
        CHECK(norm_squared(wo_local - actual) < 1e-6f, "world_to_local(frame * wo) is off by %g", norm_squared(wo_local - actual));
    }
    auto uvec = linspace(0.0f, 1.0f, 10, nullptr);
    int responses = 0;
    for (float u : uvec)
        for (float v : uvec) {
            Color f;
            Vec3 wi;
            Prob pr{};
            bsdf.sample(wo_world, u, v, &f, &wi, &pr);
            CHECK(!pr.is_mass, "a microfacet reflection returned a probability mass at (%g, %g)", u, v);
            if (pr.is_mass) continue;
            const float pdf = pr.density();
            Color response = f * pn_abs(dot(wi, isect.normal)) * pn_weak_recip(pdf);
            if (pdf > 0.0f) {
                CHECK(!pn_isnan(response.r), "NaN response at (%g, %g)", u, v);
                ++responses;
            }
        }
    CHECK(responses > 0, "no sample of the grid had a positive density");
}

// radiometry/src/spectrum.rs:471-496 test_temperature_to_color: five temperatures, each channel within 3e-3
void temperature_to_color_test() {
    const float temperatures[5] = {2700.0f, 3500.0f, 4500.0f, 5000.0f, 6500.0f};
    const Color true_colors[5] = {Color{0.533494f, 0.221571f, 0.052902f}, Color{1.007905f, 0.574979f, 0.261424f}, Color{1.215729f, 0.883807f, 0.610254f},
                                  Color{1.190014f, 0.942058f, 0.747937f}, Color{0.922219f, 0.869496f, 0.915217f}};
    for (int i = 0; i < 5; ++i) {
        Color c = temperature_to_color(temperatures[i]);
        const float THRESHOLD = 3e-3f;
        CHECK(pn_abs(c.r - true_colors[i].r) <= THRESHOLD && pn_abs(c.g - true_colors[i].g) <= THRESHOLD && pn_abs(c.b - true_colors[i].b) <= THRESHOLD,
              "%g K: (%g, %g, %g), expected (%g, %g, %g)", temperatures[i], c.r, c.g, c.b, true_colors[i].r, true_colors[i].g, true_colors[i].b);
    }
}
// math/src/spline.rs:314-331 tridiagonal_test
void tridiagonal_test() {
    std::vector<float> a(7, 1.0f), b(7, 1.5f), c(7, 1.0f), rhs(7, 7.0f), x;
    rhs[0] = 5.0f;
    rhs[6] = 5.0f;
    CHECK(tridiagonal(a, b, c, rhs, &x) && x.size() == 7, "tridiagonal refused a 7 x 7 system");
    float total_square_error = 0.0f;
    for (size_t i = 0; i < x.size(); ++i) total_square_error += pn_powi(x[i] - 2.0f, 2);
    CHECK(total_square_error <= 1e-6f, "sum of squared errors %g", total_square_error);
}
// math/src/spline.rs:333-345 cubic_spline_solve_test and :347-360 cubic_spline_eval_test
void cubic_spline_test() {
    const float x[5] = {0.2f, 0.4f, 0.6f, 0.8f, 1.0f}, y[5] = {0.97986f, 0.91777f, 0.80803f, 0.63860f, 0.38437f};
    std::vector<std::pair<float, float>> pairs;
    for (int i = 0; i < 5; ++i) pairs.push_back({x[i], y[i]});
    std::vector<float> actual_m;
    CHECK(cubic_spline_zero_hess(pairs, &actual_m) && actual_m.size() == 3, "cubic_spline_zero_hess refused five samples");
    const float expected_m[3] = {-1.5021f, -1.1390f, -2.8952f};
    float total_error_squared = 0.0f;
    for (size_t i = 0; i < actual_m.size() && i < 3; ++i) total_error_squared += pn_powi(expected_m[i] - actual_m[i], 2);
    CHECK(total_error_squared <= 1e-6f, "second derivatives off: %g", total_error_squared);
    CubicSpline spline;
    CHECK(CubicSpline::from_samples(pairs, &spline), "from_samples refused five samples");
    const float at[4] = {0.3f, 0.5f, 0.7f, 0.9f}, want[4] = {0.9526f, 0.8695f, 0.7334f, 0.5187f};
    for (int i = 0; i < 4; ++i) CHECK(pn_abs(spline.evaluate(at[i]) - want[i]) <= 3e-5f, "spline(%g) = %g, expected %g", at[i], spline.evaluate(at[i]), want[i]);
    // outside the samples the end values are returned (:41-45)
    CHECK(spline.evaluate(0.1f) == y[0] && spline.evaluate(1.5f) == y[4], "values outside the domain");
}

// math/src/spline.rs:384-408 test_find_interval
void find_interval_test() {
    const int array[8] = {16, 21, 32, 43, 55, 62, 73, 82};
    for (int x = 0; x < 10; ++x) {
        const int pivot = x * 10 + 5;
        for (int strict = 0; strict < 2; ++strict) {
            size_t start = find_interval(8, [&](size_t i) { return strict ? array[i] < pivot : array[i] <= pivot; });
            size_t end = start + 1;
            if (pivot < array[0]) {
                CHECK(start == 0, "pivot %d: start %zu", pivot, start);
            } else if (pivot > array[7]) {
                CHECK(end == 7, "pivot %d: end %zu", pivot, end);
            } else if (strict) {
                CHECK(array[start] < pivot && array[end] >= pivot, "pivot %d: [%zu, %zu]", pivot, start, end);
            } else {
                CHECK(array[start] <= pivot && array[end] > pivot, "pivot %d: [%zu, %zu]", pivot, start, end);
            }
        }
    }
}
// math/src/spline.rs:410-434 catmull_test (the nodes are the zenith cosines of a real .bsdf table)
void catmull_test() {
    const std::vector<float> values = {
        -1.0f, -0.9992586f, -0.99751526f, -0.9947773f, -0.9910477f, -0.98633015f, -0.98062944f, -0.97395116f, -0.9663021f, -0.95768976f,
        -0.94812274f, -0.9376106f, -0.9261639f, -0.913794f, -0.9005131f, -0.88633466f, -0.8712726f, -0.85534215f, -0.838559f, -0.82093996f,
        -0.80250263f, -0.7832653f, -0.7632472f, -0.7424683f, -0.7209493f, -0.69871163f, -0.67577744f, -0.65216964f, -0.62791175f, -0.60302794f,
        -0.577543f, -0.5514824f, -0.52487206f, -0.49773848f, -0.47010878f, -0.44201043f, -0.4134715f, -0.3845204f, -0.35518602f, -0.32549757f,
        -0.29548463f, -0.2651772f, -0.23460539f, -0.20379974f, -0.17279093f, -0.14160988f, -0.110287674f, -0.07885553f, -0.04734478f, -0.01578684f,
        0.0f, 0.0f, 0.01578684f, 0.04734478f, 0.07885553f, 0.110287674f, 0.14160988f, 0.17279093f, 0.20379974f, 0.23460539f,
        0.2651772f, 0.29548463f, 0.32549757f, 0.35518602f, 0.3845204f, 0.4134715f, 0.44201043f, 0.47010878f, 0.49773848f, 0.52487206f,
        0.5514824f, 0.577543f, 0.60302794f, 0.62791175f, 0.65216964f, 0.67577744f, 0.69871163f, 0.7209493f, 0.7424683f, 0.7632472f,
        0.7832653f, 0.80250263f, 0.82093996f, 0.838559f, 0.85534215f, 0.8712726f, 0.88633466f, 0.9005131f, 0.913794f, 0.9261639f,
        0.9376106f, 0.94812274f, 0.95768976f, 0.9663021f, 0.97395116f, 0.98062944f, 0.98633015f, 0.9910477f, 0.9947773f, 0.99751526f,
        0.9992586f, 1.0f};
    std::vector<float> inputs = linspace(-1.1f, 1.1f, 30, nullptr);
    inputs.push_back((values[0] + values[1]) / 2.0f);
    int inside = 0;
    for (float x : inputs) {
        long il;
        float w[4];
        if (!catmull_rom_weights(values, x, &il, w)) {
            CHECK(x < values.front() || x > values.back(), "None for %g inside the nodes", x);
            continue;
        }
        ++inside;
        const float weight_sum = w[0] + w[1] + w[2] + w[3];
        CHECK(pn_abs(weight_sum - 1.0f) < 1e-6f, "weights sum to %g at %g", weight_sum, x);
        const size_t i0 = (size_t)(il + 1), i1 = (size_t)(il + 2);
        CHECK(x >= values[i0] && x < values[i1], "%g outside [%g, %g)", x, values[i0], values[i1]);
    }
    CHECK(inside >= 27, "only %d of the inputs fell inside the nodes", inside);
}
// geometry/src/fourier.rs:494-508 fourier_sum_test: the Chebyshev sum against a_k cos(k phi) term by term
void fourier_sum_test() {
    uint64_t rng = pn_rng_init(29, 0, 0);
    float a[15];
    for (float& x : a) x = pn_rng_f32(&rng);
    for (int i = 0; i < 680; ++i) {
        const float cos_phi = pn_clamp(pn_rng_f32(&rng) * 2.0f - 1.0f, -1.0f, 1.0f);
        const float phi = pn_acos(cos_phi);
        float expected = 0.0f;
        for (int k = 0; k < 15; ++k) expected += a[k] * pn_cos(phi * (float)k);
        const float actual = fourier_sum(a, 15, cos_phi);
        CHECK(pn_abs(actual - expected) < 2e-5f, "fourier_sum %g against %g at cos_phi %g", actual, expected, cos_phi);
    }
}
// Not in the reference (its Fourier tests need assets/paint.bsdf, which the snapshot does not hold): sample_fourier on a
// series that is a probability density up to scale — the sampled angle's CDF is the random number, and the returned
// density is f(phi) / (2 pi a_0).
void sample_fourier_inverts_the_cdf() {
    const float ak[4] = {1.0f, 0.5f, -0.2f, 0.1f};  // positive everywhere: |0.5| + |0.2| + |0.1| < 1
    float recip[4];
    for (int i = 0; i < 4; ++i) recip[i] = pn_recip((float)i);
    for (int i = 0; i < 200; ++i) {
        const float u = ((float)i + 0.5f) / 200.0f;
        float f, phi, pdf;
        sample_fourier(ak, 4, recip, u, &f, &phi, &pdf);
        const double p = (double)phi;
        const double fold = p <= 3.141592653589793 ? p : 2.0 * 3.141592653589793 - p;
        double F = ak[0] * fold;
        double fv = ak[0];
        for (int k = 1; k < 4; ++k) {
            F += ak[k] / k * std::sin(k * fold);
            fv += ak[k] * std::cos(k * fold);
        }
        const double un = u >= 0.5f ? 1.0 - 2.0 * (u - 0.5) : u * 2.0;
        CHECK(std::fabs(F / (3.141592653589793 * ak[0]) - un) < 2e-6, "u %g: CDF at the sampled angle is %g", u, F / (3.141592653589793 * ak[0]));
        CHECK((u >= 0.5f) == (p > 3.141592653589793), "u %g on the wrong half: phi %g", u, p);
        CHECK(std::fabs(f - fv) < 1e-5 && std::fabs(pdf - fv / (2.0 * 3.141592653589793 * ak[0])) < 1e-6, "u %g: f %g pdf %g", u, f, pdf);
    }
}

struct Entry {
    const char* name;
    void (*fn)();
};
const Entry kTests[] = {
    {"local_trigonometry_test", local_trigonometry_test},
    {"fresnel_test", fresnel_test},
    {"specular_refl_test", specular_refl_test},
    {"diffuse_refl_test", diffuse_refl_test},
    {"play_with_mf_brdf", play_with_mf_brdf},
    {"diff_area_validate", diff_area_validate},
    {"pdf_integral_validate", pdf_integral_validate},
    {"beckmann_rho", beckmann_rho},
    {"quad_frame_test", quad_frame_test},
    {"custom_frame_test", custom_frame_test},
    {"sphere_test", sphere_test},
    {"tricky_triangle", tricky_triangle},
    {"reflect_refract_test", reflect_refract_test},
    {"float_doctests", float_doctests},
    {"bbox_transform_test", bbox_transform_test},
    {"sphere_sample_pdf_integrate", sphere_sample_pdf_integrate},
    {"observe_sphere_sample_towards", observe_sphere_sample_towards},
    {"lambertian_test", lambertian_test},
    {"mf_refl_test", mf_refl_test},
    {"temperature_to_color_test", temperature_to_color_test},
    {"tridiagonal_test", tridiagonal_test},
    {"cubic_spline_test", cubic_spline_test},
    {"find_interval_test", find_interval_test},
    {"catmull_test", catmull_test},
    {"fourier_sum_test", fourier_sum_test},
    {"sample_fourier_inverts_the_cdf", sample_fourier_inverts_the_cdf},
};
const uint32_t kNumTests = sizeof(kTests) / sizeof(kTests[0]);

}  // namespace

extern "C" {
uint32_t oracle_selftest_count(void) { return kNumTests; }
const char* oracle_selftest_name(uint32_t i) { return i < kNumTests ? kTests[i].name : nullptr; }
int oracle_selftest(const char* name, char* log, uint32_t log_cap) {
    Log l;
    g_log = &l;
    for (uint32_t i = 0; i < kNumTests; ++i)
        if (!name || std::strcmp(name, kTests[i].name) == 0) kTests[i].fn();
    g_log = nullptr;
    if (log && log_cap) {
        std::strncpy(log, l.text.c_str(), log_cap - 1);
        log[log_cap - 1] = 0;
    }
    return l.failures;
}
}
