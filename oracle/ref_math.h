// oracle/ref_math.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the reference's numeric vocabulary: math/src/hcm.rs (Vec3/Point3/Mat3/Mat4,
// reflect/refract/make_coord_system/spherical_direction), math/src/float.rs (barycentric_lerp,
// weak_recip, try_divide), math/src/prob.rs (Prob), radiometry/src/color.rs (Color, luminance),
// geometry/src/ray.rs (Ray), geometry/src/bvh.rs (BBox).  Expression order follows the Rust
// source token for token because f32 results must be reproducible bit for bit.
// Compile with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstdint>

#include "../include/pbrs_numeric.h"

namespace ref {

// Reference `assert!`/`panic!` sites that a release build of pbrs would abort on.  The oracle
// counts them instead of aborting so a test can assert the count is zero on a scene.
struct Diag {
    uint64_t panics = 0;         // would-have-panicked conditions reached
    uint64_t tlas_ties = 0;      // tlas/src/bvh.rs:94 reached with l.t == r.t
    uint64_t sphere_inside = 0;  // D4: Interaction::new assert skipped for interior sphere hits
    uint64_t nonfinite_samples = 0;  // camera samples whose radiance is not finite
};
extern thread_local Diag* g_diag;
inline void ref_panic() {
    if (g_diag) g_diag->panics++;
}
#define REF_ASSERT(c) \
    do {              \
        if (!(c)) ::ref::ref_panic(); \
    } while (0)

// ---- math/src/hcm.rs:23-34 ----------------------------------------------------------------
struct Vec3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
using Point3 = Vec3;  // the reference keeps two types for type safety only; arithmetic is identical

inline Vec3 v3(float x, float y, float z) { return Vec3{x, y, z}; }
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }  // hcm.rs:170-175
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }  // :193-198
inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }                         // :199-204
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }       // :227-232
inline Vec3 operator*(float s, Vec3 a) { return a * s; }                             // :233-238
inline Vec3 operator/(Vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }       // :239-244
inline float dot(Vec3 a, Vec3 v) { return a.x * v.x + a.y * v.y + a.z * v.z; }       // :86-88
inline Vec3 cross(Vec3 a, Vec3 v) {                                                  // :89-98
    return {a.y * v.z - a.z * v.y, a.z * v.x - a.x * v.z, a.x * v.y - a.y * v.x};
}
inline float norm_squared(Vec3 a) { return dot(a, a); }          // :100-102
inline float norm(Vec3 a) { return pn_sqrt(norm_squared(a)); }   // :103-105
inline bool has_nan(Vec3 a) { return a.x != a.x || a.y != a.y || a.z != a.z; }
inline Vec3 hat(Vec3 a) {  // :112-117
    float n2 = norm_squared(a);
    REF_ASSERT(n2 != 0.0f && pn_isfinite(n2));
    float inv_sqrt = 1.0f / norm(a);
    return a * inv_sqrt;
}
inline bool try_hat(Vec3 a, Vec3* out) {  // :118-121
    float inv_length = 1.0f / norm(a);
    if (pn_isfinite(inv_length) && inv_length != 0.0f) {
        *out = inv_length * a;
        return true;
    }
    return false;
}
inline Vec3 facing(Vec3 self, Vec3 normal) {  // :124-130
    return pn_sign_negative(dot(self, normal)) ? self : -self;
}
inline Vec3 projected_onto(Vec3 self, Vec3 other) {  // :144-146
    return dot(self, other) * other / norm_squared(other);
}
inline int abs_min_dimension(Vec3 a) {  // :149-154
    float ab[3] = {pn_abs(a.x), pn_abs(a.y), pn_abs(a.z)};
    int res = ab[0] < ab[1] ? 0 : 1;
    res = ab[res] < ab[2] ? res : 2;
    return res;
}
inline int max_dimension(Vec3 a) {  // :156-163
    int res = a.x > a.y ? 0 : 1;
    return a[2] > a[res] ? 2 : res;
}
inline float distance_to(Point3 a, Point3 p) { return norm(a - p); }             // :265-267
inline float squared_distance_to(Point3 a, Point3 p) { return norm_squared(a - p); }

// ---- Mat3 (hcm.rs:357-471) -----------------------------------------------------------------
struct Mat3 {
    Vec3 cols[3];
};
inline Mat3 mat3_cols(Vec3 a, Vec3 b, Vec3 c) { return Mat3{{a, b, c}}; }
inline Vec3 operator*(const Mat3& m, Vec3 v) {  // :448-453
    return m.cols[0] * v[0] + m.cols[1] * v[1] + m.cols[2] * v[2];
}

// ---- Mat4 with glam Vec4 columns (hcm.rs:477-576) ----------------------------------------
struct Vec4 {
    float x, y, z, w;
};
inline Vec4 operator*(Vec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline Vec4 operator+(Vec4 a, Vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
struct Mat4 {
    Vec4 cols[4];
};
inline Vec4 mul(const Mat4& m, Vec4 v) {  // :539-544
    return m.cols[0] * v.x + m.cols[1] * v.y + m.cols[2] * v.z + m.cols[3] * v.w;
}
inline Vec3 mul_vec3(const Mat4& m, Vec3 v) {  // :558-564 (three columns only)
    Vec4 v4 = m.cols[0] * v[0] + m.cols[1] * v[1] + m.cols[2] * v[2];
    return {v4.x, v4.y, v4.z};
}
inline Mat4 transpose(const Mat4& m) {  // :521-529
    const float* a = &m.cols[0].x;
    Mat4 r;
    float* b = &r.cols[0].x;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) b[4 * i + j] = a[4 * j + i];
    return r;
}

// ---- mod-level functions (hcm.rs:595-650) ------------------------------------------------
inline void make_coord_system(Vec3 v, Vec3* o1, Vec3* o2) {  // :595-605
    int i0 = abs_min_dimension(v);
    int i1 = (i0 + 1) % 3, i2 = (i0 + 2) % 3;
    Vec3 v1{0.0f, 0.0f, 0.0f};
    v1.at(i1) = v[i2];
    v1.at(i2) = -v[i1];
    REF_ASSERT(pn_abs(dot(v1, v)) < PN_EPSILON);
    Vec3 v2 = cross(v, v1);
    *o1 = hat(v1);
    *o2 = hat(v2);
}
inline Vec3 reflect(Vec3 normal, Vec3 wi) {  // :607-611
    Vec3 perp = dot(wi, normal) * normal / norm_squared(normal);
    Vec3 parallel = wi - perp;
    return wi - 2.0f * parallel;
}
// :625-640 — returns true for Transmit, false for FullReflect
inline bool refract(Vec3 normal, Vec3 wi, float ni_over_no, Vec3* out) {
    wi = hat(wi);
    normal = hat(normal);
    float cos_theta_i = dot(wi, normal);
    REF_ASSERT(cos_theta_i >= 0.0f);
    float sin2_theta_i = pn_max(1.0f - pn_sq(cos_theta_i), 0.0f);
    float sin2_theta_o = sin2_theta_i * pn_sq(ni_over_no);
    if (sin2_theta_o >= 1.0f) {
        *out = reflect(normal, wi);
        return false;
    }
    float cos_theta_o = pn_sqrt(1.0f - sin2_theta_o);
    *out = ni_over_no * -wi + (ni_over_no * cos_theta_i - cos_theta_o) * normal;
    return true;
}
// :647-650 — Q3: `let (cos_phi, sin_phi) = phi.sin_cos()` binds sin to cos_phi and cos to sin_phi.
inline Vec3 spherical_direction(float sin_theta, float cos_theta, float phi) {
    float s, c;
    pn_sincos(phi, &s, &c);
    float cos_phi = s, sin_phi = c;
    return {sin_theta * cos_phi, sin_theta * sin_phi, cos_theta};
}

// ---- math/src/float.rs:37-50 ------------------------------------------------------------
inline Vec3 barycentric_lerp(Vec3 a, Vec3 b, Vec3 c, float bc0, float bc1) {
    return (a - c) * bc0 + (b - c) * bc1 + c;
}
inline float barycentric_lerp(float a, float b, float c, float bc0, float bc1) {
    return (a - c) * bc0 + (b - c) * bc1 + c;
}

// ---- math/src/prob.rs:5-42 -----------------------------------------------------------------
struct Prob {
    bool is_mass;
    float v;
    static Prob Density(float x) { return {false, x}; }
    static Prob Mass(float x) { return {true, x}; }
    bool is_density() const { return !is_mass; }
    bool is_positive() const { return v > 0.0f; }
    float density() const { return is_mass ? 0.0f : v; }
    float mass() const { return is_mass ? v : 0.0f; }
    bool is_zero() const { return v == 0.0f; }
};

// ---- radiometry/src/color.rs ------------------------------------------------------------
struct Color {
    float r, g, b;
};
inline Color rgb(float r, float g, float b) { return Color{r, g, b}; }
inline Color gray(float l) { return Color{l, l, l}; }
inline Color black() { return Color{0.0f, 0.0f, 0.0f}; }
inline Color operator+(Color a, Color b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }  // :121-126
inline Color operator-(Color a, Color b) { return {a.r - b.r, a.g - b.g, a.b - b.b}; }  // :136-141
inline Color operator*(Color a, float s) { return {a.r * s, a.g * s, a.b * s}; }        // :143-148
inline Color operator*(float s, Color a) { return a * s; }                              // :150-155
inline Color operator*(Color a, Color b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }  // :157-162
inline bool is_black(Color c) { return c.r <= 0.0f && c.g <= 0.0f && c.b <= 0.0f; }    // :57-59
inline bool is_finite(Color c) { return pn_isfinite(c.r) && pn_isfinite(c.g) && pn_isfinite(c.b); }
inline Color cw_div(Color a, Color b) { return {a.r / b.r, a.g / b.g, a.b / b.b}; }     // :97-99
inline Color cw_sqrt(Color a) { return {pn_sqrt(a.r), pn_sqrt(a.g), pn_sqrt(a.b)}; }
inline Color cw_max(Color a, float x) { return {pn_max(a.r, x), pn_max(a.g, x), pn_max(a.b, x)}; }
inline float luminance(Color c) {  // :116-118, :222-228 (XYZ.y)
    return 0.21267127f * c.r + 0.71515972f * c.g + 0.07216883f * c.b;
}

// ---- geometry/src/ray.rs:17-51 ------------------------------------------------------------
struct Ray {
    Point3 origin;
    Vec3 dir;
    float t_max;
};
inline Ray ray_new(Point3 o, Vec3 d) { return Ray{o, d, pn_inf()}; }
inline bool truncated_t(const Ray& r, float t, float* out) {  // :40-46
    if (t < PN_EPSILON || t >= r.t_max) return false;
    *out = t;
    return true;
}
inline Point3 position_at(const Ray& r, float t) { return r.origin + t * r.dir; }  // :48-50

// ---- geometry/src/bvh.rs (BBox over glam::Vec3A) ------------------------------------------
// glam 0.20 is not vendored in the reference.  On x86-64 Vec3A::min/max are SSE minps/maxps,
// whose NaN rule is "return the second operand"; max_element/min_element are two shuffles +
// maxps/minps.  Restated from glam's published SSE2 implementation; parity unpinned (no reference
// test exercises NaN slabs).
inline float sse_min(float a, float b) { return a < b ? a : b; }
inline float sse_max(float a, float b) { return a > b ? a : b; }
struct BBox {
    Vec3 min, max;
};
inline BBox bbox_empty() { return {{pn_inf(), pn_inf(), pn_inf()}, {-pn_inf(), -pn_inf(), -pn_inf()}}; }
inline BBox bbox_new(Point3 p0, Point3 p1) {  // :26-33
    return {{sse_min(p0.x, p1.x), sse_min(p0.y, p1.y), sse_min(p0.z, p1.z)},
            {sse_max(p0.x, p1.x), sse_max(p0.y, p1.y), sse_max(p0.z, p1.z)}};
}
inline BBox bbox_union_pt(BBox b, Point3 p) {  // :35-42 (f32::min / f32::max per component)
    BBox r = b;
    for (int i = 0; i < 3; ++i) {
        r.min.at(i) = pn_min(b.min[i], p[i]);
        r.max.at(i) = pn_max(b.max[i], p[i]);
    }
    return r;
}
inline BBox bbox_union(BBox a, BBox b) {  // :138-143
    return {{sse_min(a.min.x, b.min.x), sse_min(a.min.y, b.min.y), sse_min(a.min.z, b.min.z)},
            {sse_max(a.max.x, b.max.x), sse_max(a.max.y, b.max.y), sse_max(a.max.z, b.max.z)}};
}
inline Point3 bbox_midpoint(const BBox& b) {  // :44-47
    return (b.max - b.min) * 0.5f + b.min;
}
inline Vec3 bbox_diag(const BBox& b) { return b.max - b.min; }  // :49-52
inline float bbox_area(const BBox& b) {                         // :75-82
    Vec3 d = bbox_diag(b);
    if (!pn_sign_negative(d.x) && !pn_sign_negative(d.y) && !pn_sign_negative(d.z))
        return (d.x * d.y + d.y * d.z + d.z * d.x) * 2.0f;
    return 0.0f;
}
inline bool bbox_intersect(const BBox& b, const Ray& r) {  // :84-99
    Vec3 t0 = {(b.min.x - r.origin.x) / r.dir.x, (b.min.y - r.origin.y) / r.dir.y, (b.min.z - r.origin.z) / r.dir.z};
    Vec3 t1 = {(b.max.x - r.origin.x) / r.dir.x, (b.max.y - r.origin.y) / r.dir.y, (b.max.z - r.origin.z) / r.dir.z};
    Vec3 lo = {sse_min(t0.x, t1.x), sse_min(t0.y, t1.y), sse_min(t0.z, t1.z)};
    Vec3 hi = {sse_max(t0.x, t1.x), sse_max(t0.y, t1.y), sse_max(t0.z, t1.z)};
    float lo_el = sse_max(sse_max(lo.x, lo.z), sse_max(lo.y, lo.z));  // Vec3A::max_element
    float hi_el = sse_min(sse_min(hi.x, hi.z), sse_min(hi.y, hi.z));  // Vec3A::min_element
    float t_low = pn_max(lo_el, 0.0f);
    float t_high = pn_min(hi_el, r.t_max);
    return t_low <= t_high;
}

}  // namespace ref
