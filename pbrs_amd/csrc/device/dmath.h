// device/dmath.h — f32 vector vocabulary of the HIP kernels (gfx950).
//
// Mirrors the arithmetic of math/src/hcm.rs, radiometry/src/color.rs and geometry/src/bvh.rs of the
// reference operand for operand: the kernels must reproduce the CPU path integrator bit for bit at
// matched RNG seeds (BASELINE.json north_star), so no expression here may be re-associated, fused
// (-ffp-contract=off) or replaced by a reciprocal multiply.
#pragma once
#include <hip/hip_runtime.h>

#include "../../../include/pbrs_numeric.h"

#define PD __device__ __forceinline__

struct f3 {
    float x, y, z;
};
PD f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
PD f3 ld3(const float* p) { return f3{p[0], p[1], p[2]}; }
PD float comp(f3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
PD void setc(f3& v, int i, float s) {
    if (i == 0) v.x = s;
    else if (i == 1) v.y = s;
    else v.z = s;
}
PD f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }  // hcm.rs:170-175
PD f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }  // :193-198
PD f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }                       // :199-204
PD f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }     // :227-232
PD f3 operator*(float s, f3 a) { return {a.x * s, a.y * s, a.z * s}; }     // :233-238 (v * self)
PD f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }     // :239-244
PD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }     // :86-88
PD f3 cross(f3 a, f3 v) { return {a.y * v.z - a.z * v.y, a.z * v.x - a.x * v.z, a.x * v.y - a.y * v.x}; }  // :89-98
PD float norm2(f3 a) { return dot(a, a); }
PD float norm(f3 a) { return pn_sqrt(dot(a, a)); }
PD f3 hat(f3 a) {  // :112-117 (the assert on zero / non-finite length is a counted panic site upstream)
    float inv_sqrt = 1.0f / norm(a);
    return a * inv_sqrt;
}
PD bool try_hat(f3 a, f3& out) {  // :118-121
    float inv_length = 1.0f / norm(a);
    if (pn_isfinite(inv_length) && inv_length != 0.0f) {
        out = inv_length * a;
        return true;
    }
    return false;
}
PD f3 facing(f3 self, f3 normal) { return pn_sign_negative(dot(self, normal)) ? self : -self; }  // :124-130
PD f3 projected_onto(f3 self, f3 other) { return dot(self, other) * other / norm2(other); }      // :144-146
PD bool has_nan3(f3 a) { return a.x != a.x || a.y != a.y || a.z != a.z; }
PD int abs_min_dimension(f3 a) {  // :149-154
    float a0 = pn_abs(a.x), a1 = pn_abs(a.y), a2 = pn_abs(a.z);
    int res = a0 < a1 ? 0 : 1;
    float ar = res == 0 ? a0 : a1;
    return ar < a2 ? res : 2;
}
PD void make_coord_system(f3 v, f3& o1, f3& o2) {  // :595-605
    int i0 = abs_min_dimension(v);
    int i1 = (i0 + 1) % 3, i2 = (i0 + 2) % 3;
    f3 v1 = mk3(0.0f, 0.0f, 0.0f);
    setc(v1, i1, comp(v, i2));
    setc(v1, i2, -comp(v, i1));
    f3 v2 = cross(v, v1);
    o1 = hat(v1);
    o2 = hat(v2);
}
PD f3 reflect3(f3 normal, f3 wi) {  // :607-611
    f3 perp = dot(wi, normal) * normal / norm2(normal);
    f3 parallel = wi - perp;
    return wi - 2.0f * parallel;
}
PD bool refract3(f3 normal, f3 wi, float ni_over_no, f3& out) {  // :625-640 (true = Transmit)
    wi = hat(wi);
    normal = hat(normal);
    float cos_theta_i = dot(wi, normal);
    float sin2_theta_i = pn_max(1.0f - pn_sq(cos_theta_i), 0.0f);
    float sin2_theta_o = sin2_theta_i * pn_sq(ni_over_no);
    if (sin2_theta_o >= 1.0f) {
        out = reflect3(normal, wi);
        return false;
    }
    float cos_theta_o = pn_sqrt(1.0f - sin2_theta_o);
    out = ni_over_no * -wi + (ni_over_no * cos_theta_i - cos_theta_o) * normal;
    return true;
}
PD f3 spherical_direction(float sin_theta, float cos_theta, float phi) {  // :647-650, Q3 (names swapped)
    float s, c;
    pn_sincos(phi, &s, &c);
    return mk3(sin_theta * s, sin_theta * c, cos_theta);
}
PD f3 bary_lerp(f3 a, f3 b, f3 c, float bc0, float bc1) { return (a - c) * bc0 + (b - c) * bc1 + c; }  // float.rs:37-50

// Mat3 (columns) * Vec3, hcm.rs:448-453
PD f3 mat3_mul(f3 c0, f3 c1, f3 c2, f3 v) { return c0 * v.x + c1 * v.y + c2 * v.z; }

// radiometry/src/color.rs — Color shares f3 (r,g,b) = (x,y,z)
PD f3 cmul(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }   // :157-162
PD f3 cdiv(f3 a, f3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }   // cw_div :97-99
PD f3 csqrt(f3 a) { return {pn_sqrt(a.x), pn_sqrt(a.y), pn_sqrt(a.z)}; }
PD f3 cmaxs(f3 a, float s) { return {pn_max(a.x, s), pn_max(a.y, s), pn_max(a.z, s)}; }
PD f3 gray(float l) { return {l, l, l}; }
PD bool is_black(f3 c) { return c.x <= 0.0f && c.y <= 0.0f && c.z <= 0.0f; }  // :57-59
PD float luminance(f3 c) { return 0.21267127f * c.x + 0.71515972f * c.y + 0.07216883f * c.z; }  // :116-118,:222-228

// geometry/src/ray.rs:40-46
PD bool truncated_t(float t, float t_max) { return !(t < PN_EPSILON || t >= t_max); }

// glam Vec3A min/max = SSE minps/maxps (second operand on NaN), geometry/src/bvh.rs:84-99
PD float sse_min(float a, float b) { return a < b ? a : b; }
PD float sse_max(float a, float b) { return a > b ? a : b; }
PD bool slab_test(f3 bmin, f3 bmax, f3 o, f3 d, float t_max) {
    float t0x = (bmin.x - o.x) / d.x, t0y = (bmin.y - o.y) / d.y, t0z = (bmin.z - o.z) / d.z;
    float t1x = (bmax.x - o.x) / d.x, t1y = (bmax.y - o.y) / d.y, t1z = (bmax.z - o.z) / d.z;
    float lox = sse_min(t0x, t1x), loy = sse_min(t0y, t1y), loz = sse_min(t0z, t1z);
    float hix = sse_max(t0x, t1x), hiy = sse_max(t0y, t1y), hiz = sse_max(t0z, t1z);
    float lo_el = sse_max(sse_max(lox, loz), sse_max(loy, loz));  // Vec3A::max_element
    float hi_el = sse_min(sse_min(hix, hiz), sse_min(hiy, hiz));  // Vec3A::min_element
    float t_low = pn_max(lo_el, 0.0f);
    float t_high = pn_min(hi_el, t_max);
    return t_low <= t_high;
}

// Affine rows (pbrs_instance.inv / .fwd): Mat4 * (v, w) with glam's left-to-right column sum,
// hcm.rs:539-544; the w = 0 / w = 1 products are kept because x*0.0 and x*1.0 shape NaN/-0 cases.
PD f3 xf_apply(const float (*rows)[4], f3 v, float w) {
    return mk3(rows[0][0] * v.x + rows[0][1] * v.y + rows[0][2] * v.z + rows[0][3] * w,
               rows[1][0] * v.x + rows[1][1] * v.y + rows[1][2] * v.z + rows[1][3] * w,
               rows[2][0] * v.x + rows[2][1] * v.y + rows[2][2] * v.z + rows[2][3] * w);
}
// inverse.transpose() * n (Mat4 * Vec3: three columns only), transform.rs:314, hcm.rs:558-564
PD f3 xf_normal(const float (*inv_rows)[4], f3 n) {
    return mk3(inv_rows[0][0] * n.x + inv_rows[1][0] * n.y + inv_rows[2][0] * n.z,
               inv_rows[0][1] * n.x + inv_rows[1][1] * n.y + inv_rows[2][1] * n.z,
               inv_rows[0][2] * n.x + inv_rows[1][2] * n.y + inv_rows[2][2] * n.z);
}
