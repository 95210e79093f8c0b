// pbrs_amd/csrc/host/flatten.cpp — host side of the pbrs_gpu boundary (see include/pbrs_host.h).
//
// Builds the acceleration structures exactly as the reference's host code does and linearises
// them for HBM:
//   * TLAS: `build_bvh` spatial-median split over instance bbox midpoints (tlas/src/bvh.rs:116-152),
//     emitted in pre-order so that "left then right" (tlas/src/bvh.rs:84-88) is "i+1 then a".
//   * BLAS: `recursive_build` (shape/src/blas.rs:333-420): leaf <= 4, widest centroid axis, boxes
//     sorted by centroid, pivot where the prefix bbox area reaches half, in-place partition.
//   * triangles re-gathered in leaf order with the (i,k,j) read of shape/src/blas.rs:162 baked in.
//   * `Material::bxdfs_at` evaluated once per material (all textures are Solid in this tier).
// Tree shape is semantic (it fixes traversal order, tie-breaks and byte counts), so every f32
// expression below keeps the reference's operand order; compile with -ffp-contract=off.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/pbrs_host.h"
#include "../../../include/pbrs_numeric.h"

namespace {

thread_local std::string g_error;

struct V3 {
    float x, y, z;
    float get(int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    void set(int i, float v) { (i == 0 ? x : (i == 1 ? y : z)) = v; }
};
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 scale(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross3(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(V3 a) { return pn_sqrt(dot3(a, a)); }
inline V3 normalized(V3 a) { return scale(a, 1.0f / length(a)); }  // Vec3::hat, math/src/hcm.rs:112-117

// PBRS_MESH_SMOOTH_SHADING_OK for one triangle (blas.rs:170-200).  A hit's shading normal is
// n = +-hat(N), N = w0 n0 + w1 n1 + w2 n2 with weights in [-3e-7, 1] that sum to 1 (the normalised same-sign
// barycentrics of simple.rs:471-480, re-derived as 1 - b1 - b2), or the geometric normal if hat fails.  The
// reference then forms r = u - n (u.n) / |n|^2 (u = dpdu_raw), dpdu = hat(r), and drops the hit when |dpdu . n| >= 1e-3.
// In exact arithmetic r is orthogonal to n; in f32 the five roundings of the projection leave |r . n| <= 12 eps |u|
// while |r| >= |u| (sin(theta) - 12 eps), theta = angle(u, n), so |dpdu . n| <= 12 eps / (sin(theta) - 12 eps) + 4 eps
// (eps = 2^-24).  This function proves sin(theta) >= 0.0447 (|cos| <= 0.999) for every admissible n:
//   |u . N| <= |u| (1 + 1e-6) max_i |u^ . n_i|   and   |N|^2 >= g (1 - 1e-5),  g = min_{i,j} n_i . n_j > 0,
// hence |cos(theta)| <= c_max / sqrt(g) * (1 + 1e-5); it asks for c_max <= 0.99 sqrt(g), which bounds the dropped
// quantity by 2e-5, fifty times under the threshold.  With g > 0 and sane lengths hat(N) cannot fail, and the
// geometric normal is checked as well.  Evaluated in double: only the inequality matters, not its rounding.
bool smooth_shading_bound(V3 u, V3 n0, V3 n1, V3 n2, V3 gn) {
    auto d3 = [](V3 a, V3 b) { return (double)a.x * b.x + (double)a.y * b.y + (double)a.z * b.z; };
    const double L = std::sqrt(d3(u, u));
    if (!(L >= 1e-10 && L <= 1e15)) return false;
    const V3 n[3] = {n0, n1, n2};
    if (!(std::fabs(d3(u, gn)) / L <= 0.99)) return false;  // the fallback normal (unit length)
    double g = 1e300, lmax = 0.0, cmax = 0.0;
    for (int i = 0; i < 3; ++i) {
        cmax = std::max(cmax, std::fabs(d3(u, n[i])) / L);
        for (int j = i; j < 3; ++j) g = std::min(g, d3(n[i], n[j]));
        lmax = std::max(lmax, std::sqrt(d3(n[i], n[i])));
    }
    if (!(g > 0.0) || !(lmax <= 1e15) || !(std::sqrt(g) >= 1e-10) || !(lmax <= 1e3 * std::sqrt(g))) return false;
    return cmax <= 0.99 * std::sqrt(g);
}
inline V3 v3p(const float* p) { return {p[0], p[1], p[2]}; }

// geometry/src/bvh.rs — glam Vec3A min/max are SSE minps/maxps ("second operand on NaN").
inline float mn_sse(float a, float b) { return a < b ? a : b; }
inline float mx_sse(float a, float b) { return a > b ? a : b; }
struct Box {
    V3 lo, hi;
};
inline Box box_empty() { return {{pn_inf(), pn_inf(), pn_inf()}, {-pn_inf(), -pn_inf(), -pn_inf()}}; }
inline Box box_of(V3 p0, V3 p1) {  // BBox::new :26-33
    return {{mn_sse(p0.x, p1.x), mn_sse(p0.y, p1.y), mn_sse(p0.z, p1.z)}, {mx_sse(p0.x, p1.x), mx_sse(p0.y, p1.y), mx_sse(p0.z, p1.z)}};
}
inline Box box_grow(Box b, V3 p) {  // BBox::union(Point3) :35-42 — f32::min / f32::max
    return {{pn_min(b.lo.x, p.x), pn_min(b.lo.y, p.y), pn_min(b.lo.z, p.z)}, {pn_max(b.hi.x, p.x), pn_max(b.hi.y, p.y), pn_max(b.hi.z, p.z)}};
}
inline Box box_merge(Box a, Box b) {  // bvh::union :138-143
    return {{mn_sse(a.lo.x, b.lo.x), mn_sse(a.lo.y, b.lo.y), mn_sse(a.lo.z, b.lo.z)},
            {mx_sse(a.hi.x, b.hi.x), mx_sse(a.hi.y, b.hi.y), mx_sse(a.hi.z, b.hi.z)}};
}
inline V3 box_mid(const Box& b) { return add(scale(sub(b.hi, b.lo), 0.5f), b.lo); }  // :44-47
inline V3 box_diag(const Box& b) { return sub(b.hi, b.lo); }                         // :49-52
inline float box_area(const Box& b) {                                                // :75-82
    V3 d = box_diag(b);
    if (!pn_sign_negative(d.x) && !pn_sign_negative(d.y) && !pn_sign_negative(d.z)) return (d.x * d.y + d.y * d.z + d.z * d.x) * 2.0f;
    return 0.0f;
}
inline int widest_axis(V3 d) {  // Vec3::max_dimension, math/src/hcm.rs:156-163
    int res = d.x > d.y ? 0 : 1;
    return d.z > d.get(res) ? 2 : res;
}
inline int narrowest_abs_axis(V3 a) {  // Vec3::abs_min_dimension, math/src/hcm.rs:149-154
    float ab[3] = {pn_abs(a.x), pn_abs(a.y), pn_abs(a.z)};
    int res = ab[0] < ab[1] ? 0 : 1;
    return ab[res] < ab[2] ? res : 2;
}
inline void coord_system(V3 v, V3* o1, V3* o2) {  // make_coord_system, math/src/hcm.rs:595-605
    int i0 = narrowest_abs_axis(v), i1 = (i0 + 1) % 3, i2 = (i0 + 2) % 3;
    V3 v1{0, 0, 0};
    v1.set(i1, v.get(i2));
    v1.set(i2, -v.get(i1));
    V3 v2 = cross3(v, v1);
    *o1 = normalized(v1);
    *o2 = normalized(v2);
}
inline void put_node(pbrs_node& n, const Box& b) {
    n.min[0] = b.lo.x; n.min[1] = b.lo.y; n.min[2] = b.lo.z;
    n.max[0] = b.hi.x; n.max[1] = b.hi.y; n.max[2] = b.hi.z;
}
inline Box node_box(const pbrs_node& n) { return {{n.min[0], n.min[1], n.min[2]}, {n.max[0], n.max[1], n.max[2]}}; }

// ---- BLAS ------------------------------------------------------------------------------------------------
struct Tri {
    uint32_t i, j, k;  // index_triple as given to from_soa (blas.rs:72-76)
    Box box;
    uint32_t orig;
};

// The `partition` crate (0.1.2, shape/Cargo.toml:7) is not vendored with the reference: restated as
// the in-place two-pointer scheme its documentation describes; intra-side order is unpinned (Q12).
template <class P>
size_t hoare_partition(Tri* data, size_t len, P pred) {
    if (len == 0) return 0;
    size_t l = 0, r = len - 1;
    for (;;) {
        while (l < len && pred(data[l])) ++l;
        while (r > 0 && !pred(data[r])) --r;
        if (l >= r) return l;
        std::swap(data[l], data[r]);
    }
}

struct BlasBuilder {
    std::vector<Tri>& tris;
    std::vector<pbrs_node>& nodes;
    uint32_t tri_base;  // offset of this mesh's triangles in the global arrays

    // shape/src/blas.rs:333-420; returns (node index, subtree height)
    std::pair<uint32_t, uint32_t> build(size_t start, size_t end) {
        uint32_t idx = (uint32_t)nodes.size();
        nodes.push_back(pbrs_node{});
        size_t len = end - start;
        auto make_leaf = [&](const Box& b) {
            put_node(nodes[idx], b);
            nodes[idx].a = tri_base + (uint32_t)start;
            nodes[idx].b = PBRS_LEAF_FLAG | (uint32_t)len;
            return std::make_pair(idx, 1u);
        };
        if (len <= 4) {
            Box b = box_empty();
            for (size_t s = start; s < end; ++s) b = box_merge(b, tris[s].box);
            return make_leaf(b);
        }
        std::vector<Box> boxes(len);
        for (size_t s = 0; s < len; ++s) boxes[s] = tris[start + s].box;
        Box centroid_box = box_empty();
        for (const Box& b : boxes) centroid_box = box_grow(centroid_box, box_mid(b));
        int axis = widest_axis(box_diag(centroid_box));
        if (box_diag(centroid_box).get(axis) < 1e-8f) {  // "tiny leaf", :354-360
            Box b = box_empty();
            for (const Box& bb : boxes) b = box_merge(b, bb);
            return make_leaf(b);
        }
        std::stable_sort(boxes.begin(), boxes.end(), [axis](const Box& b0, const Box& b1) { return box_mid(b0).get(axis) < box_mid(b1).get(axis); });
        float area_sum = 0.0f;
        for (const Box& b : boxes) area_sum += box_area(b);
        float half_area = area_sum * 0.5f;
        float running = 0.0f;
        size_t split_index = 0;
        for (size_t s = 0; s < len; ++s) {
            running += box_area(boxes[s]);
            if (running >= half_area) {
                split_index = s;
                break;
            }
        }
        float pivot = box_mid(boxes[split_index]).get(axis);
        size_t left_len = hoare_partition(&tris[start], len, [axis, pivot](const Tri& t) { return box_mid(t.box).get(axis) <= pivot; });
        size_t mid = start + left_len;
        if (left_len == 0 || left_len == len) {
            // :403-410 `select_nth_unstable_by(len/2)`: order unspecified upstream; a stable sort by the
            // same key split at len/2 is this build's documented rule (Q12).
            std::stable_sort(tris.begin() + start, tris.begin() + end, [axis](const Tri& a, const Tri& b) { return box_mid(a.box).get(axis) < box_mid(b.box).get(axis); });
            mid = start + len / 2;
        }
        boxes.clear();
        boxes.shrink_to_fit();
        auto l = build(start, mid);
        auto r = build(mid, end);
        put_node(nodes[idx], box_merge(node_box(nodes[l.first]), node_box(nodes[r.first])));
        nodes[idx].a = r.first;
        nodes[idx].b = (uint32_t)axis;
        return std::make_pair(idx, std::max(l.second, r.second) + 1);
    }
};

// ---- geometry/src/transform.rs:273-308 on column-major Mat4 ------------------------------------------------
struct M4 {
    float m[16];  // m[4*col + row]
};
inline V3 xf_point(const M4& t, V3 p) {  // forward * (p,1), sums left to right
    float r[3];
    for (int i = 0; i < 3; ++i) r[i] = t.m[i] * p.x + t.m[4 + i] * p.y + t.m[8 + i] * p.z + t.m[12 + i] * 1.0f;
    return {r[0], r[1], r[2]};
}
inline Box xf_box(const M4& fwd, const Box& b) {  // :287-308
    V3 bases[3] = {{fwd.m[0], fwd.m[1], fwd.m[2]}, {fwd.m[4], fwd.m[5], fwd.m[6]}, {fwd.m[8], fwd.m[9], fwd.m[10]}};
    Box res = box_empty();
    V3 diag = box_diag(b);
    for (int i = 0; i < 8; ++i) {
        V3 corner = xf_point(fwd, b.lo);
        if (i & 1) corner = add(corner, scale(bases[0], diag.x));
        if (i & 2) corner = add(corner, scale(bases[1], diag.y));
        if (i & 4) corner = add(corner, scale(bases[2], diag.z));
        res = box_grow(res, corner);
    }
    return res;
}

// ---- shape bboxes (shape/src/simple.rs) ------------------------------------------------------------------------
Box shape_box(const pbrs_shape_spec& s, const std::vector<pbrs_mesh>& meshes, const std::vector<pbrs_node>& blas_nodes) {
    const float* p = s.p;
    switch (s.kind) {
        case PBRS_SHAPE_SPHERE: {  // :203-206
            V3 c = v3p(p);
            V3 hd = scale(V3{1.0f, 1.0f, 1.0f}, p[3]);
            return box_of(sub(c, hd), add(c, hd));
        }
        case PBRS_SHAPE_QUAD: {  // :106-113
            V3 o = v3p(p), u = v3p(p + 3), v = v3p(p + 6);
            return box_merge(box_of(o, add(o, u)), box_of(add(o, v), add(add(o, u), v)));
        }
        case PBRS_SHAPE_CUBOID: return box_of(v3p(p), v3p(p + 3));  // :339-341 (already ordered)
        case PBRS_SHAPE_DISK: {                                      // :298-305
            V3 c = v3p(p), n = normalized(v3p(p + 3)), radial = v3p(p + 6);
            V3 v1, v2;
            coord_system(n, &v1, &v2);
            float rn = length(radial);
            v1 = scale(v1, rn);
            v2 = scale(v2, rn);
            return box_merge(box_of(add(add(c, v1), v2), sub(add(c, v1), v2)), box_of(sub(sub(c, v1), v2), add(sub(c, v1), v2)));
        }
        case PBRS_SHAPE_TRIANGLE: return box_grow(box_of(v3p(p), v3p(p + 3)), v3p(p + 6));  // :422-424
        default: return node_box(blas_nodes[meshes[s.mesh].root]);                          // blas.rs:313-321
    }
}

// ---- TLAS (tlas/src/bvh.rs:116-152) -----------------------------------------------------------------------------
struct TlasBuilder {
    const std::vector<Box>& inst_box;
    const std::vector<pbrs_instance>& inst;
    std::vector<pbrs_node>& nodes;
    std::pair<uint32_t, uint32_t> build(std::vector<uint32_t> insts) {
        uint32_t idx = (uint32_t)nodes.size();
        nodes.push_back(pbrs_node{});
        if (insts.size() == 1) {
            put_node(nodes[idx], inst_box[insts[0]]);  // BvhNode::new_leaf :44-49
            nodes[idx].a = insts[0];
            nodes[idx].b = PBRS_LEAF_FLAG | (inst[insts[0]].shape_kind << PBRS_TLAS_LEAF_KIND_SHIFT) | 1u;
            return {idx, 1u};
        }
        size_t num_all = insts.size();
        Box all = box_empty();
        for (uint32_t i : insts) all = box_merge(all, inst_box[i]);
        int axis = widest_axis(box_diag(all));
        float plane = box_mid(all).get(axis);
        std::vector<uint32_t> left, right;  // Iterator::partition preserves order
        for (uint32_t i : insts) (box_mid(inst_box[i]).get(axis) < plane ? left : right).push_back(i);
        if (left.empty()) {
            for (size_t n = 0; n < num_all / 2; ++n) {
                left.push_back(right.back());
                right.pop_back();
            }
        } else if (right.empty()) {
            for (size_t n = 0; n < num_all / 2; ++n) {
                right.push_back(left.back());
                left.pop_back();
            }
        }
        auto l = build(std::move(left));
        auto r = build(std::move(right));
        put_node(nodes[idx], box_merge(node_box(nodes[l.first]), node_box(nodes[r.first])));  // new_internal :50-55
        nodes[idx].a = r.first;
        nodes[idx].b = 0;
        return {idx, std::max(l.second, r.second) + 1};
    }
};

// ---- materials (material/src/lib.rs `bxdfs_at`) ------------------------------------------------------------------
float roughness_to_alpha(float roughness) {  // geometry/src/microfacet.rs:16-23
    float x = pn_max(pn_ln(roughness), -8.0f);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}
pbrs_bxdf bx_zero() {
    pbrs_bxdf b;
    std::memset(&b, 0, sizeof b);
    return b;
}
void set3(float* d, const float* s) { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; }
bool black3(const float* c) { return c[0] <= 0.0f && c[1] <= 0.0f && c[2] <= 0.0f; }  // Color::is_black
pbrs_bxdf bx_lambert(const float* albedo) {
    pbrs_bxdf b = bx_zero();
    b.kind = PBRS_BXDF_DIFFUSE;
    set3(b.albedo, albedo);
    return b;
}
pbrs_bxdf bx_specular(const float* albedo, uint32_t intrusion, uint32_t fresnel, float eta_front, float eta_back) {
    pbrs_bxdf b = bx_zero();
    b.kind = PBRS_BXDF_SPECULAR;
    b.intrusion = intrusion;
    b.fresnel = fresnel;
    set3(b.albedo, albedo);
    b.eta[0] = eta_front;
    b.eta[1] = eta_back;
    return b;
}
pbrs_bxdf bx_microfacet(const float* albedo, float ax, float ay, uint32_t fresnel, const float* eta3, const float* k3) {
    pbrs_bxdf b = bx_zero();
    b.kind = PBRS_BXDF_MICROFACET;
    b.fresnel = fresnel;
    set3(b.albedo, albedo);
    b.alpha_x = ax;
    b.alpha_y = ay;
    if (eta3) set3(b.eta, eta3);
    if (k3) set3(b.k, k3);
    return b;
}
void flatten_material(const pbrs_material_spec& m, pbrs_material* out, std::vector<pbrs_bxdf>& bx) {
    std::memset(out, 0, sizeof *out);
    out->first_bxdf = (uint32_t)bx.size();
    const float* p = m.p;
    const float white[3] = {1.0f, 1.0f, 1.0f};
    // a lobe whose colour is a non-Solid texture: evaluated per hit on the device (pbrs_bxdf::tex)
    auto textured = [&](pbrs_bxdf b, uint32_t tex, bool drop_if_black) {
        b.tex = tex | (drop_if_black ? PBRS_BXDF_TEX_DROP_IF_BLACK : 0u);
        out->flags |= PBRS_MATERIAL_TEXTURED;
        return b;
    };
    switch (m.kind) {
        case PBRS_MTL_LAMBERTIAN:  // :180-184
            bx.push_back(m.tex[0] ? textured(bx_lambert(p), m.tex[0], false) : bx_lambert(p));
            break;
        case PBRS_MTL_METAL: {                                          // :200-206
            float alpha = roughness_to_alpha(p[6]);
            bx.push_back(bx_microfacet(white, alpha, alpha, PBRS_FRESNEL_CONDUCTOR, p, p + 3));
            break;
        }
        case PBRS_MTL_GLOSSY: {  // :71-78, :216-218
            float alpha = roughness_to_alpha(p[3]);
            bx.push_back(bx_microfacet(p, alpha, alpha, PBRS_FRESNEL_NOP, nullptr, nullptr));
            break;
        }
        case PBRS_MTL_MIRROR: bx.push_back(bx_specular(p, PBRS_REFLECTION, PBRS_FRESNEL_NOP, 0.0f, 0.0f)); break;  // :229-232
        case PBRS_MTL_PLASTIC: {                                                                                   // :433-445
            float alpha = (m.flags & PBRS_MTL_FLAG_REMAP_ROUGHNESS) ? roughness_to_alpha(p[6]) : p[6];
            bx.push_back(bx_microfacet(p + 3, alpha, alpha, PBRS_FRESNEL_NOP, nullptr, nullptr));
            bx.push_back(bx_lambert(p));
            break;
        }
        case PBRS_MTL_DIELECTRIC:  // :265-268 — `reflect` colour for the single hybrid lobe (Q18)
            bx.push_back(bx_specular(p + 1, PBRS_HYBRID, PBRS_FRESNEL_DIELECTRIC, 1.0f, p[0]));
            break;
        case PBRS_MTL_DIFFUSE_LIGHT: set3(out->emission, p); break;  // :291-296
        case PBRS_MTL_UBER: {                                        // :317-365
            float opacity = p[15], eta = p[14];
            float tr = pn_clamp(1.0f - opacity, 0.0f, 1.0f);
            float transmission[3] = {tr, tr, tr};
            if (!black3(transmission)) bx.push_back(bx_specular(transmission, PBRS_TRANSMISSION, PBRS_FRESNEL_DIELECTRIC, 1.0f, eta));
            if (m.tex[0]) bx.push_back(textured(bx_lambert(p), m.tex[0], true));
            else if (!black3(p)) bx.push_back(bx_lambert(p));
            if (m.tex[1] || !black3(p + 3)) {
                float au = p[12], av = p[13];
                if (m.flags & PBRS_MTL_FLAG_REMAP_ROUGHNESS) {
                    au = roughness_to_alpha(p[12]);
                    av = roughness_to_alpha(p[13]);
                }
                float etas[3] = {1.0f, eta, 0.0f};
                pbrs_bxdf b = bx_microfacet(p + 3, au, av, PBRS_FRESNEL_DIELECTRIC, etas, nullptr);
                bx.push_back(m.tex[1] ? textured(b, m.tex[1], true) : b);
            }
            if (m.flags & PBRS_MTL_FLAG_HAS_KR) {
                pbrs_bxdf b = bx_specular(p + 6, PBRS_HYBRID, PBRS_FRESNEL_DIELECTRIC, 1.0f, eta);
                if (m.tex[2]) bx.push_back(textured(b, m.tex[2], true));
                else if (!black3(p + 6)) bx.push_back(b);
            }
            if (m.flags & PBRS_MTL_FLAG_HAS_KT) {
                pbrs_bxdf b = bx_specular(p + 9, PBRS_TRANSMISSION, PBRS_FRESNEL_DIELECTRIC, 1.0f, eta);
                if (m.tex[3]) bx.push_back(textured(b, m.tex[3], true));
                else if (!black3(p + 9)) bx.push_back(b);
            }
            break;
        }
        case PBRS_MTL_SUBSTRATE:  // :393-420 — degenerates to Lambert (Q18)
            if (!(black3(p) && black3(p + 3))) bx.push_back(bx_lambert(p));
            break;
        case PBRS_MTL_FOURIER: {  // :467-470: one FourierBSDF over the material's table
            pbrs_bxdf b = bx_zero();
            b.kind = PBRS_BXDF_FOURIER;
            b.intrusion = m.tex[0];
            bx.push_back(b);
            break;
        }
        default: break;
    }
    out->n_bxdfs = (uint32_t)bx.size() - out->first_bxdf;
    // pbrs_material::vis_bxdf: the colour `scatter` returns, for normal_visualizer (src/directlighting.rs:283)
    const float black[3] = {0.0f, 0.0f, 0.0f};
    pbrs_bxdf vis = bx_lambert(black);
    switch (m.kind) {
        case PBRS_MTL_LAMBERTIAN:  // `self.albedo.value(isect.uv, isect.pos)`, :177
            set3(vis.albedo, p);
            vis.tex = m.tex[0];
            break;
        case PBRS_MTL_MIRROR: set3(vis.albedo, p); break;          // :227
        case PBRS_MTL_PLASTIC: set3(vis.albedo, p); break;         // `self.diffuse`, :431
        case PBRS_MTL_DIELECTRIC: set3(vis.albedo, p + 4); break;  // `self.transmit` (`reflect` is the lobe's colour), :256
        default: break;  // Metal: Fresnel of the lobe; DiffuseLight: black; the others never return
    }
    bx.push_back(vis);
    out->vis_bxdf = (uint32_t)bx.size();
}

}  // namespace

struct pbrs_host_scene {
    std::vector<pbrs_node> tlas_nodes, blas_nodes;
    std::vector<pbrs_instance> instances;
    std::vector<pbrs_shape> shapes;
    std::vector<pbrs_mesh> meshes;
    std::vector<pbrs_tri_verts> tri_verts;
    std::vector<pbrs_tri_shade> tri_shade;
    std::vector<pbrs_material> materials;
    std::vector<pbrs_bxdf> bxdfs;
    std::vector<pbrs_area_light> area_lights;
    std::vector<pbrs_delta_light> delta_lights;
    std::vector<pbrs_texture> textures;
    std::vector<float> tex_floats;
    std::vector<uint32_t> tex_words;
    std::vector<pbrs_fourier_table> fourier_tables;
    pbrs_scene_desc desc;
    pbrs_camera camera;
    uint32_t stack_depth;
};

extern "C" {

const char* pbrs_host_last_error(void) { return g_error.c_str(); }

int pbrs_host_scene_build(const pbrs_scene_spec* spec, pbrs_host_scene** out) {
    if (!spec || !out) {
        g_error = "null argument";
        return PBRS_E_INVALID;
    }
    if (spec->n_instances == 0) {
        g_error = "empty instances";  // tlas/src/bvh.rs:117
        return PBRS_E_INVALID;
    }
    auto hs = std::make_unique<pbrs_host_scene>();
    // -- meshes: TriangleMesh::from_soa (shape/src/blas.rs:134-159)
    uint32_t max_blas_height = 0;
    for (uint32_t mi = 0; mi < spec->n_meshes; ++mi) {
        const pbrs_mesh_spec& m = spec->meshes[mi];
        if (m.n_triangles == 0) {
            g_error = "mesh without triangles";
            return PBRS_E_INVALID;
        }
        std::vector<Tri> tris(m.n_triangles);
        for (uint32_t t = 0; t < m.n_triangles; ++t) {
            uint32_t i = m.indices[3 * t], j = m.indices[3 * t + 1], k = m.indices[3 * t + 2];
            if (i >= m.n_vertices || j >= m.n_vertices || k >= m.n_vertices) {
                g_error = "triangle index out of range";
                return PBRS_E_INVALID;
            }
            Box b = box_grow(box_of(v3p(m.positions + 3 * i), v3p(m.positions + 3 * j)), v3p(m.positions + 3 * k));
            tris[t] = Tri{i, j, k, b, t};
        }
        pbrs_mesh pm{};
        pm.first_tri = (uint32_t)hs->tri_verts.size();
        pm.n_tris = m.n_triangles;
        uint32_t node0 = (uint32_t)hs->blas_nodes.size();
        BlasBuilder bb{tris, hs->blas_nodes, pm.first_tri};
        auto root = bb.build(0, tris.size());
        pm.root = root.first;
        pm.height = root.second;
        pm.n_nodes = (uint32_t)hs->blas_nodes.size() - node0;
        max_blas_height = std::max(max_blas_height, pm.height);
        bool flat_ok = true, smooth_ok = true;
        for (const Tri& t : tris) {
            // `let (i, k, j) = tri.index_triple` (blas.rs:162): vertex order read by the mesh is (i, 3rd, 2nd)
            uint32_t v0 = t.i, v1 = t.k, v2 = t.j;
            V3 p0 = v3p(m.positions + 3 * v0), p1 = v3p(m.positions + 3 * v1), p2 = v3p(m.positions + 3 * v2);
            pbrs_tri_verts tv{};
            set3(tv.p0, m.positions + 3 * v0);
            set3(tv.p1, m.positions + 3 * v1);
            set3(tv.p2, m.positions + 3 * v2);
            // `(p0 - p1).cross(p2 - p1).try_hat()` (simple.rs:436; Vec3::try_hat math/src/hcm.rs:118-121)
            V3 gn{pn_nan(), pn_nan(), pn_nan()};
            bool has_normal = false;
            {
                V3 c = cross3(sub(p0, p1), sub(p2, p1));
                float inv_length = 1.0f / length(c);
                if (pn_isfinite(inv_length) && inv_length != 0.0f) {
                    gn = scale(c, inv_length);
                    has_normal = true;
                }
            }
            tv.nx = gn.x; tv.ny = gn.y; tv.nz = gn.z;
            hs->tri_verts.push_back(tv);
            pbrs_tri_shade ts{};
            set3(ts.n0, m.normals + 3 * v0);
            set3(ts.n1, m.normals + 3 * v1);
            set3(ts.n2, m.normals + 3 * v2);
            ts.uv0[0] = m.uvs[2 * v0]; ts.uv0[1] = m.uvs[2 * v0 + 1];
            ts.uv1[0] = m.uvs[2 * v1]; ts.uv1[1] = m.uvs[2 * v1 + 1];
            ts.uv2[0] = m.uvs[2 * v2]; ts.uv2[1] = m.uvs[2 * v2 + 1];
            ts.orig = t.orig;
            hs->tri_shade.push_back(ts);
            // dpdu of blas.rs:186-190, a property of the triangle; evaluated with the reference's operand order
            V3 dpdu_raw;
            {
                float u0 = ts.uv0[0], w0 = ts.uv0[1];
                float u1 = ts.uv1[0] - u0, w1 = ts.uv1[1] - w0;
                float u2 = ts.uv2[0] - u0, w2 = ts.uv2[1] - w0;
                float den = u1 * w2 - u2 * w1;
                V3 num = sub(scale(sub(p2, p0), w2), scale(sub(p1, p0), w1));
                dpdu_raw = V3{num.x / den, num.y / den, num.z / den};
                if (!pn_isfinite(dot3(dpdu_raw, dpdu_raw))) dpdu_raw = sub(p1, p0);
            }
            // PBRS_MESH_FLAT_SHADING_OK: with n0 == n1 == n2 (bitwise) barycentric_lerp returns that normal for any
            // finite barycentrics ((a-c)*b0 + (b-c)*b1 + c = 0 + 0 + c), `facing` only flips its sign, and the
            // projection / hat / abs(dot) of blas.rs:186-193 are invariant under that flip: the Q22 test is a
            // property of the triangle.  Evaluated here with the reference's operand order.
            if (flat_ok) {
                bool same = std::memcmp(ts.n0, ts.n1, 12) == 0 && std::memcmp(ts.n0, ts.n2, 12) == 0;
                bool pass = false;
                if (same && has_normal) {
                    V3 n0 = v3p(ts.n0);
                    V3 n = n0;
                    float inv_n = 1.0f / length(n0);
                    if (pn_isfinite(inv_n) && inv_n != 0.0f) n = scale(n0, inv_n); else n = gn;  // `.try_hat().unwrap_or(hit.normal)`
                    V3 dpdu = dpdu_raw;
                    float dn = dot3(dpdu, n);
                    V3 sn = scale(n, dn);
                    float n2 = dot3(n, n);
                    V3 proj{sn.x / n2, sn.y / n2, sn.z / n2};  // projected_onto: self.dot(other) * other / other.norm_squared()
                    dpdu = normalized(sub(dpdu, proj));
                    pass = !(pn_abs(dot3(dpdu, n)) >= 1e-3f);
                }
                flat_ok = same && pass;
            }
            if (smooth_ok) smooth_ok = !has_normal || smooth_shading_bound(dpdu_raw, v3p(ts.n0), v3p(ts.n1), v3p(ts.n2), gn);
        }
        if (flat_ok) pm.flags |= PBRS_MESH_FLAT_SHADING_OK;
        if (smooth_ok) pm.flags |= PBRS_MESH_SMOOTH_SHADING_OK;
        hs->meshes.push_back(pm);
    }
    // -- analytic shapes
    std::vector<uint32_t> shape_slot(spec->n_shapes, 0), shape_tri(spec->n_shapes, 0);
    for (uint32_t s = 0; s < spec->n_shapes; ++s) {
        const pbrs_shape_spec& sp = spec->shapes[s];
        if (sp.kind == PBRS_SHAPE_MESH) {
            if (sp.mesh >= spec->n_meshes) {
                g_error = "shape references a missing mesh";
                return PBRS_E_INVALID;
            }
            shape_slot[s] = sp.mesh;
            continue;
        }
        if (sp.kind > PBRS_SHAPE_MESH) {
            g_error = "unknown shape kind";
            return PBRS_E_INVALID;
        }
        pbrs_shape ps{};
        std::memcpy(ps.p, sp.p, 9 * sizeof(float));
        if (sp.kind == PBRS_SHAPE_CUBOID) {  // Cuboid::from_points, simple.rs:173-182 (float::min_max)
            for (int a = 0; a < 3; ++a) {
                float lo = sp.p[a] < sp.p[3 + a] ? sp.p[a] : sp.p[3 + a];
                float hi = sp.p[a] < sp.p[3 + a] ? sp.p[3 + a] : sp.p[a];
                ps.p[a] = lo;
                ps.p[3 + a] = hi;
            }
        } else if (sp.kind == PBRS_SHAPE_DISK) {  // Disk::new normalises the normal, simple.rs:42-51
            V3 n = normalized(v3p(sp.p + 3));
            ps.p[3] = n.x; ps.p[4] = n.y; ps.p[5] = n.z;
        }
        shape_slot[s] = (uint32_t)hs->shapes.size();
        hs->shapes.push_back(ps);
        if (sp.kind == PBRS_SHAPE_TRIANGLE) {
            // IsolatedTriangle::intersect / occludes (simple.rs:417-433) call the same intersect_triangle(_pred) as a mesh
            // triangle, on (p0, p1, p2) as given: the traversal kernels test it through the triangle record path, so the
            // shape also gets a record (geometric normal precomputed as for mesh triangles; no shading attributes).
            V3 p0 = v3p(sp.p), p1 = v3p(sp.p + 3), p2 = v3p(sp.p + 6);
            pbrs_tri_verts tv{};
            set3(tv.p0, sp.p);
            set3(tv.p1, sp.p + 3);
            set3(tv.p2, sp.p + 6);
            V3 gn{pn_nan(), pn_nan(), pn_nan()};
            V3 c = cross3(sub(p0, p1), sub(p2, p1));
            float inv_length = 1.0f / length(c);
            if (pn_isfinite(inv_length) && inv_length != 0.0f) gn = scale(c, inv_length);
            tv.nx = gn.x; tv.ny = gn.y; tv.nz = gn.z;
            shape_tri[s] = (uint32_t)hs->tri_verts.size();
            hs->tri_verts.push_back(tv);
            hs->tri_shade.push_back(pbrs_tri_shade{});
        }
    }
    // -- textures (texture/src/lib.rs): tables and texels pooled into two flat arrays
    for (uint32_t t = 0; t < spec->n_textures; ++t) {
        const pbrs_texture_spec& ts = spec->textures[t];
        pbrs_texture pt{};
        pt.kind = ts.kind;
        set3(pt.odd, ts.odd);
        set3(pt.even, ts.even);
        pt.freq = ts.freq;
        pt.width = ts.width;
        pt.height = ts.height;
        pt.data = (uint32_t)hs->tex_floats.size();
        pt.perm = (uint32_t)hs->tex_words.size();
        if (ts.kind == PBRS_TEX_PERLIN) {
            if (!ts.data || !ts.perm) {
                g_error = "perlin texture without tables";
                return PBRS_E_INVALID;
            }
            for (uint32_t k = 0; k < 768; ++k)
                if (ts.perm[k] > 255u) {
                    g_error = "perlin permutation entry above 255";
                    return PBRS_E_INVALID;
                }
            hs->tex_floats.insert(hs->tex_floats.end(), ts.data, ts.data + 768);
            hs->tex_words.insert(hs->tex_words.end(), ts.perm, ts.perm + 768);
        } else if (ts.kind == PBRS_TEX_IMAGE) {
            if (!ts.data || ts.width == 0 || ts.height == 0) {
                g_error = "image texture without texels";
                return PBRS_E_INVALID;
            }
            hs->tex_floats.insert(hs->tex_floats.end(), ts.data, ts.data + 3 * (size_t)ts.width * ts.height);
        } else if (ts.kind != PBRS_TEX_CHECKER) {
            g_error = "unknown texture kind";
            return PBRS_E_INVALID;
        }
        hs->textures.push_back(pt);
    }
    if (spec->env_kind > PBRS_ENV_DUSK ||
        (spec->env_kind == PBRS_ENV_IMAGE && (spec->env_texture >= spec->n_textures || spec->textures[spec->env_texture].kind != PBRS_TEX_IMAGE))) {
        g_error = "bad environment light";
        return PBRS_E_INVALID;
    }
    // -- Fourier BSDF tables: FourierTable::build (geometry/src/fourier.rs:115-151) over the arrays of the file; what its
    // asserts and slice bounds would stop is an error here
    for (uint32_t t = 0; t < spec->n_fourier_tables; ++t) {
        const pbrs_fourier_table_spec& fs = spec->fourier_tables[t];
        const size_t n = fs.n_mu, nn = n * n;
        if (n < 3 || n > 4096 || (fs.n_channels != 1 && fs.n_channels != 3) || !fs.mu || !fs.cdf || !fs.offset_and_length || (fs.n_coeffs && !fs.a)) {
            g_error = "Fourier table: bad sizes or missing arrays";
            return PBRS_E_INVALID;
        }
        // :198-200 asserts mu[i] <= mu[i + 1]; a repeated or non-finite node makes catmull_rom_weights divide 0 by 0 there
        // (NaN weights, then a knot index of -1 upstream): such a table is refused here
        for (size_t i = 0; i < n; ++i)
            if (!pn_isfinite(fs.mu[i]) || (i + 1 < n && !(fs.mu[i] < fs.mu[i + 1]))) {
                g_error = "Fourier table: mu is not finite and strictly ascending";
                return PBRS_E_INVALID;
            }
        int32_t m_max = 0;
        for (size_t i = 0; i < nn; ++i) {
            const int32_t off = fs.offset_and_length[2 * i], len = fs.offset_and_length[2 * i + 1];
            if (off < 0 || len < 0 || (uint64_t)off + (uint64_t)len * fs.n_channels > fs.n_coeffs) {
                g_error = "Fourier table: a coefficient series lies outside the coefficient array";  // :127-130
                return PBRS_E_INVALID;
            }
            m_max = len > m_max ? len : m_max;
        }
        pbrs_fourier_table ft{};
        ft.n_mu = fs.n_mu;
        ft.n_channels = fs.n_channels;
        ft.m_max = (uint32_t)m_max;
        ft.n_coeffs = fs.n_coeffs;
        auto& F = hs->tex_floats;
        auto& W = hs->tex_words;
        if (F.size() + n + 2 * nn + fs.n_coeffs + (size_t)m_max > 0xffffffffull || W.size() + 2 * nn > 0xffffffffull) {
            g_error = "Fourier table: texture pools exceed 2^32 entries";
            return PBRS_E_INVALID;
        }
        ft.mu = (uint32_t)F.size();
        F.insert(F.end(), fs.mu, fs.mu + n);
        ft.cdf = (uint32_t)F.size();
        F.insert(F.end(), fs.cdf, fs.cdf + nn);
        ft.a0 = (uint32_t)F.size();
        for (size_t i = 0; i < nn; ++i)  // :131-137
            F.push_back(fs.offset_and_length[2 * i + 1] > 0 ? fs.a[fs.offset_and_length[2 * i]] : 0.0f);
        ft.a = (uint32_t)F.size();
        F.insert(F.end(), fs.a, fs.a + fs.n_coeffs);
        ft.recip = (uint32_t)F.size();
        for (int32_t i = 0; i < m_max; ++i) F.push_back(1.0f / (float)i);  // :138 (`(i as f32).recip()`; entry 0 is never read)
        ft.a_offset = (uint32_t)W.size();
        for (size_t i = 0; i < nn; ++i) W.push_back((uint32_t)fs.offset_and_length[2 * i]);
        ft.m_lookup = (uint32_t)W.size();
        for (size_t i = 0; i < nn; ++i) W.push_back((uint32_t)fs.offset_and_length[2 * i + 1]);
        hs->fourier_tables.push_back(ft);
    }
    // -- materials
    for (uint32_t m = 0; m < spec->n_materials; ++m) {
        if (spec->materials[m].kind == PBRS_MTL_FOURIER) {
            if (spec->materials[m].tex[0] >= spec->n_fourier_tables) {
                g_error = "Fourier material references a missing table";
                return PBRS_E_INVALID;
            }
        } else
        for (int k = 0; k < 4; ++k)
            if (spec->materials[m].tex[k] > spec->n_textures) {
                g_error = "material references a missing texture";
                return PBRS_E_INVALID;
            }
        pbrs_material pm;
        flatten_material(spec->materials[m], &pm, hs->bxdfs);
        {  // material_visualizer's `match mtl.summary()` (src/directlighting.rs:248-259; summaries: material/src/lib.rs)
            static const uint32_t palette_of_kind[] = {/* Lambertian */ 8, /* Metal */ 7, /* Glossy: no arm */ 9, /* Mirror */ 5,
                                                       /* plastic */ 0,    /* Dielectric */ 4, /* DiffuseLight */ 3, /* uber */ 2,
                                                       /* substrate */ 1};
            const uint32_t kind = spec->materials[m].kind;
            pm.vis_class = kind <= PBRS_MTL_SUBSTRATE ? palette_of_kind[kind] : kind == PBRS_MTL_FOURIER ? 6u : 9u;  // "Fourier" => 6
        }
        if (pm.n_bxdfs > PBRS_MAX_BXDFS) {
            g_error = "material with more than PBRS_MAX_BXDFS lobes";
            return PBRS_E_INVALID;
        }
        hs->materials.push_back(pm);
    }
    // -- instances + their world boxes (Instance::bbox, tlas/src/instance.rs:47-49)
    std::vector<Box> inst_box(spec->n_instances);
    for (uint32_t i = 0; i < spec->n_instances; ++i) {
        const pbrs_instance_spec& is = spec->instances[i];
        if (is.shape >= spec->n_shapes || is.material >= spec->n_materials) {
            g_error = "instance references a missing shape or material";
            return PBRS_E_INVALID;
        }
        pbrs_instance pi{};
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) {
                pi.inv[r][c] = is.inverse[4 * c + r];
                pi.fwd[r][c] = is.forward[4 * c + r];
            }
        pi.shape_kind = spec->shapes[is.shape].kind;
        pi.shape_index = shape_slot[is.shape];
        pi.material = is.material;
        if (pi.shape_kind == PBRS_SHAPE_MESH) {
            pi.blas_root = hs->meshes[pi.shape_index].root;
            pi.mesh_flags = hs->meshes[pi.shape_index].flags;
        } else if (pi.shape_kind == PBRS_SHAPE_TRIANGLE) {
            pi.blas_root = shape_tri[is.shape];  // its record in tri_verts[]
        }
        {
            static const float kIdentity[3][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}};
            if (std::memcmp(pi.inv, kIdentity, sizeof kIdentity) == 0 && std::memcmp(pi.fwd, kIdentity, sizeof kIdentity) == 0)
                pi.flags |= PBRS_INSTANCE_IDENTITY;
        }
        hs->instances.push_back(pi);
        M4 fwd;
        std::memcpy(fwd.m, is.forward, sizeof fwd.m);
        inst_box[i] = xf_box(fwd, shape_box(spec->shapes[is.shape], hs->meshes, hs->blas_nodes));
    }
    // -- TLAS
    std::vector<uint32_t> all(spec->n_instances);
    for (uint32_t i = 0; i < spec->n_instances; ++i) all[i] = i;
    TlasBuilder tb{inst_box, hs->instances, hs->tlas_nodes};
    auto troot = tb.build(std::move(all));
    // -- lights (light/src/lib.rs:114-121; areas: sample_shape.rs:252-254, :271-273, :291-293, :306-308)
    for (uint32_t l = 0; l < spec->n_area_lights; ++l) {
        const pbrs_area_light_spec& a = spec->area_lights[l];
        pbrs_area_light pl{};
        set3(pl.emit, a.emit);
        pl.shape_kind = a.shape.kind;
        std::memcpy(pl.p, a.shape.p, 9 * sizeof(float));
        const float* p = a.shape.p;
        switch (a.shape.kind) {
            case PBRS_SHAPE_SPHERE: pl.area = pn_sq(p[3]) * 4.0f * PN_PI; break;
            case PBRS_SHAPE_DISK: {
                V3 n = normalized(v3p(p + 3));
                pl.p[3] = n.x; pl.p[4] = n.y; pl.p[5] = n.z;
                pl.area = dot3(v3p(p + 6), v3p(p + 6)) * PN_PI;
                break;
            }
            case PBRS_SHAPE_TRIANGLE: pl.area = length(cross3(sub(v3p(p), v3p(p + 3)), sub(v3p(p + 6), v3p(p + 3)))) * 0.5f; break;
            case PBRS_SHAPE_QUAD: pl.area = length(cross3(v3p(p + 3), v3p(p + 6))); break;
            default: g_error = "area light shape must be sphere, disk, triangle or quad"; return PBRS_E_INVALID;
        }
        hs->area_lights.push_back(pl);
    }
    for (uint32_t l = 0; l < spec->n_delta_lights; ++l) {
        const pbrs_delta_light_spec& d = spec->delta_lights[l];
        pbrs_delta_light pd{};
        pd.kind = d.kind;
        set3(pd.v, d.v);
        set3(pd.color, d.color);
        pd.world_radius = d.world_radius;
        hs->delta_lights.push_back(pd);
    }
    // -- camera (geometry/src/camera.rs:19-44 and the hoisted products of :68-70)
    {
        const pbrs_camera_spec& cs = spec->camera;
        if (cs.width < 2 || cs.height < 2) {
            g_error = "camera resolution below 2x2";
            return PBRS_E_INVALID;
        }
        float aspect_ratio = (float)cs.width / (float)cs.height;
        float half_vertical = pn_tan(cs.fov_y_rad * 0.5f);
        float half_horizontal = half_vertical * aspect_ratio;
        V3 a{half_horizontal / (float)(cs.width / 2), 0.0f, 0.0f};
        V3 b{0.0f, -half_vertical / (float)(cs.height / 2), 0.0f};
        V3 c{-half_horizontal, half_vertical, 1.0f};
        V3 from = v3p(cs.from), target = v3p(cs.target), up = v3p(cs.up);
        V3 forward = normalized(sub(target, from));
        V3 right = normalized(cross3(up, forward));
        up = cross3(forward, right);
        auto mul = [&](V3 v) {  // Mat3 * Vec3, math/src/hcm.rs:448-453, cols = (right, up, forward)
            return add(add(scale(right, v.x), scale(up, v.y)), scale(forward, v.z));
        };
        V3 oc = mul(c), oa = mul(a), ob = mul(b);
        pbrs_camera& cam = hs->camera;
        std::memset(&cam, 0, sizeof cam);
        cam.center[0] = from.x; cam.center[1] = from.y; cam.center[2] = from.z;
        cam.c[0] = oc.x; cam.c[1] = oc.y; cam.c[2] = oc.z;
        cam.a[0] = oa.x; cam.a[1] = oa.y; cam.a[2] = oa.z;
        cam.b[0] = ob.x; cam.b[1] = ob.y; cam.b[2] = ob.z;
        cam.width = cs.width;
        cam.height = cs.height;
    }
    pbrs_scene_desc& d = hs->desc;
    std::memset(&d, 0, sizeof d);
    d.n_tlas_nodes = (uint32_t)hs->tlas_nodes.size(); d.tlas_nodes = hs->tlas_nodes.data();
    d.tlas_height = troot.second;
    d.n_instances = (uint32_t)hs->instances.size(); d.instances = hs->instances.data();
    d.n_shapes = (uint32_t)hs->shapes.size(); d.shapes = hs->shapes.data();
    d.n_meshes = (uint32_t)hs->meshes.size(); d.meshes = hs->meshes.data();
    d.n_blas_nodes = (uint32_t)hs->blas_nodes.size(); d.blas_nodes = hs->blas_nodes.data();
    d.n_triangles = (uint32_t)hs->tri_verts.size(); d.tri_verts = hs->tri_verts.data(); d.tri_shade = hs->tri_shade.data();
    d.n_materials = (uint32_t)hs->materials.size(); d.materials = hs->materials.data();
    d.n_bxdfs = (uint32_t)hs->bxdfs.size(); d.bxdfs = hs->bxdfs.data();
    d.n_area_lights = (uint32_t)hs->area_lights.size(); d.area_lights = hs->area_lights.data();
    d.n_delta_lights = (uint32_t)hs->delta_lights.size(); d.delta_lights = hs->delta_lights.data();
    set3(d.env_constant, spec->env_constant);
    d.env_kind = spec->env_kind;
    d.env_texture = spec->env_texture;
    set3(d.env_scale, spec->env_scale);
    d.n_textures = (uint32_t)hs->textures.size(); d.textures = hs->textures.data();
    d.n_tex_floats = (uint32_t)hs->tex_floats.size(); d.tex_floats = hs->tex_floats.data();
    d.n_tex_words = (uint32_t)hs->tex_words.size(); d.tex_words = hs->tex_words.data();
    d.n_fourier_tables = (uint32_t)hs->fourier_tables.size(); d.fourier_tables = hs->fourier_tables.data();
    hs->stack_depth = troot.second + max_blas_height;
    *out = hs.release();
    return PBRS_OK;
}

void pbrs_host_scene_free(pbrs_host_scene* hs) { delete hs; }
const pbrs_scene_desc* pbrs_host_scene_desc(const pbrs_host_scene* hs) { return &hs->desc; }
const pbrs_camera* pbrs_host_scene_camera(const pbrs_host_scene* hs) { return &hs->camera; }
uint32_t pbrs_host_scene_stack_depth(const pbrs_host_scene* hs) { return hs->stack_depth; }

}  // extern "C"
