// oracle/ref_shapes.cpp — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// Restates geometry/src/interaction.rs, shape/src/simple.rs, shape/src/blas.rs,
// tlas/src/{bvh,instance}.rs, geometry/src/transform.rs:267-320, geometry/src/camera.rs.
#include <algorithm>
#include <cstring>

#include "ref_scene.h"

namespace ref {

thread_local Diag* g_diag = nullptr;
thread_local Counters* g_cnt = nullptr;

// ---- geometry/src/interaction.rs ---------------------------------------------------------------
Interaction isect_new(Point3 pos, float ray_t, float u, float v, Vec3 normal, Vec3 wo) {  // :23-34
    REF_ASSERT(dot(normal, wo) >= 0.0f);
    Interaction i;
    i.pos = pos;
    i.ray_t = ray_t;
    i.u = u;
    i.v = v;
    i.normal = normal;
    i.wo = wo;
    i.tbn = Mat3{{{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}};
    return i;
}
Interaction isect_rayless(Point3 pos, float u, float v, Vec3 normal) {  // :37-39
    return isect_new(pos, 0.0f, u, v, normal, Vec3{0, 0, 0});
}
Interaction with_dpdu(Interaction self, Vec3 dpdu) {  // :45-61
    REF_ASSERT(pn_abs(dot(self.normal, dpdu)) < 1e-3f);
    Vec3 normal = hat(self.normal);
    Vec3 bitangent = hat(cross(normal, dpdu));
    dpdu = cross(bitangent, normal);
    float det = dot(cross(dpdu, bitangent), normal);
    REF_ASSERT(pn_abs(det - 1.0f) < 1e-4f);
    self.tbn = mat3_cols(dpdu, bitangent, normal);
    return self;
}
Ray spawn_ray(const Interaction& i, Vec3 dir) {  // :63-66
    Vec3 out_normal = pn_signum(dot(dir, i.normal)) * i.normal;
    return ray_new(i.pos + out_normal * 0.001f, dir);
}
Ray spawn_limited_ray_to(const Interaction& i, Point3 pos) {  // :68-70
    Ray r = spawn_ray(i, pos - i.pos);
    r.t_max = 1.0f - 0.001f;
    return r;
}
bool has_valid_frame(const Interaction& i) {  // :72-87
    float det = dot(cross(i.tbn.cols[0], i.tbn.cols[1]), i.tbn.cols[2]);
    return pn_abs(det - 1.0f) < 1e-4f;
}

// ---- shape/src/simple.rs ---------------------------------------------------------------------------
static BBox sphere_bbox(const Sphere& s) {  // :203-206
    Vec3 half_diagonal = Vec3{1.0f, 1.0f, 1.0f} * s.radius;
    return bbox_new(s.center - half_diagonal, s.center + half_diagonal);
}
static bool sphere_intersect(const Sphere& s, const Ray& r, Interaction* out) {  // :207-267
    REF_COUNT(spheres);
    Vec3 f = r.origin - s.center;
    float a = norm_squared(r.dir);
    float b_prime = -dot(f, r.dir);
    float delta = s.radius * s.radius - norm_squared(f + b_prime / a * r.dir);
    if (delta < 0.0f) return false;
    float c = norm_squared(f) - s.radius * s.radius;
    float q = b_prime + pn_signum(b_prime) * pn_sqrt(delta * a);
    float t0 = c / q, t1 = q / a;
    float t_low, t_high;
    if (t0 < t1) {
        t_low = t0;
        t_high = t1;
    } else {
        t_low = t1;
        t_high = t0;
    }
    float lo, hi;
    bool has_lo = truncated_t(r, t_low, &lo);
    bool has_hi = truncated_t(r, t_high, &hi);
    float ray_t;
    if (has_lo)
        ray_t = lo;
    else if (has_hi)
        ray_t = hi;
    else
        return false;

    Point3 pos = position_at(r, ray_t);
    Vec3 normal = hat(pos - s.center);
    pos = s.center + normal * s.radius * 1.00001f;

    float theta = pn_acos(normal.y);
    float phi = pn_atan2(normal.z, normal.x) + PN_PI;
    float u = phi / (2.0f * PN_PI), v = theta / PN_PI;

    Vec3 dpdu;
    if (!try_hat(Vec3{-normal.y, normal.x, 0.0f}, &dpdu)) dpdu = Vec3{1.0f, 0.0f, 0.0f};
    REF_ASSERT(distance_to(pos, s.center) >= s.radius);

    // D4: the reference's Interaction::new asserts n·wo >= 0, which panics for
    // every ray that hits a sphere from inside.  Documented deviation: the assert is skipped here.
    Vec3 wo = -r.dir;
    if (!(dot(normal, wo) >= 0.0f) && g_diag) g_diag->sphere_inside++;
    Interaction i;
    i.pos = pos;
    i.ray_t = ray_t;
    i.u = u;
    i.v = v;
    i.normal = normal;
    i.wo = wo;
    i.tbn = Mat3{{{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}};
    *out = with_dpdu(i, dpdu);
    return true;
}
static bool sphere_occludes(const Sphere& s, const Ray& r) {  // :268-288
    REF_COUNT(spheres);
    Vec3 f = r.origin - s.center;
    float a = norm_squared(r.dir);
    float b_prime = -dot(f, r.dir);
    float delta = s.radius * s.radius - norm_squared(f + b_prime / a * r.dir);
    if (delta < 0.0f) return false;
    float c = norm_squared(f) - s.radius * s.radius;
    float q = b_prime + pn_signum(b_prime) * pn_sqrt(delta * a);
    float t0 = c / q, t1 = q / a;
    float tmp;
    return truncated_t(r, t0, &tmp) && truncated_t(r, t1, &tmp);  // Q13: both roots
}

static BBox disk_bbox(const Disk& d) {  // :298-305
    Vec3 v1, v2;
    make_coord_system(d.normal, &v1, &v2);
    float rn = norm(d.radial);
    v1 = v1 * rn;
    v2 = v2 * rn;
    return bbox_union(bbox_new(d.center + v1 + v2, d.center + v1 - v2), bbox_new(d.center - v1 - v2, d.center - v1 + v2));
}
static bool disk_intersect(const Disk& d, const Ray& r, Interaction* out) {  // :306-326
    REF_COUNT(disks);
    float t = dot(d.center - r.origin, d.normal) / dot(r.dir, d.normal);
    if (!truncated_t(r, t, &t)) return false;
    Point3 isect_point = position_at(r, t);
    if (!(squared_distance_to(isect_point, d.center) <= norm_squared(d.radial))) return false;
    Vec3 cp = isect_point - d.center;
    cp = cp - dot(cp, d.normal) * d.normal;
    REF_ASSERT(pn_abs(dot(cp, d.normal)) < 1e-6f);
    Vec3 normal = d.normal * pn_signum(dot(d.normal, -r.dir));
    Vec3 tan = hat(cross(normal, cp));
    float u = pn_atan2(dot(cross(d.radial, cp), normal), dot(d.radial, cp));
    u = pn_fract(u * PN_FRAC_1_PI + 1.0f);
    float v = norm(cp) / norm(d.radial);
    *out = with_dpdu(isect_new(d.center + cp, t, u, v, normal, -r.dir), tan);
    return true;
}
static bool disk_occludes(const Disk& d, const Ray& r) {  // :328-332 (Q14: ignores the t range)
    REF_COUNT(disks);
    float t = dot(d.center - r.origin, d.normal) / dot(r.dir, d.normal);
    Point3 isect_point = position_at(r, t);
    return squared_distance_to(isect_point, d.center) <= norm_squared(d.radial);
}

static BBox quad_bbox(const ParallelQuad& q) {  // :106-113
    BBox bu = bbox_new(q.origin, q.origin + q.side_u);
    BBox bv = bbox_new(q.origin + q.side_v, q.origin + q.side_u + q.side_v);
    return bbox_union(bu, bv);
}
// :120-150.  D1: u, v are norms (unsigned) and the `accurate_hit` assert panics for the mirrored
// part; the benchmark scenes avoid ParallelQuad instances for that reason.
static bool quad_intersect(const ParallelQuad& q, const Ray& r, Interaction* out) {
    REF_COUNT(quads);
    Vec3 normal = facing(cross(q.side_u, q.side_v), r.dir);
    float t = dot(q.origin - r.origin, normal) / dot(r.dir, normal);
    if (!truncated_t(r, t, &t)) return false;
    Point3 coarse_hit = position_at(r, t);
    Vec3 a = q.side_u, b = q.side_v, d = coarse_hit - q.origin;
    float v = norm(cross(a, d)) / norm(cross(a, b));
    float u = norm(cross(b, d)) / norm(cross(b, a));
    if (!((0.0f <= v && v <= 1.0f) && (0.0f <= u && u <= 1.0f))) return false;
    Point3 accurate_hit = q.origin + u * a + b * v;
    REF_ASSERT(distance_to(accurate_hit, coarse_hit) < 1e-3f);
    *out = with_dpdu(isect_new(accurate_hit, t, u, v, hat(normal), -r.dir), q.side_u);
    return true;
}
static bool quad_occludes(const ParallelQuad& q, const Ray& r) {  // :151-163 (D2: inverted t)
    REF_COUNT(quads);
    Vec3 normal = cross(q.side_u, q.side_v);
    float t = dot(r.dir, normal) / dot(q.origin - r.origin, normal);
    if (!truncated_t(r, t, &t)) return false;
    Point3 coarse_hit = position_at(r, t);
    Vec3 a = q.side_u, b = q.side_v, d = coarse_hit - q.origin;
    float v = norm(cross(a, d)) / norm(cross(a, b));
    float u = norm(cross(b, d)) / norm(cross(b, a));
    return (0.0f <= v && v <= 1.0f) && (0.0f <= u && u <= 1.0f);
}

static bool cuboid_intersect(const Cuboid& cb, const Ray& r, Interaction* out) {  // :343-411
    REF_COUNT(cuboids);
    struct HitInfo {
        float t, bound;
        int axis;
    };
    HitInfo hit_min{0.0f, pn_inf(), 0};
    HitInfo hit_max{r.t_max, -pn_inf(), 0};
    for (int axis = 0; axis < 3; ++axis) {
        float inv_dir = 1.0f / r.dir[axis];
        float t0 = (cb.min[axis] - r.origin[axis]) * inv_dir;
        float t1 = (cb.max[axis] - r.origin[axis]) * inv_dir;
        HitInfo hit_0{t0, cb.min[axis], axis};
        HitInfo hit_1{t1, cb.max[axis], axis};
        if (t0 > t1) {
            std::swap(hit_0, hit_1);
            std::swap(t0, t1);
        }
        if (t0 > hit_min.t) hit_min = hit_0;
        if (t1 < hit_max.t) hit_max = hit_1;
        if (hit_max.t < hit_min.t) return false;
    }
    // Interval::new(a, b) asserts no NaN then orders (float.rs:162-167); contains(0.0) (:174-176)
    REF_ASSERT(!pn_isnan(hit_min.t) && !pn_isnan(hit_max.t));
    float lo = hit_min.t < hit_max.t ? hit_min.t : hit_max.t;
    float hi = hit_min.t < hit_max.t ? hit_max.t : hit_min.t;
    HitInfo h = (0.0f >= lo && 0.0f <= hi) ? hit_max : hit_min;
    if (pn_isinf(h.bound)) return false;
    Point3 hit_pos = position_at(r, h.t);
    hit_pos.at(h.axis) = h.bound;
    Vec3 normal{0, 0, 0};
    normal.at(h.axis) = pn_signum(r.dir[h.axis]) * -1.0f;
    Vec3 tan{0, 0, 0};
    tan.at((h.axis + 1) % 3) = 1.0f;
    *out = with_dpdu(isect_new(hit_pos, h.t, 0.5f, 0.5f, normal, -r.dir), tan);
    return true;
}
static bool cuboid_occludes(const Cuboid& cb, const Ray& r) {  // :412-415 (Q14)
    REF_COUNT(cuboids);
    return bbox_intersect(bbox_new(cb.min, cb.max), r);
}

bool intersect_triangle(Point3 p0, Point3 p1, Point3 p2, const Ray& r, Interaction* out) {  // :435-475
    REF_COUNT(triangles);
    Vec3 normal;
    if (!try_hat(cross(p0 - p1, p2 - p1), &normal)) return false;
    normal = facing(normal, r.dir);
    REF_ASSERT(dot(normal, r.dir) <= 0.0f);
    float t = dot(normal, p0 - r.origin) / dot(normal, r.dir);
    if (!truncated_t(r, t, &t)) return false;
    Point3 p = position_at(r, t);
    float b2 = dot(cross(p - p0, p - p1), normal);
    float b0 = dot(cross(p - p1, p - p2), normal);
    float b1 = dot(cross(p - p2, p - p0), normal);
    if (pn_isnan(b0) || pn_isnan(b1) || pn_isnan(b2)) return false;  // Q22
    bool p0s = b0 > 0.0f, p1s = b1 > 0.0f, p2s = b2 > 0.0f;
    if (!((p0s && p1s && p2s) || (!p0s && !p1s && !p2s))) return false;
    float total_area = b0 + b1 + b2;
    b0 = b0 / total_area;
    b1 = b1 / total_area;
    b2 = b2 / total_area;
    Point3 hit_pos = barycentric_lerp(p0, p1, p2, b0, b1);
    if (has_nan(hit_pos)) return false;
    *out = isect_new(hit_pos, t, b1, b2, normal, -r.dir);
    out->b1 = b1;
    out->b2 = b2;
    return true;
}
bool intersect_triangle_pred(Point3 p0, Point3 p1, Point3 p2, const Ray& r) {  // :477-495
    REF_COUNT(triangles);
    Vec3 normal;
    if (!try_hat(cross(p0 - p1, p2 - p1), &normal)) return false;
    float t = dot(normal, p0 - r.origin) / dot(normal, r.dir);
    if (!truncated_t(r, t, &t)) return false;
    Point3 p = position_at(r, t);
    float b0 = dot(cross(p - p0, p - p1), normal);
    float b1 = dot(cross(p - p1, p - p2), normal);
    float b2 = dot(cross(p - p2, p - p0), normal);
    REF_ASSERT(!(pn_isnan(b0) || pn_isnan(b1) || pn_isnan(b2)));
    bool p0s = b0 > 0.0f, p1s = b1 > 0.0f, p2s = b2 > 0.0f;
    return (p0s && p1s && p2s) || (!p0s && !p1s && !p2s);
}

// ---- shape/src/blas.rs -----------------------------------------------------------------------------
size_t IsoBvhNode::height() const {
    return is_leaf ? 1 : std::max(child[0]->height(), child[1]->height()) + 1;
}
size_t IsoBvhNode::count() const { return is_leaf ? 1 : child[0]->count() + child[1]->count() + 1; }

// `partition` crate 0.1.2 is a dependency of the reference (shape/Cargo.toml:7) that is not
// vendored under /root/reference.  Its documented behaviour: in-place, unstable two-pointer
// partition returning (left = predicate holds, right = rest).  Restated here as the Hoare-style
// scan; the intra-side order it leaves is unpinned by anything in the reference (Q12).
template <class T, class P>
static size_t partition_in_place(T* data, size_t len, P pred) {
    if (len == 0) return 0;
    size_t l = 0, r = len - 1;
    for (;;) {
        while (l < len && pred(data[l])) l += 1;
        while (r > 0 && !pred(data[r])) r -= 1;
        if (l >= r) return l;
        std::swap(data[l], data[r]);
    }
}

// blas.rs:333-420
static std::unique_ptr<IsoBvhNode> recursive_build(std::vector<MeshTriangle>& shapes, size_t start, size_t end) {
    auto node = std::make_unique<IsoBvhNode>();
    size_t len = end - start;
    if (len <= 4) {
        BBox b = bbox_empty();
        for (size_t i = start; i < end; ++i) b = bbox_union(b, shapes[i].bbox);
        node->bbox = b;
        node->is_leaf = true;
        node->start = start;
        node->end = end;
        return node;
    }
    std::vector<BBox> bboxes(len);
    for (size_t i = 0; i < len; ++i) bboxes[i] = shapes[start + i].bbox;
    BBox centroid_bbox = bbox_empty();
    for (auto& b : bboxes) centroid_bbox = bbox_union_pt(centroid_bbox, bbox_midpoint(b));
    int split_axis = max_dimension(bbox_diag(centroid_bbox));
    if (bbox_diag(centroid_bbox)[split_axis] < 1e-8f) {
        BBox b = bbox_empty();
        for (auto& bb : bboxes) b = bbox_union(b, bb);
        node->bbox = b;
        node->is_leaf = true;
        node->start = start;
        node->end = end;
        return node;
    }
    // `sort_by` is a stable merge sort; partial_cmp().unwrap() panics on NaN.
    std::stable_sort(bboxes.begin(), bboxes.end(), [&](const BBox& b0, const BBox& b1) {
        return bbox_midpoint(b0)[split_axis] < bbox_midpoint(b1)[split_axis];
    });
    float bbox_area_sum = 0.0f;
    for (auto& b : bboxes) bbox_area_sum += bbox_area(b);
    float pivot_area = bbox_area_sum * 0.5f;
    float partial_sum = 0.0f;
    size_t split_index = 0;
    for (size_t i = 0; i < len; ++i) {
        partial_sum += bbox_area(bboxes[i]);
        if (partial_sum >= pivot_area) {
            split_index = i;
            break;
        }
    }
    float pivot_value = bbox_midpoint(bboxes[split_index])[split_axis];
    size_t left_len = partition_in_place(&shapes[start], len, [&](const MeshTriangle& s) {
        return bbox_midpoint(s.bbox)[split_axis] <= pivot_value;
    });
    size_t mid_point = start + left_len;
    if (left_len == 0 || left_len == len) {
        // `select_nth_unstable_by(len/2)` leaves an implementation-defined order on both sides
        // (Q12).  Documented deviation: a full stable sort by the same key, split at len/2.
        std::stable_sort(shapes.begin() + start, shapes.begin() + end, [&](const MeshTriangle& s0, const MeshTriangle& s1) {
            return bbox_midpoint(s0.bbox)[split_axis] < bbox_midpoint(s1.bbox)[split_axis];
        });
        mid_point = start + len / 2;
    }
    REF_ASSERT(mid_point != start && mid_point != end);
    node->child[0] = recursive_build(shapes, start, mid_point);
    node->child[1] = recursive_build(shapes, mid_point, end);
    node->bbox = bbox_union(node->child[0]->bbox, node->child[1]->bbox);
    node->is_leaf = false;
    node->axis = split_axis;
    return node;
}

std::unique_ptr<TriangleMesh> mesh_from_soa(const pbrs_mesh_spec& m) {  // :134-159
    auto mesh = std::make_unique<TriangleMesh>();
    mesh->positions.resize(m.n_vertices);
    mesh->normals.resize(m.n_vertices);
    mesh->uvs.assign(m.uvs, m.uvs + 2 * (size_t)m.n_vertices);
    for (uint32_t i = 0; i < m.n_vertices; ++i) {
        mesh->positions[i] = Vec3{m.positions[3 * i], m.positions[3 * i + 1], m.positions[3 * i + 2]};
        mesh->normals[i] = Vec3{m.normals[3 * i], m.normals[3 * i + 1], m.normals[3 * i + 2]};
    }
    mesh->triangles.resize(m.n_triangles);
    for (uint32_t t = 0; t < m.n_triangles; ++t) {
        uint32_t i = m.indices[3 * t], j = m.indices[3 * t + 1], k = m.indices[3 * t + 2];
        BBox b = bbox_union_pt(bbox_new(mesh->positions[i], mesh->positions[j]), mesh->positions[k]);
        mesh->triangles[t] = MeshTriangle{i, j, k, b, t};
    }
    mesh->bvh_root = recursive_build(mesh->triangles, 0, mesh->triangles.size());
    return mesh;
}

bool TriangleMesh::intersect_one(const MeshTriangle& tri, const Ray& r, Interaction* out) const {  // :161-207
    // `let (i, k, j) = tri.index_triple;` — Q11: the 2nd and 3rd indices trade places.
    uint32_t i = tri.i, k = tri.j, j = tri.k;
    Point3 p0 = positions[i], p1 = positions[j], p2 = positions[k];
    Interaction hit;
    if (!intersect_triangle(p0, p1, p2, r, &hit)) return false;
    REF_COUNT(tri_shading);
    float b0 = 1.0f - hit.u - hit.v, b1 = hit.u, b2 = hit.v;
    Point3 hit_by_uv = p0 + (p1 - p0) * b1 + (p2 - p0) * b2;
    REF_ASSERT(squared_distance_to(hit_by_uv, hit.pos) < 1e-6f);
    Vec3 n0 = normals[i], n1 = normals[j], n2 = normals[k];
    Vec3 bclerp_normal;
    if (!try_hat(barycentric_lerp(n0, n1, n2, b0, b1), &bclerp_normal)) bclerp_normal = hit.normal;
    bclerp_normal = facing(bclerp_normal, r.dir);
    float bclerp_u = barycentric_lerp(uvs[2 * i], uvs[2 * j], uvs[2 * k], b0, b1);
    float bclerp_v = barycentric_lerp(uvs[2 * i + 1], uvs[2 * j + 1], uvs[2 * k + 1], b0, b1);
    float u0 = uvs[2 * i], v0 = uvs[2 * i + 1];
    float u1 = uvs[2 * j], v1 = uvs[2 * j + 1];
    float u2 = uvs[2 * k], v2 = uvs[2 * k + 1];
    u1 = u1 - u0;
    v1 = v1 - v0;
    u2 = u2 - u0;
    v2 = v2 - v0;
    Vec3 dpdu = ((p2 - p0) * v2 - (p1 - p0) * v1) / (u1 * v2 - u2 * v1);
    if (!pn_isfinite(norm_squared(dpdu))) dpdu = p1 - p0;
    dpdu = hat(dpdu - projected_onto(dpdu, bclerp_normal));
    if (pn_abs(dot(dpdu, bclerp_normal)) >= 1e-3f) return false;  // Q22: hit silently dropped
    Interaction res = isect_new(hit.pos, hit.ray_t, bclerp_u, bclerp_v, bclerp_normal, hit.wo);
    res.prim = tri.orig;
    res.b1 = hit.b1;
    res.b2 = hit.b2;
    *out = with_dpdu(res, dpdu);
    return true;
}
bool TriangleMesh::intersect_one_pred(const MeshTriangle& tri, const Ray& r) const {  // :208-211
    uint32_t i = tri.i, k = tri.j, j = tri.k;
    return intersect_triangle_pred(positions[i], positions[j], positions[k], r);
}

// blas.rs:422-476
static bool intersect_bvh(const TriangleMesh& mesh, const IsoBvhNode* tree, const Ray& r, Interaction* out) {
    // :428 tests the root box, then the loop pops the root and tests it again with the same ray (:441):
    // one box test as far as the work counters go (the second has the same operands and result).
    if (!bbox_intersect(tree->bbox, r)) {
        REF_COUNT(blas_nodes);
        return false;
    }
    std::vector<const IsoBvhNode*> node_stack;
    node_stack.reserve(60);
    node_stack.push_back(tree);
    Interaction outer_hit = isect_new(Point3{0, 0, 0}, pn_inf(), 0.0f, 0.0f, Vec3{0, 0, 1}, Vec3{0, 0, 1});
    Ray ray = r;
    while (!node_stack.empty()) {
        const IsoBvhNode* node = node_stack.back();
        node_stack.pop_back();
        REF_COUNT(blas_nodes);
        if (bbox_intersect(node->bbox, ray)) {
            if (node->is_leaf) {
                for (size_t s = node->start; s < node->end; ++s) {
                    Interaction new_isect;
                    if (mesh.intersect_one(mesh.triangles[s], ray, &new_isect)) {
                        REF_ASSERT(!has_nan(new_isect.pos));
                        REF_ASSERT(has_valid_frame(new_isect));
                        if (new_isect.ray_t < outer_hit.ray_t) outer_hit = new_isect;
                    }
                }
            } else {
                if (ray.dir[node->axis] > 0.0f) {
                    node_stack.push_back(node->child[1].get());
                    node_stack.push_back(node->child[0].get());
                } else {
                    node_stack.push_back(node->child[0].get());
                    node_stack.push_back(node->child[1].get());
                }
            }
            // blas.rs:468 sits after the match, i.e. it is skipped by the `continue` on a bbox miss.
            ray.t_max = outer_hit.ray_t;
        }
    }
    if (outer_hit.ray_t < pn_inf()) {
        *out = outer_hit;
        return true;
    }
    return false;
}
// blas.rs:478-495
static bool intersect_bvh_pred(const TriangleMesh& mesh, const IsoBvhNode* tree, const Ray& r) {
    REF_COUNT(blas_nodes);
    if (!bbox_intersect(tree->bbox, r)) return false;
    if (tree->is_leaf) {
        for (size_t s = tree->start; s < tree->end; ++s)
            if (mesh.intersect_one_pred(mesh.triangles[s], r)) return true;
        return false;
    }
    return intersect_bvh_pred(mesh, tree->child[0].get(), r) || intersect_bvh_pred(mesh, tree->child[1].get(), r);
}

BBox Shape::bbox() const {
    switch (kind) {
        case PBRS_SHAPE_SPHERE: return sphere_bbox(sphere);
        case PBRS_SHAPE_QUAD: return quad_bbox(quad);
        case PBRS_SHAPE_CUBOID: return bbox_new(cuboid.min, cuboid.max);                       // :339-341
        case PBRS_SHAPE_DISK: return disk_bbox(disk);
        case PBRS_SHAPE_TRIANGLE: return bbox_union_pt(bbox_new(tri.p0, tri.p1), tri.p2);    // :422-424
        default: return mesh->bvh_root->bbox;                                                  // blas.rs:313-321
    }
}
bool Shape::intersect(const Ray& r, Interaction* out) const {
    switch (kind) {
        case PBRS_SHAPE_SPHERE: return sphere_intersect(sphere, r, out);
        case PBRS_SHAPE_QUAD: return quad_intersect(quad, r, out);
        case PBRS_SHAPE_CUBOID: return cuboid_intersect(cuboid, r, out);
        case PBRS_SHAPE_DISK: return disk_intersect(disk, r, out);
        case PBRS_SHAPE_TRIANGLE: {  // :425-427
            Interaction i;
            if (!intersect_triangle(tri.p0, tri.p1, tri.p2, r, &i)) return false;
            *out = with_dpdu(i, tri.p1 - tri.p0);
            return true;
        }
        default: return intersect_bvh(*mesh, mesh->bvh_root.get(), r, out);  // blas.rs:295-306
    }
}
bool Shape::occludes(const Ray& r) const {
    switch (kind) {
        case PBRS_SHAPE_SPHERE: return sphere_occludes(sphere, r);
        case PBRS_SHAPE_QUAD: return quad_occludes(quad, r);
        case PBRS_SHAPE_CUBOID: return cuboid_occludes(cuboid, r);
        case PBRS_SHAPE_DISK: return disk_occludes(disk, r);
        case PBRS_SHAPE_TRIANGLE: return intersect_triangle_pred(tri.p0, tri.p1, tri.p2, r);  // :428-430
        default: return intersect_bvh_pred(*mesh, mesh->bvh_root.get(), r);                   // blas.rs:307-312
    }
}

static Vec3 p3(const float* p) { return Vec3{p[0], p[1], p[2]}; }
Shape shape_from_spec(const pbrs_shape_spec& s, const std::vector<std::shared_ptr<TriangleMesh>>& meshes) {
    Shape sh{};
    sh.kind = s.kind;
    switch (s.kind) {
        case PBRS_SHAPE_SPHERE: sh.sphere = Sphere{p3(s.p), s.p[3]}; break;
        case PBRS_SHAPE_QUAD: sh.quad = ParallelQuad{p3(s.p), p3(s.p + 3), p3(s.p + 6)}; break;
        case PBRS_SHAPE_CUBOID: {  // Cuboid::from_points, simple.rs:173-182
            Vec3 a = p3(s.p), b = p3(s.p + 3);
            sh.cuboid.min = Vec3{a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z};
            sh.cuboid.max = Vec3{a.x < b.x ? b.x : a.x, a.y < b.y ? b.y : a.y, a.z < b.z ? b.z : a.z};
            break;
        }
        case PBRS_SHAPE_DISK: sh.disk = Disk{p3(s.p), hat(p3(s.p + 3)), p3(s.p + 6)}; break;  // Disk::new :42-51
        case PBRS_SHAPE_TRIANGLE: sh.tri = IsolatedTriangle{p3(s.p), p3(s.p + 3), p3(s.p + 6)}; break;
        default: sh.mesh = meshes.at(s.mesh); break;
    }
    return sh;
}

// ---- tlas/src/instance.rs + geometry/src/transform.rs:267-320 ----------------------------------------
static Vec3 xf_vec(const Mat4& m, Vec3 x) {  // transform.rs:267-272: forward * (x,0)
    Vec4 r = mul(m, Vec4{x.x, x.y, x.z, 0.0f});
    return {r.x, r.y, r.z};
}
static Point3 xf_point(const Mat4& m, Point3 p) {  // transform.rs:273-281
    Vec4 v4 = mul(m, Vec4{p.x, p.y, p.z, 1.0f});
    REF_ASSERT(v4.w == 1.0f);
    if (v4.w == 1.0f) return {v4.x, v4.y, v4.z};  // hcm.rs:340-352 (TryFrom<Vec4>)
    return {v4.x / v4.w, v4.y / v4.w, v4.z / v4.w};
}
static Ray xf_ray(const Mat4& m, const Ray& r) {  // transform.rs:282-286
    return Ray{xf_point(m, r.origin), xf_vec(m, r.dir), r.t_max};
}
static BBox xf_bbox(const Mat4& fwd, const BBox& b) {  // transform.rs:287-308
    Vec3 bases[3] = {Vec3{fwd.cols[0].x, fwd.cols[0].y, fwd.cols[0].z}, Vec3{fwd.cols[1].x, fwd.cols[1].y, fwd.cols[1].z},
                     Vec3{fwd.cols[2].x, fwd.cols[2].y, fwd.cols[2].z}};
    BBox res = bbox_empty();
    Vec3 diag = bbox_diag(b);
    for (int i = 0; i < 8; ++i) {
        Point3 corner = xf_point(fwd, b.min);
        if (i & 1) corner = corner + diag[0] * bases[0];
        if (i & 2) corner = corner + diag[1] * bases[1];
        if (i & 4) corner = corner + diag[2] * bases[2];
        res = bbox_union_pt(res, corner);
    }
    return res;
}
static Interaction xf_interaction(const Mat4& fwd, const Mat4& inv, const Interaction& i) {  // transform.rs:309-320
    REF_ASSERT(!has_nan(i.pos));
    Point3 new_pos = xf_point(fwd, i.pos);
    Vec3 new_wo = xf_vec(fwd, i.wo);
    Vec3 new_normal = mul_vec3(transpose(inv), i.normal);
    // Interaction::new's n·wo assert: already relaxed for spheres (D4); a rigid transform keeps the sign.
    Interaction res;
    res.pos = new_pos;
    res.ray_t = i.ray_t;
    res.u = i.u;
    res.v = i.v;
    res.normal = new_normal;
    res.wo = new_wo;
    res.prim = i.prim;
    res.b1 = i.b1;
    res.b2 = i.b2;
    res.tbn = Mat3{{{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}};
    res = with_dpdu(res, xf_vec(fwd, tangent(i)));
    REF_ASSERT(has_valid_frame(res));
    return res;
}

BBox Instance::bbox() const { return xf_bbox(forward, shape->bbox()); }  // instance.rs:47-49
bool Instance::intersect(const Ray& ray, Interaction* out) const {      // instance.rs:50-67
    REF_COUNT(instances);
    Ray inv_ray = xf_ray(inverse, ray);
    REF_ASSERT(norm_squared(inv_ray.dir) > 1e-3f);
    Interaction hit;
    if (!shape->intersect(inv_ray, &hit)) return false;
    REF_ASSERT(!has_nan(hit.pos));
    REF_ASSERT(has_valid_frame(hit));
    REF_COUNT(instance_hits);
    *out = xf_interaction(forward, inverse, hit);
    return true;
}
bool Instance::occludes(const Ray& ray) const {  // instance.rs:68-72
    REF_COUNT(instances);
    Ray inv_ray = xf_ray(inverse, ray);
    REF_ASSERT(norm_squared(inv_ray.dir) > 1e-6f);
    return shape->occludes(inv_ray);
}

// ---- tlas/src/bvh.rs ---------------------------------------------------------------------------------
bool BvhNode::intersect(Ray& ray, Hit* out) const {  // :77-103
    REF_COUNT(tlas_nodes);
    if (!bbox_intersect(bbox, ray)) return false;
    if (leaf) {
        Interaction i;
        if (!leaf->intersect(ray, &i)) return false;
        out->isect = i;
        out->inst = leaf.get();
        return true;
    }
    Hit l, r;
    bool has_l = child[0]->intersect(ray, &l);
    if (has_l) ray.t_max = l.isect.ray_t;
    bool has_r = child[1]->intersect(ray, &r);
    if (!has_l && !has_r) return false;
    if (has_l && !has_r) {
        *out = l;
        return true;
    }
    if (!has_l && has_r) {
        *out = r;
        return true;
    }
    if (l.isect.ray_t == r.isect.ray_t && g_diag) g_diag->tlas_ties++;
    *out = (l.isect.ray_t < r.isect.ray_t) ? l : r;
    return true;
}
bool BvhNode::occludes(const Ray& ray) const {  // :105-113
    REF_COUNT(tlas_nodes);
    if (!bbox_intersect(bbox, ray)) return false;
    if (leaf) return leaf->occludes(ray);
    return child[0]->occludes(ray) || child[1]->occludes(ray);
}
uint32_t BvhNode::height() const {  // :56-61
    if (leaf) return 1;
    return std::max(child[0]->height(), child[1]->height()) + 1;
}
std::unique_ptr<BvhNode> build_bvh(std::vector<std::unique_ptr<Instance>> instances) {  // :116-152
    REF_ASSERT(!instances.empty());
    auto node = std::make_unique<BvhNode>();
    if (instances.size() == 1) {
        node->leaf = std::move(instances.back());
        node->bbox = node->leaf->bbox();
        return node;
    }
    size_t num_all = instances.size();
    BBox bbox_all = bbox_empty();
    for (auto& i : instances) bbox_all = bbox_union(bbox_all, i->bbox());
    Vec3 span = bbox_diag(bbox_all);
    int max_span_axis = max_dimension(span);
    float split_plane = bbox_midpoint(bbox_all)[max_span_axis];
    std::vector<std::unique_ptr<Instance>> left, right;  // Iterator::partition keeps the order
    for (auto& inst : instances) {
        if (bbox_midpoint(inst->bbox())[max_span_axis] < split_plane)
            left.push_back(std::move(inst));
        else
            right.push_back(std::move(inst));
    }
    if (left.empty()) {
        for (size_t n = 0; n < num_all / 2; ++n) {
            left.push_back(std::move(right.back()));
            right.pop_back();
        }
    } else if (right.empty()) {
        for (size_t n = 0; n < num_all / 2; ++n) {
            right.push_back(std::move(left.back()));
            left.pop_back();
        }
    }
    REF_ASSERT(left.size() < num_all && right.size() < num_all);
    node->child[0] = build_bvh(std::move(left));
    node->child[1] = build_bvh(std::move(right));
    node->bbox = bbox_union(node->child[0]->bbox, node->child[1]->bbox);  // new_internal :50-55
    return node;
}

// ---- geometry/src/camera.rs ---------------------------------------------------------------------------
Camera camera_from_spec(const pbrs_camera_spec& s) {
    Camera cam;
    float aspect_ratio = (float)s.width / (float)s.height;  // :19-35
    float half_vertical = pn_tan(s.fov_y_rad * 0.5f);
    float half_horizontal = half_vertical * aspect_ratio;
    cam.a = Vec3{half_horizontal / (float)(s.width / 2), 0.0f, 0.0f};
    cam.b = Vec3{0.0f, -half_vertical / (float)(s.height / 2), 0.0f};
    cam.c = Vec3{-half_horizontal, half_vertical, 1.0f};
    cam.width = s.width;
    cam.height = s.height;
    Vec3 from = p3(s.from), target = p3(s.target), up = p3(s.up);  // look_at :37-44
    Vec3 forward = hat(target - from);
    Vec3 right = hat(cross(up, forward));
    up = cross(forward, right);
    cam.orientation = mat3_cols(right, up, forward);
    cam.center = from;
    return cam;
}
bool Camera::shoot_ray(uint32_t row, uint32_t col, float dx, float dy, Ray* out) const {  // :65-77
    float x = (float)col + pn_fract(dx);
    float y = (float)row + pn_fract(dy);
    Vec3 cc = orientation * c;
    Vec3 aa = orientation * a;
    Vec3 bb = orientation * b;
    if (row >= height || col >= width) return false;
    Vec3 dir = cc + aa * x + bb * y;
    *out = ray_new(center, dir);
    return true;
}

}  // namespace ref
