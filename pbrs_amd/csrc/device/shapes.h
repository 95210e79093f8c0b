// device/shapes.h — primitive intersection and two-level BVH traversal for the `extend` (closest
// hit) and `shadow` (any hit) stages.
//
// Restates, for one lane = one ray: shape/src/simple.rs (Sphere :207-288, Disk :306-332,
// ParallelQuad :120-163, Cuboid :343-415, intersect_triangle :435-495), shape/src/blas.rs
// (TriangleMesh::intersect_triangle :161-211, intersect_bvh :422-476, intersect_bvh_pred :478-495),
// tlas/src/instance.rs:50-72 and tlas/src/bvh.rs:77-113.  The reference recurses over Box-linked
// nodes; here the trees are flat pre-order arrays in HBM and the pending nodes live in a per-lane
// LDS stack laid out lane-major (stack[level][lane]: conflict-free for a wave).
#pragma once
#include "../../../include/pbrs_gpu.h"
#include "../../../include/pbrs_scene_spec.h"
#include "dmath.h"

struct pbrs_wnode;  // device/wide.h
struct DevScene {
    // Every BVH node of the scene in one array with absolute links: the TLAS at 0 (root = node 0), its leaves again at
    // flat_off when the TLAS is small (below), then the BLASes; a mesh instance's blas_root is an index into it.
    const pbrs_node* nodes;
    const pbrs_instance* inst;
    const pbrs_shape* shapes;
    const pbrs_mesh* meshes;
    const pbrs_tri_verts* tv;
    const pbrs_tri_shade* ts;
    const pbrs_material* mats;
    const pbrs_bxdf* bxdfs;
    const pbrs_area_light* alights;
    const pbrs_delta_light* dlights;
    uint32_t n_area, n_delta;
    float env[3];
    uint32_t has_env;
    uint32_t fast_slab;  // node coordinates are inside the range the division-free box test is exact for (traverse.h)
    uint32_t exact_extent;  // a ParallelQuad instance (hits outside its own box, D1) next to a mesh (hits beyond the extent it was given): the closest-hit walks take PBRS_FEAT_EXTENT
    // Small TLAS (PBRS_FLAT_TLAS_MIN..MAX instances): its leaves alone, in pre-order = the order the tree walk reaches them.
    // A box inside a box that a ray misses is missed too (each slab bound is a correctly rounded, hence monotonic,
    // function of the box coordinate), so testing the leaf boxes in this order — each against the t_max of its turn —
    // processes exactly the leaves, in exactly the order, of the reference's recursion (tlas/src/bvh.rs:84-88) without
    // visiting the inner nodes.  The wave runs those tests for its new rays together (traverse.h, FlatScan).  Only for
    // rays on the division-free box test (no NaN quotients); other rays walk the tree.  n_flat = 0 outside the range
    // (kernels without PBRS_FEAT_FLAT_TLAS do not contain the scan).
    uint32_t flat_off, n_flat;
    uint32_t features;   // PBRS_FEAT_*: what the traversal kernels must be able to do for this scene
    uint32_t refill_below;  // a wave of a traversal kernel takes new rays when fewer of its lanes than this are walking
    uint32_t refill_below_shadow;  // ... of k_shadow (any-hit walks end at the first occluder: later, larger refills)
    // texture/src/lib.rs (device/textures.h) and the environment light (scene/src/lib.rs:105-117)
    const pbrs_texture* textures;
    const float* tex_floats;
    const uint32_t* tex_words;
    uint32_t env_kind, env_texture;
    float env_scale[3];
    const pbrs_fourier_table* fourier;  // geometry/src/fourier.rs tables (device/fourier.h); their arrays are in the texture pools
    // Shading classes: materials with the same lobe signature (kinds, Fresnel forms, textured or not) share one; the device
    // copy of an instance carries its material's class in pad[0].  More than one class with lobes: the bounce queues are
    // ordered by class before k_shade (kernels.h, k_class_sort).
    uint32_t n_classes;
    // Four-wide nodes over every BLAS (device/wide.h; a mesh instance's wide root is in the device copy of its record, pad[1])
    // and the entries a lane's stack may hold in the kernels that walk them (beyond that a ray goes to the binary-walk kernels)
    const pbrs_wnode* wnodes;
    uint32_t wide_cap;
    // Scenes of a few KB (a Cornell box: 8 KB): what the walks read — every node, triangle-vertex record, instance record and analytic
    // shape — is copied into each block's LDS behind its stack rows at kernel start (PBRS_FEAT_LDS_SCENE kernels; kernels.h,
    // stage_scene): element counts, and the word offset of the copy in the block's dynamic LDS (a multiple of 4).  0 nodes: not staged.
    uint32_t lds_off_words, lds_nodes, lds_tris, lds_inst, lds_shapes;
    // PBRS_FEAT_LDS_TOP kernels: only the first lds_nodes nodes (the TLAS) are staged, and read through this pointer (nullptr in the uploaded
    // scene; the kernel points it at its block's copy): nodes[i] for i < lds_nodes comes from the LDS, every other node from DevScene::nodes
    const pbrs_node* nodes_top;
    // element counts of the arrays k_shade may stage in LDS (kernels.h, stage_shade_scene); n_area / n_delta above
    uint32_t n_inst, n_shapes, n_tris, n_mats, n_bxdfs;
};

// Scene features the traversal kernels are specialised on (pbrs_upload_scene derives them from the arrays it checks).
// Code a scene cannot reach still costs registers and issue slots on every wave, so each combination is its own
// instantiation: a mesh-only scene whose meshes all carry a PBRS_MESH_*_SHADING_OK flag runs the leanest one.
// pbrs_instance::flags bit set by pbrs_upload_scene on the device copy (not part of the ABI): the 3x3 part of `inv` is
// bit-exactly the identity, i.e. the instance is only translated (traverse.h, enter_instance)
#define PBRS_INSTANCE_TRANSLATION 0x100u
#define PBRS_FEAT_ANALYTIC 1u       // some instance is an analytic shape (sphere, disk, quad, cuboid, triangle)
#define PBRS_FEAT_SHADING_CHECK 2u  // some mesh needs the tangent check of blas.rs:193-200 evaluated per candidate hit
#define PBRS_FEAT_FLAT_TLAS 4u      // the leaf copies at DevScene::flat_off are built: rays on the division-free box test scan the TLAS leaves
#define PBRS_FEAT_ALL 7u
#define PBRS_FEAT_EXTENT 256u        // closest-hit walk only, outside the kernel tables: the TLAS extent is the reference's ray.t_max to the letter, rises included
                                    // (traverse.h, ClosestWalk::EXT): scenes with a ParallelQuad next to a mesh (pbrs_upload_scene)
#define PBRS_FEAT_LONG_WALKS 8u     // kernels only (not a property of the walks): several node steps per loop round (kernels.h)
#define PBRS_FEAT_WIDE 16u          // kernels only: the walks over four-wide nodes (device/wide.h); needs PBRS_FEAT_FLAT_TLAS
#define PBRS_FEAT_LDS_TOP 128u      // kernels only: the head of DevScene::nodes — a TLAS too large to scan — is copied into the block's LDS (scenes whose arrays do not fit as a whole)
#define PBRS_FEAT_LDS_SCENE 64u     // kernels only: the arrays the walks read are copied into the block's LDS at kernel start (scenes of a few KB; kernels.h)
#define PBRS_FEAT_FULL_STEPS 32u    // kernels only (with PBRS_FEAT_LONG_WALKS): a round's further node steps are full steps (kernels.h): scenes outside the guarded range of the division-free box test
#define PBRS_FLAT_TLAS_MIN 2u
// Largest TLAS the wave scans instead of walking (tools/tlas_probe.py, C5's scene family at 960x540, ms per 64 spp, walk vs
// scan): closest hit 4.84 / 4.74 at 20 instances, 5.10 / 5.28 at 24, 5.66 / 6.15 at 30 — the scan only filters there and every
// surviving leaf is still visited; any hit 3.33 / 2.38 at 20, 3.58 / 2.53 at 24, 3.97 / 2.86 at 30 — there the scan is the test.
// The candidate mask is one word: <= 32.
#ifndef PBRS_FLAT_TLAS_MAX
#define PBRS_FLAT_TLAS_MAX 20u         // k_extend
#endif
#ifndef PBRS_FLAT_TLAS_MAX_ANYHIT
#define PBRS_FLAT_TLAS_MAX_ANYHIT 32u  // k_shadow
#endif

// Per-lane work counters (instrumented kernel variant only; SURVEY.md §8(d) units).
struct WorkCounters {
    uint32_t tlas_nodes, blas_nodes, instances, instance_hits, triangles, tri_shading, spheres, quads, cuboids, disks;
};
template <bool STATS>
struct Cnt {
    WorkCounters c;
    PD void init() {
        if (STATS) c = WorkCounters{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    }
};
#define CNT(field)             \
    do {                       \
        if (STATS) cnt.c.field++; \
    } while (0)

// Interaction::with_dpdu (geometry/src/interaction.rs:45-61): returns tbn.cols[0].
PD f3 with_dpdu(f3 normal, f3 dpdu) {
    f3 n = hat(normal);
    f3 bitangent = hat(cross(n, dpdu));
    return cross(bitangent, n);
}

// What shade needs of geometry/src/interaction.rs:12-20.  uv is read by non-Solid textures only: kernels that never
// evaluate one leave it dead and the compiler drops its computation.
struct Isect {
    f3 pos, normal, wo, tangent;
    float u, v;
};

// ---- Sphere (shape/src/simple.rs:207-288) ----------------------------------------------------------------
PD bool sphere_roots(f3 center, float radius, f3 o, f3 d, float& t0, float& t1) {
    f3 f = o - center;
    float a = norm2(d);
    float b_prime = -dot(f, d);
    float delta = radius * radius - norm2(f + b_prime / a * d);
    if (delta < 0.0f) return false;
    float c = norm2(f) - radius * radius;
    float q = b_prime + pn_signum(b_prime) * pn_sqrt(delta * a);
    t0 = c / q;
    t1 = q / a;
    return true;
}
PD bool sphere_hit_t(f3 center, float radius, f3 o, f3 d, float t_max, float& t) {
    float t0, t1;
    if (!sphere_roots(center, radius, o, d, t0, t1)) return false;
    float t_low = t0 < t1 ? t0 : t1;
    float t_high = t0 < t1 ? t1 : t0;
    if (truncated_t(t_low, t_max)) {
        t = t_low;
        return true;
    }
    if (truncated_t(t_high, t_max)) {
        t = t_high;
        return true;
    }
    return false;
}
PD bool sphere_occludes(f3 center, float radius, f3 o, f3 d, float t_max) {  // :268-288 (Q13)
    float t0, t1;
    if (!sphere_roots(center, radius, o, d, t0, t1)) return false;
    return truncated_t(t0, t_max) && truncated_t(t1, t_max);
}
PD Isect sphere_isect(f3 center, float radius, f3 o, f3 d, float t) {  // :241-266
    f3 pos = o + t * d;
    f3 normal = hat(pos - center);
    pos = center + normal * radius * 1.00001f;
    f3 dpdu;
    if (!try_hat(mk3(-normal.y, normal.x, 0.0f), dpdu)) dpdu = mk3(1.0f, 0.0f, 0.0f);
    Isect i;
    i.pos = pos;
    i.normal = normal;  // always outward: D4, interior hits keep n·wo < 0
    i.wo = -d;
    i.tangent = with_dpdu(normal, dpdu);
    i.u = (pn_atan2(normal.z, normal.x) + PN_PI) / (2.0f * PN_PI);  // :248-250
    i.v = pn_acos(normal.y) / PN_PI;
    return i;
}

// ---- Disk (:306-332) ------------------------------------------------------------------------------------
PD bool disk_hit_t(f3 center, f3 normal, f3 radial, f3 o, f3 d, float t_max, float& t) {
    float tt = dot(center - o, normal) / dot(d, normal);
    if (!truncated_t(tt, t_max)) return false;
    f3 p = o + tt * d;
    if (!(norm2(p - center) <= norm2(radial))) return false;
    t = tt;
    return true;
}
PD bool disk_occludes(f3 center, f3 normal, f3 radial, f3 o, f3 d) {  // Q14: no t range
    float tt = dot(center - o, normal) / dot(d, normal);
    f3 p = o + tt * d;
    return norm2(p - center) <= norm2(radial);
}
PD Isect disk_isect(f3 center, f3 dn, f3 radial, f3 o, f3 d, float t) {
    f3 p = o + t * d;
    f3 cp = p - center;
    cp = cp - dot(cp, dn) * dn;
    f3 normal = dn * pn_signum(dot(dn, -d));
    f3 tan = hat(cross(normal, cp));
    Isect i;
    i.pos = center + cp;
    i.normal = normal;
    i.wo = -d;
    i.tangent = with_dpdu(normal, tan);
    i.u = pn_fract(pn_atan2(dot(cross(radial, cp), normal), dot(radial, cp)) * PN_FRAC_1_PI + 1.0f);  // :319-321
    i.v = norm(cp) / norm(radial);
    return i;
}

// ---- ParallelQuad (:120-163; defects D1/D2 kept) ---------------------------------------------------------
PD bool quad_hit(f3 origin, f3 su, f3 sv, f3 o, f3 d, float t_max, float& t, float& u, float& v, f3& nrm) {
    f3 normal = facing(cross(su, sv), d);
    float tt = dot(origin - o, normal) / dot(d, normal);
    if (!truncated_t(tt, t_max)) return false;
    f3 coarse = o + tt * d;
    f3 dd = coarse - origin;
    float vv = norm(cross(su, dd)) / norm(cross(su, sv));
    float uu = norm(cross(sv, dd)) / norm(cross(sv, su));
    if (!((0.0f <= vv && vv <= 1.0f) && (0.0f <= uu && uu <= 1.0f))) return false;
    t = tt;
    u = uu;
    v = vv;
    nrm = normal;
    return true;
}
PD bool quad_occludes(f3 origin, f3 su, f3 sv, f3 o, f3 d, float t_max) {  // D2: reciprocal t
    f3 normal = cross(su, sv);
    float tt = dot(d, normal) / dot(origin - o, normal);
    if (!truncated_t(tt, t_max)) return false;
    f3 coarse = o + tt * d;
    f3 dd = coarse - origin;
    float vv = norm(cross(su, dd)) / norm(cross(su, sv));
    float uu = norm(cross(sv, dd)) / norm(cross(sv, su));
    return (0.0f <= vv && vv <= 1.0f) && (0.0f <= uu && uu <= 1.0f);
}
PD Isect quad_isect(f3 origin, f3 su, f3 sv, f3 o, f3 d) {
    float t, u, v;
    f3 normal;
    quad_hit(origin, su, sv, o, d, pn_inf(), t, u, v, normal);
    Isect i;
    i.pos = origin + u * su + sv * v;
    i.normal = hat(normal);
    i.wo = -d;
    i.tangent = with_dpdu(i.normal, su);
    i.u = u;
    i.v = v;
    return i;
}

// ---- Cuboid (:343-415) ------------------------------------------------------------------------------------
PD bool cuboid_hit(f3 bmin, f3 bmax, f3 o, f3 d, float t_max, float& t, int& axis_out, float& bound_out) {
    float min_t = 0.0f, min_bound = pn_inf();
    int min_axis = 0;
    float max_t = t_max, max_bound = -pn_inf();
    int max_axis = 0;
#pragma unroll
    for (int axis = 0; axis < 3; ++axis) {
        float inv_dir = 1.0f / comp(d, axis);
        float t0 = (comp(bmin, axis) - comp(o, axis)) * inv_dir;
        float t1 = (comp(bmax, axis) - comp(o, axis)) * inv_dir;
        float b0 = comp(bmin, axis), b1 = comp(bmax, axis);
        if (t0 > t1) {
            float tmp = t0; t0 = t1; t1 = tmp;
            tmp = b0; b0 = b1; b1 = tmp;
        }
        if (t0 > min_t) {
            min_t = t0; min_bound = b0; min_axis = axis;
        }
        if (t1 < max_t) {
            max_t = t1; max_bound = b1; max_axis = axis;
        }
        if (max_t < min_t) return false;
    }
    float lo = min_t < max_t ? min_t : max_t;
    float hi = min_t < max_t ? max_t : min_t;
    bool inside = (0.0f >= lo && 0.0f <= hi);
    float ht = inside ? max_t : min_t;
    float hb = inside ? max_bound : min_bound;
    int ha = inside ? max_axis : min_axis;
    if (pn_isinf(hb)) return false;
    t = ht;
    axis_out = ha;
    bound_out = hb;
    return true;
}
PD Isect cuboid_isect(f3 bmin, f3 bmax, f3 o, f3 d) {
    float t, bound;
    int axis;
    cuboid_hit(bmin, bmax, o, d, pn_inf(), t, axis, bound);
    f3 pos = o + t * d;
    setc(pos, axis, bound);
    f3 normal = mk3(0.0f, 0.0f, 0.0f);
    setc(normal, axis, pn_signum(comp(d, axis)) * -1.0f);
    f3 tan = mk3(0.0f, 0.0f, 0.0f);
    setc(tan, (axis + 1) % 3, 1.0f);
    Isect i;
    i.pos = pos;
    i.normal = normal;
    i.wo = -d;
    i.tangent = with_dpdu(normal, tan);
    i.u = 0.5f;  // :409
    i.v = 0.5f;
    return i;
}

// ---- triangles (simple.rs:435-495) -------------------------------------------------------------------------
struct TriHit {
    float t, b0, b1, b2;
    f3 normal;
};
PD bool tri_hit(f3 p0, f3 p1, f3 p2, f3 o, f3 d, float t_max, TriHit& h) {
    f3 normal;
    if (!try_hat(cross(p0 - p1, p2 - p1), normal)) return false;
    normal = facing(normal, d);
    float t = dot(normal, p0 - o) / dot(normal, d);
    if (!truncated_t(t, t_max)) return false;
    f3 p = o + t * d;
    float b2 = dot(cross(p - p0, p - p1), normal);
    float b0 = dot(cross(p - p1, p - p2), normal);
    float b1 = dot(cross(p - p2, p - p0), normal);
    if (pn_isnan(b0) || pn_isnan(b1) || pn_isnan(b2)) return false;
    bool s0 = b0 > 0.0f, s1 = b1 > 0.0f, s2 = b2 > 0.0f;
    if (!((s0 && s1 && s2) || (!s0 && !s1 && !s2))) return false;
    float total_area = b0 + b1 + b2;
    b0 = b0 / total_area;
    b1 = b1 / total_area;
    b2 = b2 / total_area;
    f3 hit_pos = bary_lerp(p0, p1, p2, b0, b1);
    if (has_nan3(hit_pos)) return false;
    h.t = t;
    h.b0 = b0;
    h.b1 = b1;
    h.b2 = b2;
    h.normal = normal;
    return true;
}
PD bool tri_pred(f3 p0, f3 p1, f3 p2, f3 o, f3 d, float t_max) {
    f3 normal;
    if (!try_hat(cross(p0 - p1, p2 - p1), normal)) return false;
    float t = dot(normal, p0 - o) / dot(normal, d);
    if (!truncated_t(t, t_max)) return false;
    f3 p = o + t * d;
    float b0 = dot(cross(p - p0, p - p1), normal);
    float b1 = dot(cross(p - p1, p - p2), normal);
    float b2 = dot(cross(p - p2, p - p0), normal);
    bool s0 = b0 > 0.0f, s1 = b1 > 0.0f, s2 = b2 > 0.0f;
    return (s0 && s1 && s2) || (!s0 && !s1 && !s2);
}
// TriangleMesh::intersect_triangle past the geometric test (blas.rs:166-206): shading normal, tangent,
// and the Q22 rejection `abs(dpdu·n) >= 1e-3`.
PD bool mesh_tri_shading(const pbrs_tri_verts& tv, const pbrs_tri_shade& ts, f3 d, const TriHit& h, f3& normal_out, f3& dpdu_out) {
    f3 p0 = ld3(tv.p0), p1 = ld3(tv.p1), p2 = ld3(tv.p2);
    float hb1 = h.b1, hb2 = h.b2;  // hit.uv
    float b0 = 1.0f - hb1 - hb2, b1 = hb1;
    f3 n;
    if (!try_hat(bary_lerp(ld3(ts.n0), ld3(ts.n1), ld3(ts.n2), b0, b1), n)) n = h.normal;
    n = facing(n, d);
    float u0 = ts.uv0[0], v0 = ts.uv0[1];
    float u1 = ts.uv1[0] - u0, v1 = ts.uv1[1] - v0;
    float u2 = ts.uv2[0] - u0, v2 = ts.uv2[1] - v0;
    f3 dpdu = ((p2 - p0) * v2 - (p1 - p0) * v1) / (u1 * v2 - u2 * v1);
    if (!pn_isfinite(norm2(dpdu))) dpdu = p1 - p0;
    dpdu = hat(dpdu - projected_onto(dpdu, n));
    if (pn_abs(dot(dpdu, n)) >= 1e-3f) return false;
    normal_out = n;
    dpdu_out = dpdu;
    return true;
}

struct Hit {
    float t;
    uint32_t inst, prim;
    float b1, b2;
};

PD f3 nmin(const pbrs_node& n) { return mk3(n.min[0], n.min[1], n.min[2]); }
PD f3 nmax(const pbrs_node& n) { return mk3(n.max[0], n.max[1], n.max[2]); }

// ... from the block's LDS, through pointers that say so (left to infer it, the compiler merges an LDS path and a global path of the same
// loads into flat loads behind a pointer select, which send every fetch through the texture path: measured on the wide nodes)
#define PBRS_LDS_AS __attribute__((address_space(3)))
typedef float pbrs_f4v __attribute__((ext_vector_type(4)));
PD pbrs_node load_node_lds(const PBRS_LDS_AS char* p) {
    const pbrs_f4v a = *reinterpret_cast<const PBRS_LDS_AS pbrs_f4v*>(p), b = *reinterpret_cast<const PBRS_LDS_AS pbrs_f4v*>(p + 16);
    pbrs_node n;
    n.min[0] = a.x; n.min[1] = a.y; n.min[2] = a.z; n.a = __float_as_uint(a.w);
    n.max[0] = b.x; n.max[1] = b.y; n.max[2] = b.z; n.b = __float_as_uint(b.w);
    return n;
}
// Loads one 32-byte node as two 16-byte vectors (coalescing unit of the LDS/HBM path on gfx950).
PD pbrs_node load_node(const pbrs_node* p) {
    const float4* q = reinterpret_cast<const float4*>(p);
    float4 a = q[0], b = q[1];
    pbrs_node n;
    n.min[0] = a.x; n.min[1] = a.y; n.min[2] = a.z; n.a = __float_as_uint(a.w);
    n.max[0] = b.x; n.max[1] = b.y; n.max[2] = b.z; n.b = __float_as_uint(b.w);
    return n;
}
// Node i of an array addressed with a 32-bit byte offset from a wave-uniform base: one 32-bit shift for the address where the
// pointer form takes two 64-bit instructions (pbrs_upload_scene keeps DevScene::nodes below 4 GiB)
PD pbrs_node load_node_at(const pbrs_node* base, uint32_t i) {
    const char* p = reinterpret_cast<const char*>(base) + (uint32_t)(i * (uint32_t)sizeof(pbrs_node));
    return load_node(reinterpret_cast<const pbrs_node*>(p));
}
// A walk's node fetch: in the PBRS_FEAT_LDS_TOP kernels the head of the node array (the TLAS) comes from the block's LDS
template <uint32_t FEAT>
PD pbrs_node walk_node(const DevScene& S, uint32_t i) {
    if constexpr ((FEAT & PBRS_FEAT_LDS_TOP) != 0u) {
        if (i < S.lds_nodes) return load_node_lds((const PBRS_LDS_AS char*)S.nodes_top + i * (uint32_t)sizeof(pbrs_node));
    }
    return load_node_at(S.nodes, i);
}
PD pbrs_tri_verts load_tri(const pbrs_tri_verts* p) {
    const float4* q = reinterpret_cast<const float4*>(p);
    float4 a = q[0], b = q[1], c = q[2];
    // three 16-byte loads, issued together: left to itself the compiler narrows them to what each stage of the test reads — five
    // loads (a lane's load costs the L1 a cycle whatever its width), the third vertex fetched behind the test of t, a second
    // memory latency for every triangle whose plane the ray meets
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w), "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.w));
    pbrs_tri_verts t;
    t.p0[0] = a.x; t.p0[1] = a.y; t.p0[2] = a.z; t.nx = a.w;
    t.p1[0] = b.x; t.p1[1] = b.y; t.p1[2] = b.z; t.ny = b.w;
    t.p2[0] = c.x; t.p2[1] = c.y; t.p2[2] = c.z; t.nz = c.w;
    return t;
}
// intersect_triangle / intersect_triangle_pred (simple.rs:435-495) with `(p0-p1).cross(p2-p1).try_hat()`
// taken from the flattened triangle (host-evaluated with the same operand order; NaN = try_hat was None).
//
// BARY = false (the closest-hit kernels that pass on t only: k_shade recomputes the winner's barycentrics): the three
// normalising divisions and the `hit_pos` NaN test (simple.rs:462-469, Q22) are skipped where their outcome is known.
// Past the sign test b0, b1, b2 are non-NaN and of one sign, so |b_i| <= |b0 + b1 + b2|: if that sum is finite and not
// zero the quotients lie in [0, 1], and with vertex coordinates of moderate size (`coords_ok`: DevScene::fast_slab, every
// BLAS box and hence every vertex within 2^40) `barycentric_lerp` of them is finite — the hit stands.  A zero or
// non-finite sum takes the literal path.
template <bool BARY>
PD bool mesh_tri_hit_t(const pbrs_tri_verts& tv, f3 o, f3 d, float t_max, bool coords_ok, TriHit& h) {
    if (tv.nx != tv.nx) return false;
    f3 p0 = ld3(tv.p0), p1 = ld3(tv.p1), p2 = ld3(tv.p2);
    f3 normal = facing(mk3(tv.nx, tv.ny, tv.nz), d);
    float t = dot(normal, p0 - o) / dot(normal, d);
    if (!truncated_t(t, t_max)) return false;
    f3 p = o + t * d;
    float b2 = dot(cross(p - p0, p - p1), normal);
    float b0 = dot(cross(p - p1, p - p2), normal);
    float b1 = dot(cross(p - p2, p - p0), normal);
    if (pn_isnan(b0) || pn_isnan(b1) || pn_isnan(b2)) return false;
    bool s0 = b0 > 0.0f, s1 = b1 > 0.0f, s2 = b2 > 0.0f;
    if (!((s0 && s1 && s2) || (!s0 && !s1 && !s2))) return false;
    float total_area = b0 + b1 + b2;
    h.t = t;
    h.normal = normal;
    h.b0 = h.b1 = h.b2 = 0.0f;
    if (!BARY && coords_ok && pn_isfinite(total_area) && total_area != 0.0f) return true;
    b0 = b0 / total_area;
    b1 = b1 / total_area;
    b2 = b2 / total_area;
    f3 hit_pos = bary_lerp(p0, p1, p2, b0, b1);
    if (has_nan3(hit_pos)) return false;
    h.b0 = b0;
    h.b1 = b1;
    h.b2 = b2;
    return true;
}
PD bool mesh_tri_hit(const pbrs_tri_verts& tv, f3 o, f3 d, float t_max, TriHit& h) { return mesh_tri_hit_t<true>(tv, o, d, t_max, false, h); }
PD bool mesh_tri_pred(const pbrs_tri_verts& tv, f3 o, f3 d, float t_max) {
    if (tv.nx != tv.nx) return false;
    f3 p0 = ld3(tv.p0), p1 = ld3(tv.p1), p2 = ld3(tv.p2);
    f3 normal = mk3(tv.nx, tv.ny, tv.nz);
    float t = dot(normal, p0 - o) / dot(normal, d);
    if (!truncated_t(t, t_max)) return false;
    f3 p = o + t * d;
    float b0 = dot(cross(p - p0, p - p1), normal);
    float b1 = dot(cross(p - p1, p - p2), normal);
    float b2 = dot(cross(p - p2, p - p0), normal);
    bool s0 = b0 > 0.0f, s1 = b1 > 0.0f, s2 = b2 > 0.0f;
    return (s0 && s1 && s2) || (!s0 && !s1 && !s2);
}

// A per-lane stack in LDS: entry `level` of a lane at base[level * 256] (lane-major rows of the 256-thread block:
// a wave's 64 accesses to one level hit 64 consecutive dwords — no bank conflicts; the constant stride folds the
// index arithmetic into the ds_read/ds_write address).
#define PBRS_TRAVERSAL_BLOCK 256
// The traversal kernels are persistent: at most this many blocks, each lane keeps pulling rays (kernels.h).
#define PBRS_PERSISTENT_BLOCKS 1536
#define PBRS_TRAVERSAL_LANES (PBRS_PERSISTENT_BLOCKS * PBRS_TRAVERSAL_BLOCK)
struct LaneStack {
    uint32_t* base;    // &lds[threadIdx.x]
    // where the lane's own ray can be read again: origin.xyz at ro[item], direction.xyz at rd[item] (traverse.h, reload_world)
    const float4* ro;
    const float4* rd;
    uint32_t item;
    PD void put(int level, uint32_t v) { base[level * PBRS_TRAVERSAL_BLOCK] = v; }
    PD uint32_t get(int level) const { return base[level * PBRS_TRAVERSAL_BLOCK]; }
};

// Rebuilds the reference's world-space Interaction for the winning primitive: the object-space
// Interaction of the shape, then AffineTransform::apply (geometry/src/transform.rs:309-320).
PD Isect reconstruct_isect(const DevScene& S, const Hit& h, f3 o, f3 d) {
    const pbrs_instance& in = S.inst[h.inst];
    f3 oo = xf_apply(in.inv, o, 1.0f);
    f3 od = xf_apply(in.inv, d, 0.0f);
    Isect li;
    switch (in.shape_kind) {
        case PBRS_SHAPE_SPHERE: {
            const float* p = S.shapes[in.shape_index].p;
            li = sphere_isect(ld3(p), p[3], oo, od, h.t);
            break;
        }
        case PBRS_SHAPE_QUAD: {
            const float* p = S.shapes[in.shape_index].p;
            li = quad_isect(ld3(p), ld3(p + 3), ld3(p + 6), oo, od);
            break;
        }
        case PBRS_SHAPE_CUBOID: {
            const float* p = S.shapes[in.shape_index].p;
            li = cuboid_isect(ld3(p), ld3(p + 3), oo, od);
            break;
        }
        case PBRS_SHAPE_DISK: {
            const float* p = S.shapes[in.shape_index].p;
            li = disk_isect(ld3(p), ld3(p + 3), ld3(p + 6), oo, od, h.t);
            break;
        }
        case PBRS_SHAPE_TRIANGLE: {  // simple.rs:425-427
            const float* p = S.shapes[in.shape_index].p;
            f3 p0 = ld3(p), p1 = ld3(p + 3), p2 = ld3(p + 6);
            TriHit th;
            tri_hit(p0, p1, p2, oo, od, pn_inf(), th);
            li.pos = bary_lerp(p0, p1, p2, th.b0, th.b1);
            li.normal = th.normal;
            li.wo = -od;
            li.tangent = with_dpdu(th.normal, p1 - p0);
            li.u = th.b1;  // simple.rs:470-474: uv = (b1, b2)
            li.v = th.b2;
            break;
        }
        default: {
            pbrs_tri_verts tv = load_tri(S.tv + h.prim);
            f3 p0 = ld3(tv.p0), p1 = ld3(tv.p1), p2 = ld3(tv.p2);
            TriHit th;
            mesh_tri_hit(tv, oo, od, pn_inf(), th);
            f3 n, dpdu;
            mesh_tri_shading(tv, S.ts[h.prim], od, th, n, dpdu);
            li.pos = bary_lerp(p0, p1, p2, th.b0, th.b1);
            li.normal = n;
            li.wo = -od;
            li.tangent = with_dpdu(n, dpdu);
            {  // blas.rs:176-177: uv = barycentric_lerp of the vertex uvs (float.rs:37-50, scalar form)
                const pbrs_tri_shade& ts = S.ts[h.prim];
                const float b0 = 1.0f - th.b1 - th.b2, b1 = th.b1;  // blas.rs:166: (1 - u - v, u, v) of the geometric hit
                li.u = (ts.uv0[0] - ts.uv2[0]) * b0 + (ts.uv1[0] - ts.uv2[0]) * b1 + ts.uv2[0];
                li.v = (ts.uv0[1] - ts.uv2[1]) * b0 + (ts.uv1[1] - ts.uv2[1]) * b1 + ts.uv2[1];
            }
            break;
        }
    }
    Isect w;
    w.pos = xf_apply(in.fwd, li.pos, 1.0f);
    w.wo = xf_apply(in.fwd, li.wo, 0.0f);
    w.normal = xf_normal(in.inv, li.normal);
    w.tangent = with_dpdu(w.normal, xf_apply(in.fwd, li.tangent, 0.0f));
    w.u = li.u;  // transform.rs:309-320 keeps uv
    w.v = li.v;
    return w;
}
