// device/wide.h — four-wide BVH nodes for the BLAS walk of k_shadow (round 3; the closest-hit walk over them: device/experimental/).
//
// What the reference's traversal DEFINES, for a ray on the division-free box test (no NaN quotients, traverse.h), is small:
//   * closest hit in a mesh (shape/src/blas.rs:422-476): the leaves are taken in the order of the near-first depth-first walk
//     (the child the ray enters first along the split axis, `ray.dir[axis] > 0`); a leaf's triangles are tested iff the leaf's
//     OWN box passes the reference's test against the best hit of that moment (after the root: the root is tested against the
//     incoming extent).  Every slab bound is a correctly rounded — monotone — function of the box coordinate and a child's box
//     lies inside its parent's, so a leaf's box passing implies that all its ancestors passed earlier, against larger extents:
//     the tests of the inner nodes only prune, they decide nothing.
//   * any hit (:478-495): occluded iff some leaf whose own box passes (fixed extent) holds a triangle that passes; order free.
// So inner nodes may be pruned with ANY test that never rejects a box the reference's test accepts, in ANY grouping.  The walks
// below take the reference's binary tree two levels at a time: a wide node holds the boxes of up to four grandchildren (or a
// child, where it is a leaf) of an inner node X, in left-first order, tested together with pn_slab_filter (include/pbrs_numeric.h:
// f32 products by the rounded reciprocals, bounds widened by 2^-21 — proved and tested never to reject what the reference
// accepts); the children that pass are visited in the reference's order (three split axes per wide node: X's and its two
// children's).  A leaf that comes up gets the reference's exact test (slab_rs) at ITS turn, with the best hit
// of that moment, before its triangles are tested (closest hit), or once one of its triangles has passed (any hit).  The
// sequence of (leaf, extent) pairs whose triangles are tested — hence every hit, every tie-break and the final result — is the
// reference's.  Measured on C4's terrain (tools/trav_stats): 54 box tests in 54 dependent steps per bounce ray become 54
// tests in 14 steps + 2.4 exact leaf tests; shadow rays 61 -> 16 steps.
//
// Rays outside the guarded range (a zero / denormal / huge direction or origin component, in the world or inside an instance)
// never take this path: the wide kernels hand them to the binary-walk kernels through a list (kernels.h, slow list).
#pragma once
#include "shapes.h"  // included by traverse.h once RaySpace exists

// 128 bytes = one L2 line.  Planes as structure-of-arrays over the four slots so that a lane reads the planes its ray meets
// first / last on each axis as one 16-byte vector each (the choice follows the sign of the direction: a per-lane byte offset).
struct pbrs_wnode {
    float lo[3][4];     // [axis][slot]: min planes   (bytes   0 ..  47)
    float hi[3][4];     // [axis][slot]: max planes   (bytes  48 ..  95)
    uint32_t child[4];  // PBRS_WREF_LEAF | index of the reference's leaf node in DevScene::nodes; else index of a wide node; PBRS_WREF_NONE
                        // in a slot not in use (its box is inverted — lo = 2^60, hi = -2^60 — and fails the filter for every ray of the
                        // guarded range, without an overflow).  Slots 0 and 2 are always in use and carry the three split axes above
                        // the index (PBRS_WREF_AXIS_SHIFT): child[0] bits 27-28 X's, bits 29-30 its left child's; child[2] bits 27-28 its
                        // right child's — a node step reads seven vectors, not eight (a load whose lanes name different lines costs
                        // the L1 a cycle per lane whatever its width: C4 k_shadow 151 accesses per ray)
    uint32_t pad[4];
};
#ifndef PBRS_WIDE_STACK_MAX
#define PBRS_WIDE_STACK_MAX 16  // LDS stack entries per lane of the wide-walk kernels (C4's terrain: 12 at most over a frame's rays)
#endif
#ifndef PBRS_WIDE_MIN_LEVELS
#define PBRS_WIDE_MIN_LEVELS 4u  // scenes whose deepest BLAS has fewer wide levels keep the binary-walk kernels
#endif
#define PBRS_WREF_LEAF 0x80000000u
#define PBRS_WREF_NONE 0xffffffffu
#define PBRS_WREF_INDEX 0x07ffffffu  // a leaf's index in DevScene::nodes (below 2^27: the node array is addressed with 32-bit byte offsets); a wide
#define PBRS_WREF_AXIS_SHIFT 27      // node's index loses the bits above it in `index * sizeof(pbrs_wnode)` (below 2^25 for the same reason)
#define PBRS_WIDE_UNUSED_PLANE 1152921504606846976.0f /* 2^60 */
// slots 0, 1: the children of X's left child (or that child itself in slot 0, where it is a leaf); slots 2, 3: of its right child

struct WideRay {
    f3 r32;       // RN(1 / d) per component: the correctly rounded reciprocals of the direction (-RaySpace::nr)
    uint32_t nb;  // byte offsets of the planes met first on each axis, one per byte: axis * 16 (+ 48 where the direction is negative)
    PD void set(const RaySpace& C) {
        r32 = -C.nr;
        nb = (C.d.x > 0.0f ? 0u : 48u) | (C.d.y > 0.0f ? 16u : 64u) << 8 | (C.d.z > 0.0f ? 32u : 80u) << 16;
    }
    PD uint32_t nx() const { return nb & 0xffu; }
    PD uint32_t ny() const { return (nb >> 8) & 0xffu; }
    PD uint32_t nz() const { return nb >> 16; }
};

struct WideTest {
    uint32_t pass;  // bit s: slot s passed the filter
    uint32_t child[4];  // as stored: axis bits included
    PD int axis_x() const { return (int)((child[0] >> PBRS_WREF_AXIS_SHIFT) & 3u); }
    PD int axis_left() const { return (int)((child[0] >> (PBRS_WREF_AXIS_SHIFT + 2)) & 3u); }
    PD int axis_right() const { return (int)((child[2] >> PBRS_WREF_AXIS_SHIFT) & 3u); }
};
// The four slots of wide node `wi` against the ray (C: origin; W: reciprocals, plane choice) within `t_max`.  One uniform base
// and 32-bit byte offsets per lane (wide nodes are indexed below 2^24: pbrs_upload_scene).
PD WideTest wide_test(const pbrs_wnode* nodes, uint32_t wi, const RaySpace& C, const WideRay& W, float t_max) {
    const char* base = reinterpret_cast<const char*>(nodes);
    const uint32_t at = wi * (uint32_t)sizeof(pbrs_wnode);
    const float4 nx = *reinterpret_cast<const float4*>(base + (at + W.nx())), ny = *reinterpret_cast<const float4*>(base + (at + W.ny())),
                 nz = *reinterpret_cast<const float4*>(base + (at + W.nz()));
    const float4 fx = *reinterpret_cast<const float4*>(base + (at + (48u - W.nx()))), fy = *reinterpret_cast<const float4*>(base + (at + (80u - W.ny()))),
                 fz = *reinterpret_cast<const float4*>(base + (at + (112u - W.nz())));
    const uint4 ch = *reinterpret_cast<const uint4*>(base + (at + 96u));
    WideTest t;
    t.child[0] = ch.x, t.child[1] = ch.y, t.child[2] = ch.z, t.child[3] = ch.w;
    // (scalar operations: the packed forms v_pk_add_f32 / v_pk_mul_f32 for two slots at a time were measured slower — C4 k_shadow
    // 236 -> 292 ms per frame, the register allocator spilling the ray's origin around them)
    const uint32_t p0 = pn_slab_filter(nx.x, ny.x, nz.x, fx.x, fy.x, fz.x, C.o.x, C.o.y, C.o.z, W.r32.x, W.r32.y, W.r32.z, t_max) ? 1u : 0u;
    const uint32_t p1 = pn_slab_filter(nx.y, ny.y, nz.y, fx.y, fy.y, fz.y, C.o.x, C.o.y, C.o.z, W.r32.x, W.r32.y, W.r32.z, t_max) ? 2u : 0u;
    const uint32_t p2 = pn_slab_filter(nx.z, ny.z, nz.z, fx.z, fy.z, fz.z, C.o.x, C.o.y, C.o.z, W.r32.x, W.r32.y, W.r32.z, t_max) ? 4u : 0u;
    const uint32_t p3 = pn_slab_filter(nx.w, ny.w, nz.w, fx.w, fy.w, fz.w, C.o.x, C.o.y, C.o.z, W.r32.x, W.r32.y, W.r32.z, t_max) ? 8u : 0u;
    t.pass = p0 | p1 | p2 | p3;  // (a slot not in use never passes: its box is inverted)
    // the links are wanted here, with the planes: left to itself the compiler sinks their load behind the test of `pass` — a second
    // memory latency on the dependent chain of every node step that goes on (C4 k_shadow 219 -> 228 ms per frame)
    asm volatile("" : "+v"(t.child[0]), "+v"(t.child[1]), "+v"(t.child[2]), "+v"(t.child[3]));
    return t;
}

// The survivors of a node step in visiting order (r[0] first) with their pass bits, by selects only — the obvious loop over
// positions with a four-way pick compiles into nested exec-mask branches, a couple of hundred scalar and vector instructions.
struct WideOrder {
    uint32_t r[4], p[4];
};
// The reference's order (blas.rs:456-466: the left child first iff `ray.dir[axis] > 0`, at X and at each of its children)
PD WideOrder wide_order(const WideTest& t, f3 d) {
    const bool sx = !(comp(d, t.axis_x()) > 0.0f);         // X: its right child's side first
    const bool sa = !(comp(d, t.axis_left()) > 0.0f);  // within X's left child
    const bool sb = !(comp(d, t.axis_right()) > 0.0f);  // within X's right child
    const uint32_t q0 = t.pass & 1u, q1 = (t.pass >> 1) & 1u, q2 = (t.pass >> 2) & 1u, q3 = (t.pass >> 3) & 1u;
    const uint32_t a0 = sa ? t.child[1] : t.child[0], a1 = sa ? t.child[0] : t.child[1], qa0 = sa ? q1 : q0, qa1 = sa ? q0 : q1;
    const uint32_t b0 = sb ? t.child[3] : t.child[2], b1 = sb ? t.child[2] : t.child[3], qb0 = sb ? q3 : q2, qb1 = sb ? q2 : q3;
    WideOrder o;
    o.r[0] = sx ? b0 : a0, o.r[1] = sx ? b1 : a1, o.r[2] = sx ? a0 : b0, o.r[3] = sx ? a1 : b1;
    o.p[0] = sx ? qb0 : qa0, o.p[1] = sx ? qb1 : qa1, o.p[2] = sx ? qa0 : qb0, o.p[3] = sx ? qa1 : qb1;
    return o;
}
// Any order gives an any-hit walk the same answer; the side of X the ray enters first goes first (finds an occluder sooner)
PD WideOrder wide_order_any(const WideTest& t, f3 d) {
    const bool sx = !(comp(d, t.axis_x()) > 0.0f);
    const uint32_t q0 = t.pass & 1u, q1 = (t.pass >> 1) & 1u, q2 = (t.pass >> 2) & 1u, q3 = (t.pass >> 3) & 1u;
    WideOrder o;
    o.r[0] = sx ? t.child[3] : t.child[0], o.r[1] = sx ? t.child[2] : t.child[1], o.r[2] = sx ? t.child[1] : t.child[2], o.r[3] = sx ? t.child[0] : t.child[3];
    o.p[0] = sx ? q3 : q0, o.p[1] = sx ? q2 : q1, o.p[2] = sx ? q1 : q2, o.p[3] = sx ? q0 : q3;
    return o;
}
// Pushes every survivor but the first (the stack pops them in visiting order) and returns the first; at least one survives.
PD uint32_t wide_push(const WideOrder& o, LaneStack stk, int& sp) {
    const uint32_t any01 = o.p[0] | o.p[1], any012 = any01 | o.p[2];
    if (o.p[3] & any012) stk.put(sp, o.r[3]);
    if (o.p[2] & any01) stk.put(sp + (int)o.p[3], o.r[2]);
    if (o.p[1] & o.p[0]) stk.put(sp + (int)(o.p[3] + o.p[2]), o.r[1]);
    sp += (int)(o.p[0] + o.p[1] + o.p[2] + o.p[3]) - 1;
    return o.p[0] ? o.r[0] : o.p[1] ? o.r[1] : o.p[2] ? o.r[2] : o.r[3];
}
