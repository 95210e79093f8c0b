// device/traverse.h — two-level BVH traversal for `extend` (closest hit) and `shadow` (any hit).
//
// Restates tlas/src/bvh.rs:77-113 (BvhNode::intersect / occludes), tlas/src/instance.rs:50-72 and
// shape/src/blas.rs:422-495 (intersect_bvh / intersect_bvh_pred) for one lane = one ray.
//
// MI355X shape of the loop.  The reference recurses through the TLAS and runs a second, nested loop per mesh instance.
// On a 64-wide wave nested loops make every lane wait for the slowest, so a walk is a small state machine over ONE
// pending-node stack (per lane, in LDS, lane-major) shared by the TLAS and the BLAS of the instance being visited, and it
// advances by single steps: a node (pop + box test), the primitives of the held leaf (triangles: shared out over the
// wave's lanes, TriShare), or an instance boundary (ray into / out of the instance's space).  The kernels (kernels.h, PBRS_STEP_WALK) run, each round, whichever steps their lanes are
// waiting for — lanes in different phases of their walks share the instruction stream — and refill finished lanes.
//
// Box test.  geometry/src/bvh.rs:84-99 divides six times per node; IEEE f32 division costs ~11 instructions on gfx950.  With
// nr = -RN(1 / d) — the correctly rounded reciprocal, once per ray and per instance — the quotient of a numerator n is three
// instructions on its negation nn = -n:   q0 = RN(nn nr);  e = RN(d q0 + nn), which is exact;  q = RN(e nr + q0),   and q IS
// RN(n / d), the correctly rounded quotient: for all 2^23 x 2^23 pairs of f32 significands the sequence returns the bits of the
// IEEE division (tools/microbench/div_exhaustive.hip: 70 368 744 177 664 pairs, no mismatch, 27 s on one MI355X;
// profiles/r03_div_exhaustive.log), and every operation in it commutes with scaling by powers of two and with the operands' signs
// as long as nothing overflows or underflows (a zero quotient may come out with the other sign, which no comparison of the box
// test sees).  Rounds 1 and 2 took the same quotient through f64 — cvt, v_mul_f64 by a refined v_rcp_f64, cvt: three
// instructions at 4.3 issue cycles each where these are 2.7 each (tools/microbench/issue_rates.hip), and six registers for
// the reciprocals where these take three.  Lanes whose ray leaves the guarded range (a zero / denormal / huge direction
// component, an origin component that is tiny but non-zero) take the reference's literal divisions instead;
// pbrs_upload_scene checks the node coordinates once.
#pragma once
#include "shapes.h"

#include "probes.h"

struct RaySpace {
    f3 o, d;
    f3 nr;  // -RN(1 / d) per component (rays on the division-free test; zero otherwise)
    bool fast;
};
// Branch-free (one unsigned compare per range; `&`, not `&&`): six short-circuit branches here cost more than the tests.
PD uint32_t dir_in_range(float x) {  // normal, 2^-40 <= |x| <= 2^40
    uint32_t e = (pn_bits(x) >> 23) & 0xffu;
    return (e - (127u - 40u) <= 80u) ? 1u : 0u;
}
PD uint32_t origin_in_range(float x) {  // zero, or 2^-60 <= |x| <= 2^40
    uint32_t u = pn_bits(x) & 0x7fffffffu;
    uint32_t e = u >> 23;
    return ((u == 0u) | (e - (127u - 60u) <= 100u)) ? 1u : 0u;
}
PD RaySpace make_space(f3 o, f3 d, bool scene_ok) {
    RaySpace r;
    r.o = o;
    r.d = d;
    r.fast = ((scene_ok ? 1u : 0u) & dir_in_range(d.x) & dir_in_range(d.y) & dir_in_range(d.z) & origin_in_range(o.x) &
              origin_in_range(o.y) & origin_in_range(o.z)) != 0u;
    r.nr = gray(0.0f);
    if (r.fast) r.nr = mk3(-(1.0f / d.x), -(1.0f / d.y), -(1.0f / d.z));  // IEEE divisions: correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt)
    return r;
}
// The world-space ray of a lane that is inside an instance is not kept anywhere: when the lane comes back out it reads
// its ray record again (LaneStack::ro / rd at the lane's item — the 32 bytes its walk started from, usually still in L2)
// and rebuilds the space with the same operations, hence the same bits, as ClosestWalk::start did.
PD RaySpace reload_world(const DevScene& S, LaneStack stk) {
    const float4 a = stk.ro[stk.item], b = stk.rd[stk.item];
    return make_space(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), S.fast_slab != 0);
}
// What enter_instance did to the lane's space, for the way back out
#define PBRS_SPACE_WORLD 0u       // nothing: the instance is the identity
#define PBRS_SPACE_MOVED 1u       // origin, direction and reciprocals are the instance's: reload_world
#define PBRS_SPACE_TRANSLATED 2u  // only the origin moved and the ray stayed on the division-free test: the world ray is
                                  // the same direction, the same reciprocals and the origin of the ray record
PD void leave_instance(const DevScene& S, LaneStack stk, uint32_t moved, RaySpace& C) {
    if (moved == PBRS_SPACE_TRANSLATED && C.fast) {
        const float4 a = stk.ro[stk.item];
        C.o = mk3(a.x, a.y, a.z);  // in range: the walk started on the division-free test with it
    } else if (moved != PBRS_SPACE_WORLD) {
        C = reload_world(S, stk);
    }
}
// RN(n / d) from the NEGATED numerator nn = -n, the denominator and nr = -RN(1 / d): see the head of this file
PD float qdiv(float nn, float d, float nr) {
    const float q0 = nn * nr;
    const float e = __builtin_fmaf(d, q0, nn);
    return __builtin_fmaf(e, nr, q0);
}
// geometry/src/bvh.rs:84-99 (same min/max/NaN conventions as dmath.h::slab_test)
PD bool slab_rs(const pbrs_node& n, const RaySpace& R, float t_max) {
    if (!R.fast) return slab_test(nmin(n), nmax(n), R.o, R.d, t_max);
    float t0x = qdiv(R.o.x - n.min[0], R.d.x, R.nr.x), t0y = qdiv(R.o.y - n.min[1], R.d.y, R.nr.y), t0z = qdiv(R.o.z - n.min[2], R.d.z, R.nr.z);
    float t1x = qdiv(R.o.x - n.max[0], R.d.x, R.nr.x), t1y = qdiv(R.o.y - n.max[1], R.d.y, R.nr.y), t1z = qdiv(R.o.z - n.max[2], R.d.z, R.nr.z);
    // Inside the guarded range every quotient is finite, so the SSE / f32::max conventions of the reference reduce
    // to plain min/max (they differ only on NaN operands and on the sign of a zero, which no comparison sees):
    // v_min_f32 / v_max_f32 / v_min3 / v_max3 instead of compare+select chains.
    float lo_el = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)), __builtin_fminf(t0z, t1z));
    float hi_el = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fmaxf(t0y, t1y)), __builtin_fmaxf(t0z, t1z));
    float t_low = __builtin_fmaxf(lo_el, 0.0f);
    float t_high = __builtin_fminf(hi_el, t_max);  // minNum: a NaN extent is ignored, as f32::min does
    return t_low <= t_high;
}

// The same test with an infinite extent, for a ray on the division-free test, returning t_low = max(lo_el, 0) as well: with a
// finite extent t the reference's test is t_low <= min(hi_el, t), i.e. this result AND t_low <= t — which is how a closest-hit
// walk re-evaluates a scanned TLAS leaf at its turn without fetching the box again (ClosestWalkW, FlatScan::run_tlow).
PD bool slab_rs_tlow(const pbrs_node& n, const RaySpace& R, float& t_low) {
    float t0x = qdiv(R.o.x - n.min[0], R.d.x, R.nr.x), t0y = qdiv(R.o.y - n.min[1], R.d.y, R.nr.y), t0z = qdiv(R.o.z - n.min[2], R.d.z, R.nr.z);
    float t1x = qdiv(R.o.x - n.max[0], R.d.x, R.nr.x), t1y = qdiv(R.o.y - n.max[1], R.d.y, R.nr.y), t1z = qdiv(R.o.z - n.max[2], R.d.z, R.nr.z);
    float lo_el = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)), __builtin_fminf(t0z, t1z));
    float hi_el = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fmaxf(t0y, t1y)), __builtin_fmaxf(t0z, t1z));
    t_low = __builtin_fmaxf(lo_el, 0.0f);
    return t_low <= hi_el;
}
#include "wide.h"

// `self.transform.inverse().apply(*ray)` (tlas/src/instance.rs:51).  For an instance whose matrices are
// bit-exactly the identity the Mat4 products return the operand's own bits as long as every component
// is finite and non-zero (1*x + 0*y + 0*z + 0*w = x exactly), which is what W.fast plus a non-zero
// origin guarantee; anything else takes the literal products.
// Returns PBRS_SPACE_*: what became of the lane's space.
// What a walk reads of an instance besides its matrix, fetched as two loads issued together before the first branch on any of
// them: the compiler otherwise leaves each field where it is first read — flags, then (behind the branch on them) the BLAS root —
// one memory latency after the other on a step every ray takes.
struct InstHead {
    uint32_t flags, blas_root, mesh_flags, wide_root;
};
// load_inst_head reads bytes 112 .. 127 of the public record as one vector, the walks address nodes and wide nodes by byte offsets of
// these sizes: a field added or reordered in include/pbrs_gpu.h must not compile into out-of-bounds reads
static_assert(offsetof(pbrs_instance, flags) == 108 && offsetof(pbrs_instance, blas_root) == 112 && offsetof(pbrs_instance, mesh_flags) == 116 &&
                  offsetof(pbrs_instance, pad) == 120 && sizeof(pbrs_instance) == 128 && alignof(pbrs_instance) <= 16,
              "pbrs_instance layout (load_inst_head, wide root in pad[1], shading class in pad[0])");
static_assert(sizeof(pbrs_node) == 32 && offsetof(pbrs_node, a) == 12 && offsetof(pbrs_node, max) == 16 && offsetof(pbrs_node, b) == 28, "pbrs_node layout (load_node)");
static_assert(sizeof(pbrs_wnode) == 128 && offsetof(pbrs_wnode, hi) == 48 && offsetof(pbrs_wnode, child) == 96, "pbrs_wnode layout (wide_test)");
static_assert(sizeof(pbrs_tri_verts) == 48, "pbrs_tri_verts layout (load_tri)");
PD InstHead load_inst_head(const pbrs_instance& in) {
    uint32_t f = in.flags;
    uint4 r = *reinterpret_cast<const uint4*>(&in.blas_root);  // blas_root, mesh_flags, pad[0], pad[1] (the wide root): bytes 112 .. 127
    asm volatile("" : "+v"(f), "+v"(r.x), "+v"(r.y), "+v"(r.w));
    return {f, r.x, r.y, r.w};
}
PD uint32_t enter_instance(const DevScene& S, const pbrs_instance& in, uint32_t in_flags, RaySpace& C, bool need_slab, LaneStack stk) {
    if ((in_flags & PBRS_INSTANCE_IDENTITY) && C.fast && C.o.x != 0.0f && C.o.y != 0.0f && C.o.z != 0.0f) return PBRS_SPACE_WORLD;
    f3 oo = xf_apply(in.inv, C.o, 1.0f);
    // A pure translation (the 3x3 part of `inverse` bit-exactly the identity, flagged at upload): the Mat4 product
    // returns the direction's own bits — 1*x + 0*y + 0*z + t*0 with x finite and non-zero, which C.fast guarantees — so
    // the reciprocals stay; only the origin moves (and must stay inside the guarded range).
    if (need_slab && (in_flags & PBRS_INSTANCE_TRANSLATION) && C.fast) {
        C.o = oo;
        if (!(origin_in_range(oo.x) & origin_in_range(oo.y) & origin_in_range(oo.z))) {
            C.fast = false;
            C.nr = gray(0.0f);
        }
        return PBRS_SPACE_TRANSLATED;
    }
    f3 od = xf_apply(in.inv, C.d, 0.0f);
    if (need_slab) {
        C = make_space(oo, od, S.fast_slab != 0);
    } else {  // an IsolatedTriangle: no boxes below the instance, the reciprocals are never read
        C.o = oo;
        C.d = od;
        C.fast = false;
    }
    return PBRS_SPACE_MOVED;
}

// Closest hit, as a resumable walk (one lane = one ray; the kernel interleaves many walks per lane, see
// k_extend).  Semantics kept from the reference:
//  * TLAS: left subtree, then right, ray.t_max lowered to the left result (bvh.rs:84-88) == pop order
//    i+1 before a, box test at pop time against the current t_max.
//  * a candidate replaces the best when !(best.t < cand.t) (bvh.rs:94-98).
//  * BLAS: root tested against the incoming extent, every later node against the mesh-local best
//    only (`ray.t_max = outer_hit.ray_t`, blas.rs:468, runs after a node passed its box test);
//    children pushed far-then-near by `ray.dir[axis] > 0` (blas.rs:456-466); within a leaf the
//    triangles see the t_max from before the leaf and `new.t < outer.t` keeps the first of equals.
//  * the mesh may return a hit beyond the incoming extent (it loses at the TLAS compare).
// Known deviation: outside PBRS_FEAT_EXTENT (below) ray.t_max is never RAISED by such an overshoot; observable
// on bit-identical t from two instances (oracle counter tlas_ties) — and where a shape reports hits outside its own box
// (ParallelQuad, D1), which is why scenes with one next to a mesh walk with PBRS_FEAT_EXTENT (fuzz seed 211699, round 4).
// Walk states.  A kernel advances every lane by single steps — one node (pop + box test) or one primitive — so that
// lanes of one wave that are in different phases of their walks still share the instruction stream (k_extend).
#ifndef PBRS_EARLY_OUT  // a failed box test that leaves nothing pending below the instance goes to the boundary state at once
#define PBRS_EARLY_OUT 1
#endif
#define PBRS_WALK_IDLE 0u
#define PBRS_WALK_NODE 1u
#define PBRS_WALK_LEAF 2u
#define PBRS_WALK_DONE 3u
#define PBRS_WALK_SCAN 5u  // just started on a small TLAS: waits, within the refill, for the wave's shared scan of the leaf boxes
#define PBRS_WALK_XFER 4u  // at an instance boundary: about to enter one (TLAS leaf popped) or to leave one (its entries are used up)

// Triangle tests are the longest step of a walk (a division, three cross products, three more divisions) and at any one
// time only some lanes of a wave hold a leaf, with one to four triangles each.  Run per lane, a wave-level execution of
// the step tests one triangle for each of those lanes and leaves the rest of the wave idle, and a lane needs as many
// executions as its leaf has triangles.  Instead the wave shares the tests out: every (owner lane, k-th triangle of its
// leaf) pair, k < 4, gets a helper lane, the helper fetches the owner's ray through ds_bpermute, runs the test, and the
// owner folds its pairs' results in leaf order — the values and their order are those of the sequential loop
// (blas.rs:440-452).  Pair (owner, k) sits at helper lane (number of pairs with smaller k) + (owners with a k-th triangle
// in lower lanes): wave-uniform ballots and mbcnt.  Lane 63 is never a helper (it takes the ds_permute writes of lanes
// with nothing to send); pairs that would land on 63 or beyond wait for the next execution.
PD uint32_t lane_prefix(uint64_t mask) {  // set bits of a wave mask below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
struct TriShare {
    uint32_t tag;     // as a helper: 0x10000 | k << 8 | owner lane — whose triangle this lane tests; 0 = none
    uint32_t cnt;     // as an owner: triangles of this lane's leaf tested in this execution
    uint64_t has[4];  // wave-uniform: lanes that want a k-th triangle tested
    uint32_t off[4];  // wave-uniform: pairs with smaller k

    PD void build(uint32_t want) {
        const uint32_t lane = threadIdx.x & 63u;
        uint32_t o = 0;
        tag = 0;
        cnt = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint64_t m = __ballot(want > j);
            has[j] = m;
            off[j] = o;
            if (m == 0) continue;
            const uint32_t p = o + lane_prefix(m);
            const bool send = want > j && p < 63u;
            tag |= (uint32_t)__builtin_amdgcn_ds_permute((int)((send ? p : 63u) << 2), send ? (int)(lane | j << 8 | 0x10000u) : 0);
            cnt += send ? 1u : 0u;
            o += (uint32_t)__popcll(m);
        }
    }
    PD bool helper() const { return tag != 0u; }
    PD uint32_t k() const { return (tag >> 8) & 3u; }
    PD uint32_t pos(uint32_t j) const { return off[j] + lane_prefix(has[j]); }  // helper lane of this lane's j-th triangle, j < cnt
    PD uint32_t from_owner(uint32_t v) const { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(tag << 2), (int)v); }
    PD float from_owner(float v) const { return __uint_as_float(from_owner(__float_as_uint(v))); }
    PD f3 from_owner(f3 v) const { return mk3(from_owner(v.x), from_owner(v.y), from_owner(v.z)); }
    PD float from_helper(uint32_t at, float v) const {
        return __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute((int)(at << 2), (int)__float_as_uint(v)));
    }
};

// Shared scan of a small TLAS (the leaf copies at DevScene::flat_off).  Lanes that have just been given a ray ("fresh") need every leaf box
// of the scene tested against it; run per lane that is one node step per leaf with the other lanes of the wave looking on.
// Instead each fresh lane gets H = ceil(n_flat / 2) helper slots, a slot tests leaves h and h + H against the owner's ray
// (pulled through ds_bpermute once per slot), and the owner collects its ray's pass bits from two ballots.  The box test
// does not depend on anything the walk changes except t_max, and only closest-hit walks change that (ClosestWalk::shrunk).
struct FlatScan {
    PD static uint32_t pull(uint32_t from, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from << 2), (int)v); }
    PD static float pull(uint32_t from, float v) { return __uint_as_float(pull(from, __float_as_uint(v))); }
    // Returns, to each fresh lane, the mask of the leaves whose box its ray R (on the division-free test) passes within
    // t_max; `tested` counts the box tests this lane ran as a helper.  Every lane of the wave calls this together;
    // S.n_flat <= PBRS_FLAT_TLAS_MAX_ANYHIT = 32.
    PD static uint32_t run(const DevScene& S, bool fresh, const RaySpace& R, float t_max, uint32_t& tested) {
        const uint64_t m = __ballot(fresh);
        if (m == 0) return 0u;
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t H = (S.n_flat + 1u) >> 1;  // 4..8 slots per owner
        const uint32_t total = (uint32_t)__popcll(m) * H;
        const uint32_t rank = lane_prefix(m);
        // lane r learns which lane holds the r-th fresh ray (the others write to lane 63, which no rank reaches unless all 64 are fresh)
        const uint32_t list = (uint32_t)__builtin_amdgcn_ds_permute((int)((fresh ? rank : 63u) << 2), (int)lane);
        const uint32_t magic = (65536u + H - 1u) / H;  // p / H == p * magic >> 16 for p < 1024
        uint32_t mine = 0;
        for (uint32_t base = 0; base < total; base += 64u) {
            const uint32_t p = base + lane;
            const bool valid = p < total;
            const uint32_t r = (p * magic) >> 16, h = p - r * H;
            const uint32_t owner = pull(r, list);
            RaySpace O;
            O.o = mk3(pull(owner, R.o.x), pull(owner, R.o.y), pull(owner, R.o.z));
            O.d = mk3(pull(owner, R.d.x), pull(owner, R.d.y), pull(owner, R.d.z));
            O.nr = mk3(pull(owner, R.nr.x), pull(owner, R.nr.y), pull(owner, R.nr.z));
            O.fast = true;
            const float ot = pull(owner, t_max);
            bool pass0 = false, pass1 = false;
            if (valid) {
                pass0 = slab_rs(load_node(S.nodes + S.flat_off + h), O, ot);
                tested += 1u;
                if (h + H < S.n_flat) {
                    pass1 = slab_rs(load_node(S.nodes + S.flat_off + h + H), O, ot);
                    tested += 1u;
                }
            }
            const uint64_t w0 = __ballot(pass0), w1 = __ballot(pass1);
            const int sft = (int)(rank * H) - (int)base;  // this lane's first slot, relative to the window
            if (fresh && sft > -(int)H && sft < 64) {
                const uint64_t a = sft >= 0 ? w0 >> sft : w0 << -sft, b = sft >= 0 ? w1 >> sft : w1 << -sft;
                const uint32_t keep = (1u << H) - 1u;
                mine |= ((uint32_t)a & keep) | (((uint32_t)b & keep) << H);
            }
        }
        return mine;
    }
    // The same for a closest-hit walk, whose scan only FILTERS (an infinite extent; the reference's test of a leaf's box is made when
    // the leaf's turn comes, ClosestWalk::node_step): the conservative f32 filter of include/pbrs_numeric.h — it passes whenever the
    // exact test passes, in 57 instead of 120 issue cycles per box — and three pulled reciprocals instead of six pulled halves.
    PD static uint32_t run_filter(const DevScene& S, bool fresh, const RaySpace& R, uint32_t& tested) {
        const uint64_t m = __ballot(fresh);
        if (m == 0) return 0u;
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t H = (S.n_flat + 1u) >> 1;
        const uint32_t total = (uint32_t)__popcll(m) * H;
        const uint32_t rank = lane_prefix(m);
        const uint32_t list = (uint32_t)__builtin_amdgcn_ds_permute((int)((fresh ? rank : 63u) << 2), (int)lane);
        const uint32_t magic = (65536u + H - 1u) / H;
        const f3 r32 = -R.nr;  // RN(1 / d): what the filter is proved on
        uint32_t mine = 0;
        for (uint32_t base = 0; base < total; base += 64u) {
            const uint32_t p = base + lane;
            const bool valid = p < total;
            const uint32_t r = (p * magic) >> 16, h = p - r * H;
            const uint32_t owner = pull(r, list);
            const f3 oo = mk3(pull(owner, R.o.x), pull(owner, R.o.y), pull(owner, R.o.z));
            const f3 rr = mk3(pull(owner, r32.x), pull(owner, r32.y), pull(owner, r32.z));
            bool pass0 = false, pass1 = false;
            if (valid) {
                const pbrs_node n0 = load_node(S.nodes + S.flat_off + h);
                // the planes met first / last on an axis by the sign of the reciprocal (= the direction's)
                pass0 = pn_slab_filter(rr.x > 0.0f ? n0.min[0] : n0.max[0], rr.y > 0.0f ? n0.min[1] : n0.max[1], rr.z > 0.0f ? n0.min[2] : n0.max[2],
                                       rr.x > 0.0f ? n0.max[0] : n0.min[0], rr.y > 0.0f ? n0.max[1] : n0.min[1], rr.z > 0.0f ? n0.max[2] : n0.min[2], oo.x, oo.y,
                                       oo.z, rr.x, rr.y, rr.z, pn_inf()) != 0;
                tested += 1u;
                if (h + H < S.n_flat) {
                    const pbrs_node n1 = load_node(S.nodes + S.flat_off + h + H);
                    pass1 = pn_slab_filter(rr.x > 0.0f ? n1.min[0] : n1.max[0], rr.y > 0.0f ? n1.min[1] : n1.max[1], rr.z > 0.0f ? n1.min[2] : n1.max[2],
                                           rr.x > 0.0f ? n1.max[0] : n1.min[0], rr.y > 0.0f ? n1.max[1] : n1.min[1], rr.z > 0.0f ? n1.max[2] : n1.min[2], oo.x,
                                           oo.y, oo.z, rr.x, rr.y, rr.z, pn_inf()) != 0;
                    tested += 1u;
                }
            }
            const uint64_t w0 = __ballot(pass0), w1 = __ballot(pass1);
            const int sft = (int)(rank * H) - (int)base;
            if (fresh && sft > -(int)H && sft < 64) {
                const uint64_t a = sft >= 0 ? w0 >> sft : w0 << -sft, b = sft >= 0 ? w1 >> sft : w1 << -sft;
                const uint32_t keep = (1u << H) - 1u;
                mine |= ((uint32_t)a & keep) | (((uint32_t)b & keep) << H);
            }
        }
        return mine;
    }
};

// FlatScan::run with an infinite extent that also leaves, for every leaf that passed, the exact entry distance t_low in the
// owner's column of a block-wide LDS table (tl_block[leaf * 256 + thread of the owner]).
PD uint32_t flat_scan_tlow(const DevScene& S, bool fresh, const RaySpace& R, float* tl_block) {
    const uint64_t m = __ballot(fresh);
    if (m == 0) return 0u;
    const uint32_t lane = threadIdx.x & 63u, wave_base = threadIdx.x & ~63u;
    const uint32_t H = (S.n_flat + 1u) >> 1;
    const uint32_t total = (uint32_t)__popcll(m) * H;
    const uint32_t rank = lane_prefix(m);
    const uint32_t list = (uint32_t)__builtin_amdgcn_ds_permute((int)((fresh ? rank : 63u) << 2), (int)lane);
    const uint32_t magic = (65536u + H - 1u) / H;
    uint32_t mine = 0;
    for (uint32_t base = 0; base < total; base += 64u) {
        const uint32_t p = base + lane;
        const bool valid = p < total;
        const uint32_t r = (p * magic) >> 16, h = p - r * H;
        const uint32_t owner = FlatScan::pull(r, list);
        RaySpace O;
        O.o = mk3(FlatScan::pull(owner, R.o.x), FlatScan::pull(owner, R.o.y), FlatScan::pull(owner, R.o.z));
        O.d = mk3(FlatScan::pull(owner, R.d.x), FlatScan::pull(owner, R.d.y), FlatScan::pull(owner, R.d.z));
        O.nr = mk3(FlatScan::pull(owner, R.nr.x), FlatScan::pull(owner, R.nr.y), FlatScan::pull(owner, R.nr.z));
        O.fast = true;
        bool pass0 = false, pass1 = false;
        if (valid) {
            float t_low;
            pass0 = slab_rs_tlow(load_node(S.nodes + S.flat_off + h), O, t_low);
            if (pass0) tl_block[h * PBRS_TRAVERSAL_BLOCK + wave_base + owner] = t_low;
            if (h + H < S.n_flat) {
                pass1 = slab_rs_tlow(load_node(S.nodes + S.flat_off + h + H), O, t_low);
                if (pass1) tl_block[(h + H) * PBRS_TRAVERSAL_BLOCK + wave_base + owner] = t_low;
            }
        }
        const uint64_t w0 = __ballot(pass0), w1 = __ballot(pass1);
        const int sft = (int)(rank * H) - (int)base;
        if (fresh && sft > -(int)H && sft < 64) {
            const uint64_t a = sft >= 0 ? w0 >> sft : w0 << -sft, b = sft >= 0 ? w1 >> sft : w1 << -sft;
            const uint32_t keep = (1u << H) - 1u;
            mine |= ((uint32_t)a & keep) | (((uint32_t)b & keep) << H);
        }
    }
    return mine;
}

template <bool STATS, uint32_t FEAT>
struct ClosestWalk {
    RaySpace C;  // the space the lane is walking in (the world ray in the TLAS, the instance's ray below a TLAS leaf): one
                 // box-test call serves lanes in either tree.  The world ray is read again on the way out (reload_world).
    Hit best;       // best.t stays +inf until the first candidate: `!(best.t < t)` then accepts it, as Option::None does
    float t_max, lt, mt, mb1, mb2;  // lt: the cloned ray's t_max inside intersect_bvh; mt: outer_hit.ray_t
    uint32_t mprim, cur_inst;
    uint32_t inst_info;  // of the instance the lane is in: shape kind | mesh flags << 3 | bit 30: entered by translating the origin only
                         // | bit 31: an analytic candidate is held
    uint32_t leaf_a, leaf_end;  // held leaf: triangles [leaf_a, leaf_end) of a BLAS leaf, or the record of an IsolatedTriangle
    int sp, blas_base;
    uint32_t cand;        // leaves of the leaf copies at DevScene::flat_off still to visit: their boxes passed the shared scan (0 on a tree walk)
    bool in_blas, moved;  // moved: C is not the world ray (inst_info bit 30: only its origin differs)
    uint32_t mode;
    // PBRS_FEAT_EXTENT: t_max is the reference's ray.t_max to the letter.  bvh.rs:84-88 sets it to the LEFT subtree's result when
    // that subtree returns one — not to the best hit so far: a mesh may return a hit beyond the extent it was given (blas.rs:468), the
    // extent then RISES, and boxes the best hit would have pruned are entered.  What they hold loses at the compare (bvh.rs:94-98)
    // unless a shape reports hits outside its own box, which ParallelQuad does (D1, the mirrored quadrants): scenes with such an
    // instance next to a mesh take this walk (pbrs_upload_scene).  A pending right sibling carries, in a second stack word, the
    // smallest t returned since the window of the entry below it began; `win` is that of the innermost window.  When the sibling is
    // popped its left subtree is complete: win is the left result (set_extent), and folds into the window below.
    static constexpr bool EXT = (FEAT & PBRS_FEAT_EXTENT) != 0u;
    float win;
    bool win_has;
    PBRS_TP_FIELDS

    PD void start(const DevScene& S, f3 o, f3 d, float tmax, LaneStack stk) {
        C = make_space(o, d, S.fast_slab != 0);
        win = 0.0f;
        win_has = false;
        moved = false;
        best.t = pn_inf();
        best.inst = 0xffffffffu;
        best.prim = 0;
        best.b1 = best.b2 = 0.0f;
        in_blas = false;
        t_max = tmax;
        lt = tmax;  // lt is the extent of the tree the lane is in: t_max in the TLAS, the cloned ray's inside a mesh
        mt = pn_inf();
        mprim = cur_inst = inst_info = leaf_a = leaf_end = 0;
        mb1 = mb2 = 0.0f;
        blas_base = 0;
        cand = 0;
        if ((FEAT & PBRS_FEAT_FLAT_TLAS) && !EXT && S.n_flat != 0u && C.fast) {  // (the extent's windows follow the tree)
            sp = 0;
            mode = PBRS_WALK_SCAN;
        } else {
            stk.put(0, 0u);
            sp = 1;
            mode = PBRS_WALK_NODE;
        }
    }
    // The leaf boxes of a small TLAS against the rays that have just started (FlatScan); every lane of the wave calls this
    // together, right after start().  For a closest-hit walk the scan is a filter: the reference tests a leaf's box when
    // its recursion gets there, with the t_max of that moment, and so does node_step for the leaves that are left.  The
    // filter runs with an infinite extent — a box the ray misses at any distance — because t_max does not only come down:
    // a mesh may return a hit beyond the extent it was given (blas.rs:468) and `set_extent` then raises it (bvh.rs:84-88).
    PD void scan_wave(const DevScene& S, Cnt<STATS>& cnt) {
        if (!(FEAT & PBRS_FEAT_FLAT_TLAS) || EXT) return;
        uint32_t tested = 0;
#ifdef PBRS_EXACT_CLOSEST_SCAN
        const uint32_t mine = FlatScan::run(S, mode == PBRS_WALK_SCAN, C, pn_inf(), tested);
#else
        const uint32_t mine = FlatScan::run_filter(S, mode == PBRS_WALK_SCAN, C, tested);
#endif
        if (STATS) cnt.c.tlas_nodes += tested;
        if (mode == PBRS_WALK_SCAN) {
            cand = mine;
            mode = PBRS_WALK_NODE;
        }
    }

    // One node: pop, box test, then push the children / hold the leaf / stop at the instance boundary.
    PD void node_step(const DevScene& S, LaneStack stk, Cnt<STATS>& cnt) {
        PBRS_TP(0);
        if (in_blas && sp == blas_base) {  // the instance's entries are used up: leave it (xfer_step, or finish at retire time)
            mode = exit_mode();
            return;
        }
        // One box test serves lanes at a BLAS / TLAS node and lanes whose turn it is to test a scanned leaf of a small TLAS
        // — the test the reference makes at this moment (scan_wave only filters).
        uint32_t ni;
        if (sp == 0) {  // not inside an instance (its exit was taken above), nothing pending: the next scanned leaf, or the end
            if (!(FEAT & PBRS_FEAT_FLAT_TLAS) || cand == 0u) {
                mode = PBRS_WALK_DONE;
                return;
            }
            ni = S.flat_off + (uint32_t)__builtin_ctz(cand);
            cand &= cand - 1u;
        } else {
            ni = stk.get(--sp);
            if constexpr (EXT) {
                if (!in_blas && (ni & 0x80000000u)) {  // a right sibling: its left subtree has returned (bvh.rs:84-88)
                    const bool below_has = (ni & 0x40000000u) != 0u;
                    const float below = __uint_as_float(stk.get(--sp));
                    ni &= 0x3fffffffu;
                    if (win_has) t_max = win;  // ray.set_extent(isect.ray_t): down or UP
                    // the window of the entry below goes on: what it held before this subtree, then this subtree's result
                    // (`l.ray_t < r.ray_t ? l : r`, :94-98: the earlier one only if strictly nearer)
                    if (below_has && (!win_has || below < win)) win = below;
                    win_has = win_has || below_has;
                    lt = t_max;
                }
            }
            if (STATS) {  // the scanned leaves were counted by the scan
                if (in_blas) CNT(blas_nodes);
                else CNT(tlas_nodes);
            }
        }
        const pbrs_node node = walk_node<FEAT>(S, ni);
        PBRS_TP(1);
        if (!slab_rs(node, C, lt)) {
            PBRS_TP(2);
            if (PBRS_EARLY_OUT && in_blas && sp == blas_base) mode = exit_mode();
            return;
        }
        if (!(node.b & PBRS_LEAF_FLAG)) {
            // TLAS: left (i+1) is popped first.  BLAS: the child the ray enters first along the split axis
            // (blas.rs:456-466), and the cloned ray's t_max follows outer_hit (blas.rs:468).
            bool left_first = !in_blas || comp(C.d, (int)(node.b & 3u)) > 0.0f;
            uint32_t left = ni + 1, right = node.a;
            if constexpr (EXT) {
                if (!in_blas) {  // a new window opens for the left subtree; the one it interrupts waits with the right sibling
                    stk.put(sp++, __float_as_uint(win));
                    right |= 0x80000000u | (win_has ? 0x40000000u : 0u);
                    win_has = false;
                }
            }
            stk.put(sp++, left_first ? right : left);
            stk.put(sp++, left_first ? left : right);
            lt = in_blas ? mt : lt;
        } else if (in_blas) {
            PBRS_TP(3);
            PBRS_TP(4);
            leaf_a = node.a;
            leaf_end = node.a + (node.b & ~PBRS_LEAF_FLAG);
            if (leaf_end != leaf_a) mode = PBRS_WALK_LEAF;
            else lt = mt;  // an empty leaf still runs blas.rs:468
        } else {  // a TLAS leaf: enter the instance (xfer_step)
            cur_inst = node.a;
            inst_info = (node.b >> PBRS_TLAS_LEAF_KIND_SHIFT) & 7u;
            mode = PBRS_WALK_XFER;
        }
    }

    // A round's FURTHER node steps (kernels.h, PBRS_MORE_NODE_STEPS): the common case only — a pending entry of the tree the lane is
    // in is popped and tested.  A lane whose stack is down to the floor of its tree (the instance used up, the next scanned TLAS
    // leaf, the end of the walk) sits the step out and lets the round's first step (node_step), which has the code for all that, take it: the
    // second and third copies of the step carry a third fewer scalar instructions (exec-mask bookkeeping of branches that nearly
    // every execution took for one or two lanes).
    PD void node_step_fast(const DevScene& S, LaneStack stk, Cnt<STATS>& cnt) {
        if constexpr (EXT) return;  // (two-word TLAS entries: node_step's)
        if (sp == (in_blas ? blas_base : 0) || !C.fast) return;  // (a ray on the literal divisions: the first step's, too)
        PBRS_TP(0);
        const uint32_t ni = stk.get(--sp);
        if (STATS) {
            if (in_blas) CNT(blas_nodes);
            else CNT(tlas_nodes);
        }
        const pbrs_node node = walk_node<FEAT>(S, ni);
        PBRS_TP(1);
        RaySpace F = C;
        F.fast = true;  // known here: this copy of the box test carries no division path
        if (!slab_rs(node, F, lt)) {
            PBRS_TP(2);
            return;
        }
        if (!(node.b & PBRS_LEAF_FLAG)) {
            bool left_first = !in_blas || comp(C.d, (int)(node.b & 3u)) > 0.0f;
            uint32_t left = ni + 1, right = node.a;
            stk.put(sp++, left_first ? right : left);
            stk.put(sp++, left_first ? left : right);
            lt = in_blas ? mt : lt;
        } else if (in_blas) {
            PBRS_TP(3);
            PBRS_TP(4);
            leaf_a = node.a;
            leaf_end = node.a + (node.b & ~PBRS_LEAF_FLAG);
            if (leaf_end != leaf_a) mode = PBRS_WALK_LEAF;
            else lt = mt;  // an empty leaf still runs blas.rs:468
        } else {  // a TLAS leaf: enter the instance (xfer_step)
            cur_inst = node.a;
            inst_info = (node.b >> PBRS_TLAS_LEAF_KIND_SHIFT) & 7u;
            mode = PBRS_WALK_XFER;
        }
    }

    // Instance boundary, both directions.  Any one lane crosses a boundary in few of its steps, but some lane of a wave
    // does in nearly every round; inline, this code (ray transform, reciprocals, scratch traffic) ran for a handful of
    // lanes each round.  As a state of its own the kernel runs it when enough lanes wait at a boundary (k_extend).
    // The instance's candidate meets the best hit so far (bvh.rs:82-95 after Instance::intersect returned)
    PD void meet_best(Cnt<STATS>& cnt) {
        // a mesh candidate always has t < inf (it beat outer_hit.ray_t = inf); an analytic one may sit at t == +inf
        // (x / 0 with an infinite extent), so those are flagged
        if (mt < pn_inf() || (inst_info & 0x80000000u)) {
            CNT(instance_hits);
            if constexpr (EXT) {  // every returned hit counts for the windows, the best one or not
                if (!win_has || !(win < mt)) win = mt;
                win_has = true;
            }
            if (!(best.t < mt)) {
                best.t = mt;
                best.inst = cur_inst;
                best.prim = mprim;
                best.b1 = mb1;
                best.b2 = mb2;
                if constexpr (!EXT) t_max = mt;  // (EXT: when the subtree this leaf is in returns to a node it is the left child of)
            }
        }
    }
    // Where a lane goes when the instance it is in has nothing left: to the boundary step — unless nothing is left above
    // it either (no pending TLAS entry, no scanned leaf still to visit).  Then the walk is over: the world-space ray is
    // not needed again, and the candidate meets `best` when the lane is retired (finish), together with the other lanes
    // of the batch instead of in a boundary step of its own (one of the ≈2.3 such steps a C4 ray takes).
    PD uint32_t exit_mode() const {
        const bool nothing_above = blas_base == 0 && (!(FEAT & PBRS_FEAT_FLAT_TLAS) || cand == 0u);
        return nothing_above ? PBRS_WALK_DONE : PBRS_WALK_XFER;
    }
    PD void finish(Cnt<STATS>& cnt) {  // at retire time, every lane whose walk is DONE
        if (in_blas) {
            in_blas = false;
            meet_best(cnt);
        }
    }
    PD void xfer_step(const DevScene& S, LaneStack stk, Cnt<STATS>& cnt) {
        mode = PBRS_WALK_NODE;
        if (in_blas) {  // intersect_bvh / the shape returned (blas.rs:471-475) -> Instance::intersect -> bvh.rs:82
            PBRS_TP(7);
            in_blas = false;
            leave_instance(S, stk, !moved ? PBRS_SPACE_WORLD : (inst_info & 0x40000000u) ? PBRS_SPACE_TRANSLATED : PBRS_SPACE_MOVED, C);
            meet_best(cnt);
            lt = t_max;  // back in the TLAS
            return;
        }
        // Instance::intersect (instance.rs:50-67): the ray goes into the instance's space and stays there until the
        // walk is back at this stack level.  A mesh continues in the node state with its BLAS root, an IsolatedTriangle
        // is one held triangle record; either way the candidate (mt, ...) meets `best` at the exit above (or at retire
        // time, finish).  An analytic shape is visited here and now (analytic_visit).
        const pbrs_instance& in = S.inst[cur_inst];
        CNT(instances);
        PBRS_TP(6);
        const uint32_t kind = inst_info;
        if ((FEAT & PBRS_FEAT_ANALYTIC) && kind != PBRS_SHAPE_MESH && kind != PBRS_SHAPE_TRIANGLE) {
            analytic_visit(S, in, kind, cnt);  // back at the TLAS already
            return;
        }
        const InstHead H = load_inst_head(in);
        const uint32_t space = enter_instance(S, in, H.flags, C, kind == PBRS_SHAPE_MESH, stk);
        moved = space != PBRS_SPACE_WORLD;
        in_blas = true;
        blas_base = sp;
        lt = t_max;
        mt = pn_inf();
        if (kind == PBRS_SHAPE_MESH) {
            inst_info = kind | (H.mesh_flags << 3) | (space == PBRS_SPACE_TRANSLATED ? 0x40000000u : 0u);
            stk.put(sp++, H.blas_root);
        } else if (kind == PBRS_SHAPE_TRIANGLE) {
            // IsolatedTriangle (simple.rs:417-426): one triangle record, no boxes, no shading frame
            inst_info = kind | (PBRS_MESH_SHADING_OK_MASK << 3);
            leaf_a = H.blas_root;
            leaf_end = H.blas_root + 1u;
            mode = PBRS_WALK_LEAF;
        }
    }

    // Instance::intersect (instance.rs:50-67) for an analytic shape, which is all there is below its TLAS leaf, in one go:
    // the ray into the shape's space (literal products, as the reference makes them), the shape's test against the TLAS
    // extent, the candidate against the best hit (bvh.rs:82-95).  The lane's space stays the world's: no way in, no way
    // out, no wait for the primitive step in between (round 1 and most of round 2: three states, three waits).
    PD void analytic_visit(const DevScene& S, const pbrs_instance& in, uint32_t kind, Cnt<STATS>& cnt) {
        if (!(FEAT & PBRS_FEAT_ANALYTIC)) return;
        const f3 o = xf_apply(in.inv, C.o, 1.0f), d = xf_apply(in.inv, C.d, 0.0f);
        const float ext = t_max;  // the TLAS extent at entry
        const float* p = S.shapes[in.shape_index].p;
        float t = 0.0f, b1 = 0.0f, b2 = 0.0f;
        bool hit = false;
        switch (kind) {
            case PBRS_SHAPE_SPHERE:
                CNT(spheres);
                hit = sphere_hit_t(ld3(p), p[3], o, d, ext, t);
                break;
            case PBRS_SHAPE_QUAD: {
                CNT(quads);
                float u, v;
                f3 n;
                hit = quad_hit(ld3(p), ld3(p + 3), ld3(p + 6), o, d, ext, t, u, v, n);
                break;
            }
            case PBRS_SHAPE_CUBOID: {
                CNT(cuboids);
                int axis;
                float bound;
                hit = cuboid_hit(ld3(p), ld3(p + 3), o, d, ext, t, axis, bound);
                break;
            }
            case PBRS_SHAPE_DISK:
                CNT(disks);
                hit = disk_hit_t(ld3(p), ld3(p + 3), ld3(p + 6), o, d, ext, t);
                break;
            default:  // PBRS_SHAPE_TRIANGLE never gets here (triangle-record path above)
                break;
        }
        inst_info = kind | (hit ? 0x80000000u : 0u);
        mt = hit ? t : pn_inf();
        mprim = 0;
        mb1 = b1;
        mb2 = b2;
        meet_best(cnt);
        lt = t_max;
    }

    // The held leaves of the whole wave in one execution (TriShare); every lane of the wave calls this together.
    PD void leaf_wave(const DevScene& S, Cnt<STATS>& cnt) {
        const bool tri_leaf = mode == PBRS_WALK_LEAF;  // analytic shapes never wait here (analytic_visit)
        TriShare sh;
        sh.build(tri_leaf ? leaf_end - leaf_a : 0u);
        if (sh.has[0] == 0) return;
        // helper: the owner's ray in the owner's space, the extent from before its leaf (blas.rs:440-452)
        const f3 ho = sh.from_owner(C.o), hd = sh.from_owner(C.d);
        const float hlt = sh.from_owner(lt);
        const uint32_t hti = sh.from_owner(leaf_a) + sh.k();
        const uint32_t hinfo = (STATS || (FEAT & PBRS_FEAT_SHADING_CHECK)) ? sh.from_owner(inst_info) : 0u;
        const float hmt = (FEAT & PBRS_FEAT_SHADING_CHECK) ? sh.from_owner(mt) : 0.0f;
        float rt = pn_inf(), rb1 = 0.0f, rb2 = 0.0f;
        if (sh.helper()) {
            PBRS_TP(5);
            pbrs_tri_verts tv = load_tri(S.tv + hti);
            CNT(triangles);
            TriHit h;
            // the barycentrics leave the walk only through the shading check, the counters' variant and the parity harness
            // (all FEAT_ALL / STATS); the lean pipeline kernels pass on t, instance and primitive
            constexpr bool need_bary = STATS || (FEAT & PBRS_FEAT_SHADING_CHECK) != 0u;
            bool hit = mesh_tri_hit_t<need_bary>(tv, ho, hd, hlt, S.fast_slab != 0u, h);
            if (hit && (hinfo & 7u) == PBRS_SHAPE_MESH) CNT(tri_shading);
            // The reference builds the shading frame of every geometric hit (blas.rs:166-206) and drops the hit when
            // the tangent check fails (Q22).  Only a hit that would replace outer_hit can change the result (hmt is
            // outer_hit's t before the leaf, an upper bound of it inside the leaf), and for a flat-shaded mesh the
            // check is a host-verified property of the triangles.
            if ((FEAT & PBRS_FEAT_SHADING_CHECK) && hit && h.t < hmt && !((hinfo >> 3) & PBRS_MESH_SHADING_OK_MASK)) {
                f3 n, dpdu;
                hit = mesh_tri_shading(tv, S.ts[hti], hd, h, n, dpdu);
            }
            if (hit) {
                rt = h.t;
                rb1 = h.b1;
                rb2 = h.b2;
            }
        }
        // owner: fold in leaf order, strictly smaller t replaces (blas.rs:445-450)
        uint32_t win = 0xffffffffu, win_tri = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            if (sh.has[j] == 0) break;
            const uint32_t at = sh.pos(j);
            const float t = sh.from_helper(at, rt);
            if (j < sh.cnt && t < mt) {
                mt = t;
                win = at;
                win_tri = leaf_a + j;
            }
        }
        if (__ballot(win != 0xffffffffu)) {
            const float b1 = sh.from_helper(win, rb1), b2 = sh.from_helper(win, rb2);
            if (win != 0xffffffffu) {
                mprim = (inst_info & 7u) == PBRS_SHAPE_MESH ? win_tri : 0u;
                mb1 = b1;
                mb2 = b2;
            }
        }
        if (tri_leaf) {
            leaf_a += sh.cnt;
            if (leaf_a == leaf_end) {
                mode = sp == blas_base ? PBRS_WALK_XFER : PBRS_WALK_NODE;  // nothing pending below this instance: what node_step would find
                lt = mt;  // within a leaf every triangle sees the t_max from before the leaf (blas.rs:440-452)
            }
        }
    }
};

// Any hit: BvhNode::occludes (bvh.rs:105-113), Instance::occludes (instance.rs:68-72), intersect_bvh_pred
// (blas.rs:478-495).  A pure OR over the leaves reached through intersecting boxes with a fixed extent:
// the visiting order cannot change the answer and nothing is carried between instances, so BLAS children
// are visited near-first (by the sign of the ray direction on the split axis), which reaches an occluder
// sooner than the reference's left-first recursion.
template <bool STATS, uint32_t FEAT>
struct AnyWalk {
    RaySpace C;
    float t_max;
    uint32_t leaf_a, leaf_end, inst_kind;
    int sp, blas_base;
    uint32_t cand;  // leaves of the leaf copies at DevScene::flat_off still to visit: their boxes passed the shared scan (0 on a tree walk)
    bool in_blas, occluded, moved;  // moved: C is not the world ray (inst_kind bit 8: only its origin differs)
    uint32_t mode;
    PBRS_TP_FIELDS

    PD void start(const DevScene& S, f3 o, f3 d, float tmax, LaneStack stk) {
        C = make_space(o, d, S.fast_slab != 0);
        moved = false;
        t_max = tmax;
        in_blas = false;
        occluded = false;
        blas_base = 0;
        leaf_a = leaf_end = inst_kind = 0;
        cand = 0;
        if ((FEAT & PBRS_FEAT_FLAT_TLAS) && S.n_flat != 0u && C.fast) {
            sp = 0;
            mode = PBRS_WALK_SCAN;
        } else {
            stk.put(0, 0u);
            sp = 1;
            mode = PBRS_WALK_NODE;
        }
    }
    // The leaf boxes of a small TLAS against the rays that have just started (FlatScan); every lane of the wave calls this
    // together, right after start().  Any-hit: nothing a box test depends on changes during the walk.
    PD void scan_wave(const DevScene& S, Cnt<STATS>& cnt) {
        if (!(FEAT & PBRS_FEAT_FLAT_TLAS)) return;
        uint32_t tested = 0;
        const uint32_t mine = FlatScan::run(S, mode == PBRS_WALK_SCAN, C, t_max, tested);
        if (STATS) cnt.c.tlas_nodes += tested;
        if (mode == PBRS_WALK_SCAN) {
            cand = mine;
            mode = PBRS_WALK_NODE;
        }
    }
    // see ClosestWalk::exit_mode; an unoccluded ray has nothing to carry out of the instance
    PD uint32_t exit_mode() const {
        const bool nothing_above = blas_base == 0 && (!(FEAT & PBRS_FEAT_FLAT_TLAS) || cand == 0u);
        return nothing_above ? PBRS_WALK_DONE : PBRS_WALK_XFER;
    }
    PD void node_step(const DevScene& S, LaneStack stk, Cnt<STATS>& cnt) {
        PBRS_TP(0);
        if (in_blas && sp == blas_base) {
            mode = exit_mode();
            return;
        }
        if (sp == 0) {  // not inside an instance (its exit was taken above), nothing pending: the next scanned leaf, or the end
            if (!(FEAT & PBRS_FEAT_FLAT_TLAS) || cand == 0u) {
                mode = PBRS_WALK_DONE;
                return;
            }
            const uint32_t k = (uint32_t)__builtin_ctz(cand);
            cand &= cand - 1u;
            const pbrs_node leaf = load_node(S.nodes + S.flat_off + k);  // its box passed in scan_wave
            leaf_a = leaf.a;
            inst_kind = (leaf.b >> PBRS_TLAS_LEAF_KIND_SHIFT) & 7u;
            mode = PBRS_WALK_XFER;
            return;
        }
        const uint32_t ni = stk.get(--sp);
        const pbrs_node node = walk_node<FEAT>(S, ni);
        if (STATS) {
            if (in_blas) CNT(blas_nodes);
            else CNT(tlas_nodes);
        }
        PBRS_TP(1);
        if (!slab_rs(node, C, t_max)) {
            PBRS_TP(2);
            if (PBRS_EARLY_OUT && in_blas && sp == blas_base) mode = exit_mode();
            return;
        }
        if (!(node.b & PBRS_LEAF_FLAG)) {
            bool left_first = !in_blas || comp(C.d, (int)(node.b & 3u)) > 0.0f;
            uint32_t left = ni + 1, right = node.a;
            stk.put(sp++, left_first ? right : left);
            stk.put(sp++, left_first ? left : right);
        } else if (in_blas) {
            PBRS_TP(3);
            PBRS_TP(4);
            leaf_a = node.a;
            leaf_end = node.a + (node.b & ~PBRS_LEAF_FLAG);
            if (leaf_end != leaf_a) mode = PBRS_WALK_LEAF;
        } else {
            leaf_a = node.a;  // the instance, until xfer_step replaces it by the held primitive
            inst_kind = (node.b >> PBRS_TLAS_LEAF_KIND_SHIFT) & 7u;
            mode = PBRS_WALK_XFER;
        }
    }
    PD void node_step_fast(const DevScene& S, LaneStack stk, Cnt<STATS>& cnt) {  // see ClosestWalk::node_step_fast
        if (sp == (in_blas ? blas_base : 0) || !C.fast) return;
        PBRS_TP(0);
        const uint32_t ni = stk.get(--sp);
        const pbrs_node node = walk_node<FEAT>(S, ni);
        if (STATS) {
            if (in_blas) CNT(blas_nodes);
            else CNT(tlas_nodes);
        }
        PBRS_TP(1);
        RaySpace F = C;
        F.fast = true;
        if (!slab_rs(node, F, t_max)) {
            PBRS_TP(2);
            return;
        }
        if (!(node.b & PBRS_LEAF_FLAG)) {
            bool left_first = !in_blas || comp(C.d, (int)(node.b & 3u)) > 0.0f;
            uint32_t left = ni + 1, right = node.a;
            stk.put(sp++, left_first ? right : left);
            stk.put(sp++, left_first ? left : right);
        } else if (in_blas) {
            PBRS_TP(3);
            PBRS_TP(4);
            leaf_a = node.a;
            leaf_end = node.a + (node.b & ~PBRS_LEAF_FLAG);
            if (leaf_end != leaf_a) mode = PBRS_WALK_LEAF;
        } else {
            leaf_a = node.a;
            inst_kind = (node.b >> PBRS_TLAS_LEAF_KIND_SHIFT) & 7u;
            mode = PBRS_WALK_XFER;
        }
    }
    // Instance::occludes (instance.rs:68-72) and the return from it; see ClosestWalk::xfer_step.
    PD void xfer_step(const DevScene& S, LaneStack stk, Cnt<STATS>& cnt) {
        mode = PBRS_WALK_NODE;
        if (in_blas) {
            PBRS_TP(7);
            in_blas = false;
            leave_instance(S, stk, !moved ? PBRS_SPACE_WORLD : (inst_kind & 0x100u) ? PBRS_SPACE_TRANSLATED : PBRS_SPACE_MOVED, C);
            return;
        }
        const pbrs_instance& in = S.inst[leaf_a];
        CNT(instances);
        PBRS_TP(6);
        if ((FEAT & PBRS_FEAT_ANALYTIC) && inst_kind != PBRS_SHAPE_MESH && inst_kind != PBRS_SHAPE_TRIANGLE) {
            analytic_visit(S, in, cnt);  // Instance::occludes in one go; back at the TLAS, or occluded
            return;
        }
        const InstHead H = load_inst_head(in);
        const uint32_t space = enter_instance(S, in, H.flags, C, inst_kind == PBRS_SHAPE_MESH, stk);
        moved = space != PBRS_SPACE_WORLD;
        in_blas = true;
        blas_base = sp;
        if (inst_kind == PBRS_SHAPE_MESH) {
            if (space == PBRS_SPACE_TRANSLATED) inst_kind |= 0x100u;  // only meshes: the tests below read the kind before or mask it
            stk.put(sp++, H.blas_root);
        } else if (inst_kind == PBRS_SHAPE_TRIANGLE) {  // IsolatedTriangle::occludes (simple.rs:428-433): its triangle record
            leaf_a = H.blas_root;
            leaf_end = H.blas_root + 1u;
            mode = PBRS_WALK_LEAF;
        }
    }
    // The held leaves of the whole wave in one execution (TriShare); every lane of the wave calls this together.
    // `intersect_bvh_pred` stops at a leaf's first occluder (blas.rs:478-495): the owner counts its triangles up to that one.
    PD void leaf_wave(const DevScene& S, Cnt<STATS>& cnt) {
        const bool tri_leaf = mode == PBRS_WALK_LEAF;  // analytic shapes never wait here (analytic_visit)
        TriShare sh;
        sh.build(tri_leaf ? leaf_end - leaf_a : 0u);
        if (sh.has[0] == 0) return;
        const f3 ho = sh.from_owner(C.o), hd = sh.from_owner(C.d);
        const float htmax = sh.from_owner(t_max);
        const uint32_t hti = sh.from_owner(leaf_a) + sh.k();
        bool hit = false;
        if (sh.helper()) {
            PBRS_TP(5);
            pbrs_tri_verts tv = load_tri(S.tv + hti);
            hit = mesh_tri_pred(tv, ho, hd, htmax);
        }
        const uint64_t hits = __ballot(hit);
        if (tri_leaf) {
            uint32_t tested = sh.cnt;
            bool occ = false;
#pragma unroll
            for (uint32_t j = 4u; j-- > 0u;) {
                if (j < sh.cnt && ((hits >> sh.pos(j)) & 1ull)) {
                    tested = j + 1u;
                    occ = true;
                }
            }
            if (STATS) cnt.c.triangles += tested;
            leaf_a += sh.cnt;
            mode = leaf_a != leaf_end ? PBRS_WALK_LEAF : sp == blas_base ? PBRS_WALK_XFER : PBRS_WALK_NODE;
            if (occ) {
                occluded = true;
                mode = PBRS_WALK_DONE;
            }
        }
    }
    // Instance::occludes (instance.rs:68-72) for an analytic shape in one go, see ClosestWalk::analytic_visit; an occluder
    // ends the walk (mode DONE, occluded set).
    PD void analytic_visit(const DevScene& S, const pbrs_instance& in, Cnt<STATS>& cnt) {
        bool hit;
        {
            if (!(FEAT & PBRS_FEAT_ANALYTIC)) return;
            const f3 o = xf_apply(in.inv, C.o, 1.0f), d = xf_apply(in.inv, C.d, 0.0f);
            const float* p = S.shapes[in.shape_index].p;
            switch (inst_kind) {
                case PBRS_SHAPE_SPHERE:
                    CNT(spheres);
                    hit = sphere_occludes(ld3(p), p[3], o, d, t_max);
                    break;
                case PBRS_SHAPE_QUAD:
                    CNT(quads);
                    hit = quad_occludes(ld3(p), ld3(p + 3), ld3(p + 6), o, d, t_max);
                    break;
                case PBRS_SHAPE_CUBOID:  // Q14: the bbox slab test
                    CNT(cuboids);
                    hit = slab_test(ld3(p), ld3(p + 3), o, d, t_max);
                    break;
                case PBRS_SHAPE_DISK:
                    CNT(disks);
                    hit = disk_occludes(ld3(p), ld3(p + 3), ld3(p + 6), o, d);
                    break;
                default:  // PBRS_SHAPE_TRIANGLE never gets here
                    hit = false;
                    break;
            }
        }
        if (hit) {
            occluded = true;
            mode = PBRS_WALK_DONE;
        }
    }
};

// ---- walks over four-wide nodes (device/wide.h) ----------------------------------------------------------------------------
// For scenes whose TLAS is scanned (PBRS_FEAT_FLAT_TLAS) and rays on the division-free box test.  Same states, same boundary
// and leaf steps as the binary walks above — they inherit them — but: inside a mesh a node step takes a WIDE node (four boxes
// through the conservative filter, the survivors pushed in the reference's order, the first one kept in a register); a BLAS leaf
// that comes up is held UNVERIFIED until the shared leaf step gives it the reference's own box test with the extent of that
// moment; at the TLAS level a scanned leaf is re-evaluated at its turn from the entry distance the scan left in LDS (t_low <=
// t_max: the reference's test, slab_rs_tlow) — no box is fetched twice.  A ray that leaves the guarded range (at its start or
// inside an instance) or whose stack would exceed DevScene::wide_cap takes mode PBRS_WALK_SLOW: the kernel hands it, whole, to
// the binary-walk kernel (kernels.h).
#define PBRS_WALK_SLOW 6u
#define PBRS_LEAF_UNVERIFIED 0xffffffffu
template <uint32_t FEAT>
struct AnyWalkW : AnyWalk<false, FEAT> {
    using B = AnyWalk<false, FEAT>;
    using B::C; using B::t_max; using B::leaf_a; using B::leaf_end; using B::inst_kind; using B::sp; using B::blas_base; using B::cand; using B::in_blas;
    using B::occluded; using B::moved; using B::mode;
    WideRay W;
    uint32_t cur;
    // the lane's space with its reciprocals, for the reference's own test: the wide walk keeps them (as RN(1 / d)) for its filter
    PD RaySpace exact_w() const {
        RaySpace E = C;
        E.nr = -W.r32;
        return E;
    }

    PD void start(const DevScene& S, f3 o, f3 d, float tmax, LaneStack stk) {
        B::start(S, o, d, tmax, stk);
        cur = PBRS_WREF_NONE;
        W.set(C);
        if (mode != PBRS_WALK_SCAN) mode = PBRS_WALK_SLOW;
    }
    PD void scan_wave(const DevScene& S, LaneStack) {
        Cnt<false> cnt;
        B::scan_wave(S, cnt);
    }
    PD uint32_t after_leaf() const { return sp == 0 ? B::exit_mode() : PBRS_WALK_NODE; }
    PD void hold_leaf(uint32_t ref) {
        leaf_a = ref & PBRS_WREF_INDEX;
        leaf_end = PBRS_LEAF_UNVERIFIED;
        mode = PBRS_WALK_LEAF;
    }
    PD void forget_reciprocals() { C.nr = gray(0.0f); }
    // a round's further node steps: a lane with nothing to take from its register or its stack (the next scanned TLAS leaf, the end
    // of a mesh or of the walk: the first step's business) sits them out — see ClosestWalk::node_step_fast
    PD void node_step_fast(const DevScene& S, LaneStack stk, Cnt<false>& cnt) {
        if (cur == PBRS_WREF_NONE && sp == 0) return;
        node_step(S, stk, cnt);
    }
    PD void node_step(const DevScene& S, LaneStack stk, Cnt<false>&) {
        PBRS_TP(0);
        uint32_t e = cur;
        if (e == PBRS_WREF_NONE) {
            if (!in_blas) {  // TLAS level: the next leaf that passed the scan (the reference's test: the extent never changes)
                if (cand == 0u) {
                    mode = PBRS_WALK_DONE;
                    return;
                }
                leaf_a = (uint32_t)__builtin_ctz(cand);
                cand &= cand - 1u;
                mode = PBRS_WALK_XFER;
                return;
            }
            if (sp == 0) {
                mode = B::exit_mode();
                return;
            }
            e = stk.get(--sp);
        }
        cur = PBRS_WREF_NONE;
        if (e & PBRS_WREF_LEAF) {
            hold_leaf(e);
            return;
        }
        PBRS_TP(1);
        const WideTest t = wide_test(S.wnodes, e, C, W, t_max);
        if (t.pass == 0u) {
            PBRS_TP(2);
            if (sp == 0) mode = B::exit_mode();
            return;
        }
        if (sp + 3 > (int)S.wide_cap) {
            mode = PBRS_WALK_SLOW;
            return;
        }
        const uint32_t first = wide_push(wide_order_any(t, C.d), stk, sp);
        if (first & PBRS_WREF_LEAF) hold_leaf(first);
        else cur = first;
    }
    PD void xfer_step(const DevScene& S, LaneStack stk, Cnt<false>& cnt) {
        if (in_blas) {
            const bool rebuilt = moved && !(inst_kind & 0x100u);  // (AnyWalk::xfer_step's first branch: see ClosestWalkW::xfer_step)
            mode = PBRS_WALK_NODE;
            in_blas = false;
            leave_instance(S, stk, !moved ? PBRS_SPACE_WORLD : (inst_kind & 0x100u) ? PBRS_SPACE_TRANSLATED : PBRS_SPACE_MOVED, C);
            if (rebuilt) W.set(C);
            return;
        }
        mode = PBRS_WALK_NODE;
        PBRS_TP(6);
        const pbrs_node leaf = load_node(S.nodes + S.flat_off + leaf_a);
        leaf_a = leaf.a;  // the instance
        inst_kind = (leaf.b >> PBRS_TLAS_LEAF_KIND_SHIFT) & 7u;
        const pbrs_instance& in = S.inst[leaf_a];
        if ((FEAT & PBRS_FEAT_ANALYTIC) && inst_kind != PBRS_SHAPE_MESH && inst_kind != PBRS_SHAPE_TRIANGLE) {
            B::analytic_visit(S, in, cnt);
            return;
        }
        const InstHead H = load_inst_head(in);
        const uint32_t space = enter_instance(S, in, H.flags, C, inst_kind == PBRS_SHAPE_MESH, stk);
        moved = space != PBRS_SPACE_WORLD;
        in_blas = true;
        blas_base = 0;
        if (inst_kind == PBRS_SHAPE_MESH) {
            if (!C.fast) {
                mode = PBRS_WALK_SLOW;
                return;
            }
            if (space == PBRS_SPACE_TRANSLATED) inst_kind |= 0x100u;
            if (space == PBRS_SPACE_MOVED) W.set(C);  // a new direction (make_space has just computed its reciprocals); else the world's stands
            const uint32_t wroot = H.wide_root;
            if (wroot == PBRS_WREF_NONE) {
                hold_leaf(H.blas_root);
                return;
            }
            // The root's own test is not needed for the answer (any hit: inner-node tests only prune), but it is one test that ends most
            // misses here — also behind an identity transform, where the scan has tested the same box: skipping it there was
            // measured slower (C4 k_shadow 238.5 -> 241.5 ms per frame; the first wide node's four tests cost more than this one)
            if (!slab_rs(load_node(S.nodes + H.blas_root), exact_w(), t_max)) {
                mode = B::exit_mode();
                return;
            }
            cur = wroot;
        } else if (inst_kind == PBRS_SHAPE_TRIANGLE) {
            leaf_a = H.blas_root;
            leaf_end = H.blas_root + 1u;
            mode = PBRS_WALK_LEAF;
        }
    }
    PD void leaf_wave(const DevScene& S, Cnt<false>& cnt) {
        if (mode == PBRS_WALK_LEAF && leaf_end == PBRS_LEAF_UNVERIFIED) {  // the reference's test of the leaf's own box (intersect_bvh_pred, blas.rs:478-495)
            PBRS_TP(3);
            const pbrs_node node = load_node(S.nodes + leaf_a);
            leaf_a = node.a;
            leaf_end = slab_rs(node, exact_w(), t_max) ? node.a + (node.b & ~PBRS_LEAF_FLAG) : node.a;
            if (leaf_end != leaf_a) PBRS_TP(4);
            if (leaf_end == leaf_a) mode = after_leaf();
        }
        const bool tri_leaf = mode == PBRS_WALK_LEAF;
        TriShare sh;
        sh.build(tri_leaf ? leaf_end - leaf_a : 0u);
        if (sh.has[0] == 0) return;
        const f3 ho = sh.from_owner(C.o), hd = sh.from_owner(C.d);
        const float htmax = sh.from_owner(t_max);
        const uint32_t hti = sh.from_owner(leaf_a) + sh.k();
        bool hit = false;
        if (sh.helper()) {
            PBRS_TP(5);
            pbrs_tri_verts tv = load_tri(S.tv + hti);
            hit = mesh_tri_pred(tv, ho, hd, htmax);
        }
        const uint64_t hits = __ballot(hit);
        if (tri_leaf) {
            bool occ = false;
#pragma unroll
            for (uint32_t j = 4u; j-- > 0u;)
                if (j < sh.cnt && ((hits >> sh.pos(j)) & 1ull)) occ = true;
            leaf_a += sh.cnt;
            mode = leaf_a != leaf_end ? PBRS_WALK_LEAF : after_leaf();
            if (occ) {
                occluded = true;
                mode = PBRS_WALK_DONE;
            }
        }
    }
};

// One ray per lane start to finish (parity harness; the pipeline kernels interleave walks and refill lanes instead).
// Every lane of the wave calls these together (leaf_wave); `active` = the lane has a ray.
template <bool STATS, uint32_t EXTRA = 0u>  // EXTRA: PBRS_FEAT_EXTENT for the scenes whose pipeline walks with it
PD void tlas_closest(const DevScene& S, bool active, f3 o, f3 d, float t_max, LaneStack stk, Hit& best, Cnt<STATS>& cnt) {
    ClosestWalk<STATS, PBRS_FEAT_ALL | EXTRA> w;
    w.start(S, o, d, t_max, stk);
    if (!active) w.mode = PBRS_WALK_DONE;
    w.scan_wave(S, cnt);
    while (__ballot(w.mode != PBRS_WALK_DONE)) {
        if (w.mode == PBRS_WALK_XFER) w.xfer_step(S, stk, cnt);
        if (w.mode == PBRS_WALK_NODE) w.node_step(S, stk, cnt);
        if (__ballot(w.mode == PBRS_WALK_LEAF)) w.leaf_wave(S, cnt);
    }
    w.finish(cnt);
    best = w.best;
}
template <bool STATS>
PD bool tlas_any(const DevScene& S, bool active, f3 o, f3 d, float t_max, LaneStack stk, Cnt<STATS>& cnt) {
    AnyWalk<STATS, PBRS_FEAT_ALL> w;
    w.start(S, o, d, t_max, stk);
    if (!active) w.mode = PBRS_WALK_DONE;
    w.scan_wave(S, cnt);
    while (__ballot(w.mode != PBRS_WALK_DONE)) {
        if (w.mode == PBRS_WALK_XFER) w.xfer_step(S, stk, cnt);
        if (w.mode == PBRS_WALK_NODE) w.node_step(S, stk, cnt);
        if (__ballot(w.mode == PBRS_WALK_LEAF)) w.leaf_wave(S, cnt);
    }
    return w.occluded;
}
// The same through the four-wide any-hit walk (parity harness of k_shadow's wide kernels): rays that walk refuses fall back to the
// binary walk, as they do in the pipeline.
PD bool tlas_any_wide(const DevScene& S, bool active, f3 o, f3 d, float t_max, LaneStack stk, bool& slow) {
    Cnt<false> cnt;
    AnyWalkW<PBRS_FEAT_ALL> w;
    w.start(S, o, d, t_max, stk);
    if (!active) w.mode = PBRS_WALK_DONE;
    w.scan_wave(S, stk);
    while (__ballot(w.mode == PBRS_WALK_NODE || w.mode == PBRS_WALK_LEAF || w.mode == PBRS_WALK_XFER)) {
        if (w.mode == PBRS_WALK_XFER) w.xfer_step(S, stk, cnt);
        if (w.mode == PBRS_WALK_NODE) w.node_step(S, stk, cnt);
        if (__ballot(w.mode == PBRS_WALK_LEAF)) w.leaf_wave(S, cnt);
    }
    slow = w.mode == PBRS_WALK_SLOW;  // refused (outside the guarded range, or the stack would not fit): the binary walk's ray, as in the pipeline
    bool occ = w.occluded;
    if (__ballot(slow)) {
        const bool o2 = tlas_any<false>(S, active && slow, o, d, t_max, stk, cnt);
        if (slow) occ = o2;
    }
    return occ;
}
