// device/experimental/closest_wide.h — the four-wide CLOSEST-hit walk (developer builds only: -DPBRS_DEV_OVERRIDES, PBRS_WIDE bit 0).
//
// Bit-exact against the oracle (tools/dev_parity.sh) and within 8 % of the binary walk on C4 (428 against 394-412 ms of k_extend per
// frame, round 3): 13 node steps per ray instead of 44, but 117-125 registers — five waves per SIMD where the binary walk has six.
// Kept out of the product headers: the shipped library never launches it (pbrs_gpu.hip, pbrs_upload_scene).  The pair-node walks,
// the walks over compressed 16-byte records and the GRID walks of round 3 (all measured slower) were deleted in round 4
// and are reproducible from commits cf37935 / ec394ac.
#pragma once
#include "../traverse.h"

template <uint32_t FEAT>
struct ClosestWalkW : ClosestWalk<false, FEAT> {
    using B = ClosestWalk<false, FEAT>;
    using B::C; using B::best; using B::t_max; using B::lt; using B::mt; using B::mb1; using B::mb2; using B::mprim; using B::cur_inst;
    using B::inst_info; using B::leaf_a; using B::leaf_end; using B::sp; using B::blas_base; using B::cand; using B::in_blas; using B::moved; using B::mode;
    WideRay W;
    uint32_t cur;  // wide node to take next (the nearest survivor of the last node step), or PBRS_WREF_NONE
    // the lane's space with its reciprocals, for the reference's own test: the wide and pair walks keep them (as RN(1 / d)) for their filter
    PD RaySpace exact_w() const {
        RaySpace E = C;
        E.nr = -W.r32;
        return E;
    }
    // A BLAS leaf held UNVERIFIED is (leaf_a = its node index, leaf_end = PBRS_LEAF_UNVERIFIED); the scanned TLAS leaf about to be
    // entered travels in leaf_a too.  The lane's column of the block's entry-distance table sits after the stack rows:
    // row DevScene::wide_cap + k for scanned leaf k.  A wide walk is never below a TLAS entry (the TLAS is scanned): blas_base = 0.

    PD void start(const DevScene& S, f3 o, f3 d, float tmax, LaneStack stk) {
        B::start(S, o, d, tmax, stk);
        cur = PBRS_WREF_NONE;
        W.set(C);
        if (mode != PBRS_WALK_SCAN) mode = PBRS_WALK_SLOW;  // not on the division-free test: the binary walk's ray
    }
    PD void scan_wave(const DevScene& S, LaneStack stk) {
        float* tl_block = reinterpret_cast<float*>(stk.base) - (threadIdx.x & (PBRS_TRAVERSAL_BLOCK - 1)) + S.wide_cap * PBRS_TRAVERSAL_BLOCK;
        const uint32_t mine = flat_scan_tlow(S, mode == PBRS_WALK_SCAN, C, tl_block);
        if (mode == PBRS_WALK_SCAN) {
            cand = mine;
            mode = PBRS_WALK_NODE;
        }
    }
    PD uint32_t after_leaf() const { return sp == 0 ? B::exit_mode() : PBRS_WALK_NODE; }
    PD void hold_leaf(uint32_t ref) {
        leaf_a = ref & PBRS_WREF_INDEX;
        leaf_end = PBRS_LEAF_UNVERIFIED;
        mode = PBRS_WALK_LEAF;
    }
    // The space's own reciprocals are not state of a wide walk (exact_w): dropping them at the end of every loop round keeps three
    // registers from living across it (the shared scan reads them from every lane of the wave, fresh or not).
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
            if (!in_blas) {  // TLAS level: the next scanned leaf whose box the reference's test passes NOW (t_low <= min(hi_el, t_max))
                if (cand == 0u) {
                    mode = PBRS_WALK_DONE;
                    return;
                }
                const uint32_t k = (uint32_t)__builtin_ctz(cand);
                cand &= cand - 1u;
                if (__uint_as_float(stk.get((int)(S.wide_cap + k))) <= t_max) {
                    leaf_a = k;
                    mode = PBRS_WALK_XFER;
                }
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
        const WideTest t = wide_test(S.wnodes, e, C, W, lt);  // inside a mesh lt == mt at every node (blas.rs:468), after the root
        if (t.pass == 0u) {
            PBRS_TP(2);
            if (sp == 0) mode = B::exit_mode();
            return;
        }
        if (sp + 3 > (int)S.wide_cap) {  // the pushes below might not fit: the binary walk takes this ray from its start
            mode = PBRS_WALK_SLOW;
            return;
        }
        const uint32_t first = wide_push(wide_order(t, C.d), stk, sp);  // the reference's order; its first is taken next, from the register
        if (first & PBRS_WREF_LEAF) hold_leaf(first);
        else cur = first;
    }
    PD void xfer_step(const DevScene& S, LaneStack stk, Cnt<false>& cnt) {
        if (in_blas) {
            // the way out: Instance::intersect returns (bvh.rs:82-95) — ClosestWalk::xfer_step's first branch, restated here so that its
            // second one (the way in, which this walk has its own version of below) is not compiled into this kernel twice: with both
            // copies the allocator spilled some 40 registers around them
            const bool rebuilt = moved && !(inst_info & 0x40000000u);  // leave_instance rebuilds the world ray, reciprocals included
            mode = PBRS_WALK_NODE;
            in_blas = false;
            leave_instance(S, stk, !moved ? PBRS_SPACE_WORLD : (inst_info & 0x40000000u) ? PBRS_SPACE_TRANSLATED : PBRS_SPACE_MOVED, C);
            B::meet_best(cnt);
            lt = t_max;  // back in the TLAS
            if (rebuilt) W.set(C);
            return;
        }
        mode = PBRS_WALK_NODE;
        PBRS_TP(6);
        const pbrs_node leaf = load_node(S.nodes + S.flat_off + leaf_a);  // its box passed at this moment (node_step)
        cur_inst = leaf.a;
        const uint32_t kind = (leaf.b >> PBRS_TLAS_LEAF_KIND_SHIFT) & 7u;
        inst_info = kind;
        const pbrs_instance& in = S.inst[cur_inst];
        if ((FEAT & PBRS_FEAT_ANALYTIC) && kind != PBRS_SHAPE_MESH && kind != PBRS_SHAPE_TRIANGLE) {
            B::analytic_visit(S, in, kind, cnt);
            return;
        }
        const InstHead H = load_inst_head(in);
        const uint32_t space = enter_instance(S, in, H.flags, C, kind == PBRS_SHAPE_MESH, stk);
        moved = space != PBRS_SPACE_WORLD;
        in_blas = true;
        blas_base = 0;
        lt = t_max;
        mt = pn_inf();
        if (kind == PBRS_SHAPE_MESH) {
            if (!C.fast) {  // the instance's space is outside the guarded range
                mode = PBRS_WALK_SLOW;
                return;
            }
            inst_info = kind | (H.mesh_flags << 3) | (space == PBRS_SPACE_TRANSLATED ? 0x40000000u : 0u);
            if (space == PBRS_SPACE_MOVED) W.set(C);  // a new direction (make_space has just computed its reciprocals); else the world's stands
            const uint32_t wroot = H.wide_root;
            if (wroot == PBRS_WREF_NONE) {  // the mesh is one leaf: its box is tested, against the incoming extent, with its triangles
                hold_leaf(H.blas_root);
                return;
            }
            // the root against the incoming extent (blas.rs:441 at the first pop), the reference's test; then lt = mt (:468)
            if (!slab_rs(load_node(S.nodes + H.blas_root), exact_w(), lt)) {
                mode = B::exit_mode();
                return;
            }
            lt = mt;
            cur = wroot;
        } else if (kind == PBRS_SHAPE_TRIANGLE) {
            inst_info = kind | (PBRS_MESH_SHADING_OK_MASK << 3);
            leaf_a = H.blas_root;
            leaf_end = H.blas_root + 1u;
            mode = PBRS_WALK_LEAF;
        }
    }
    // The held leaves of the whole wave: first the reference's box test for the unverified ones, then ClosestWalk::leaf_wave's
    // shared triangle tests (same values, same order).
    PD void leaf_wave(const DevScene& S, Cnt<false>& cnt) {
        if (mode == PBRS_WALK_LEAF && leaf_end == PBRS_LEAF_UNVERIFIED) {
            PBRS_TP(3);
            const pbrs_node node = load_node(S.nodes + leaf_a);
            leaf_a = node.a;
            leaf_end = node.a;
            if (slab_rs(node, exact_w(), lt)) {
                PBRS_TP(4);
                leaf_end = node.a + (node.b & ~PBRS_LEAF_FLAG);
                if (leaf_end == leaf_a) lt = mt;  // an empty leaf still runs blas.rs:468
            }
            if (leaf_end == leaf_a) mode = after_leaf();
        }
        const bool tri_leaf = mode == PBRS_WALK_LEAF;
        TriShare sh;
        sh.build(tri_leaf ? leaf_end - leaf_a : 0u);
        if (sh.has[0] == 0) return;
        const f3 ho = sh.from_owner(C.o), hd = sh.from_owner(C.d);
        const float hlt = sh.from_owner(lt);
        const uint32_t hti = sh.from_owner(leaf_a) + sh.k();
        const uint32_t hinfo = (FEAT & PBRS_FEAT_SHADING_CHECK) ? sh.from_owner(inst_info) : 0u;
        const float hmt = (FEAT & PBRS_FEAT_SHADING_CHECK) ? sh.from_owner(mt) : 0.0f;
        float rt = pn_inf(), rb1 = 0.0f, rb2 = 0.0f;
        if (sh.helper()) {
            PBRS_TP(5);
            pbrs_tri_verts tv = load_tri(S.tv + hti);
            TriHit h;
            constexpr bool need_bary = (FEAT & PBRS_FEAT_SHADING_CHECK) != 0u;
            bool hit = mesh_tri_hit_t<need_bary>(tv, ho, hd, hlt, S.fast_slab != 0u, h);
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
                mode = after_leaf();
                lt = mt;
            }
        }
    }
};


// The parity harness of the walk above: rays it refuses fall back to the binary walk, as they do in the pipeline.  The block's
// entry-distance table sits after DevScene::wide_cap stack rows.
PD void tlas_closest_wide(const DevScene& S, bool active, f3 o, f3 d, float t_max, LaneStack stk, Hit& best, bool& slow) {
    Cnt<false> cnt;
    ClosestWalkW<PBRS_FEAT_ALL> w;
    w.start(S, o, d, t_max, stk);
    if (!active) w.mode = PBRS_WALK_DONE;
    w.scan_wave(S, stk);
    while (__ballot(w.mode == PBRS_WALK_NODE || w.mode == PBRS_WALK_LEAF || w.mode == PBRS_WALK_XFER)) {
        if (w.mode == PBRS_WALK_XFER) w.xfer_step(S, stk, cnt);
        if (w.mode == PBRS_WALK_NODE) w.node_step(S, stk, cnt);
        if (__ballot(w.mode == PBRS_WALK_LEAF)) w.leaf_wave(S, cnt);
    }
    slow = w.mode == PBRS_WALK_SLOW;
    w.finish(cnt);
    best = w.best;
    if (__ballot(slow)) {
        Hit b2;
        tlas_closest<false>(S, active && slow, o, d, t_max, stk, b2, cnt);
        if (slow) best = b2;
    }
}
