// device/kernels.h — the wavefront stages.
//
//   raygen      src/main.rs:197-203 + geometry/src/camera.rs:65-77
//   extend      tlas/src/bvh.rs:77-103 (closest hit)            -> hit record per path
//   shade       src/pathintegrator.rs:19-71 + src/directlighting.rs:58-222 minus the occlusion calls
//   shadow      tlas/src/bvh.rs:105-113 (any hit) + the radiance add of pathintegrator.rs:35
//   accumulate  src/main.rs:205-208 (in-order f32 sum over samples, then * 1/spp)
//
// One lane owns one path vertex (k_shade) or one ray (k_extend / k_shadow).  What a path carries from bounce to bounce
// travels as DENSE records at the path's position in the bounce's queue — k_shade writes the next ray where the
// compaction puts the path, k_extend writes the hit next to it, k_shade reads both back — so every access of the
// traversal kernels and all but one of k_shade's is a full-width coalesced 16-byte-per-lane stream.  Only the radiance
// accumulator stays at a fixed address per path (slot = k * P + pixel: k_accumulate sums the samples of a pixel in
// order).  The `break`s of the bounce loop (pathintegrator.rs:25-27, :48-50, :67-69) become "not appended to the next
// queue", compacted with __ballot + mbcnt prefixes summed per block in LDS (two atomics per block).
#pragma once
#include "lights.h"
#include "textures.h"
#include "traverse.h"
#ifdef PBRS_DEV_OVERRIDES
#include "experimental/closest_wide.h"
#endif

struct PathState {
    // Path records by queue position i, structure-of-float4-arrays, ping-pong by bounce parity (k_shade reads set b & 1
    // and writes set (b + 1) & 1):
    //   q[.][0][i] = origin.xyz, slot | specular_bounce << 31        (t_max is always +inf for path rays: Ray::new)
    //   q[.][1][i] = dir.xyz,    RNG state, low word
    //   q[.][2][i] = beta.xyz,   RNG state, high word
    float4* q[2][3];
    float4* hit;  // [i] = t, inst (0xffffffff: miss), prim, shading class   written by k_extend at the ray's queue position
    uint8_t* cls;     // [queue position] shading class of the hit, a byte beside hit[].w for the class sort's two passes
    uint32_t* perm;  // scenes with several shading classes: k_shade's lane j shades the path at queue position perm[j] (k_class_sort)
    // class-major order (scenes whose Lambertian class gets the Lambert variant of k_shade): per tile and class, the count
    // and then the offset inside the class; per class (and, at [PBRS_MAX_CLASSES], for all classes but the Lambertian) the
    // range of lanes that shade it
    uint32_t* tile_hist;
    uint2* class_range;
    float4* L;    // [slot] = radiance.xyz, w: the direct integrator's 1 / mass (src/directlighting.rs:37), else unused
    // Next-event estimation hand-off.  Shadow rays by position j in the bounce's shadow queue:
    //   sr[0][j] = origin.xyz, t_max        sr[1][j] = dir.xyz, item
    //   sr[2][j] = lone ray: the path's radiance if the ray is unoccluded (k_shade has stored the occluded outcome in L)
    // item = slot | 1 << 30 for a path's only shadow ray (the lane that traces it finishes the estimate), else
    // slot | ray index << 31: a path that casts both of its rays (the two MIS terms of an area light) cannot be finished
    // by either tracing lane — its terms wait in nee[] (by slot) and k_nee_resolve combines them with the two occlusion bytes.
    float4* sr[3];
    float4* nee[3];   // [slot]: c1.xyz, 1 / light_pdf | c2.xyz, post factor | beta at the time of the estimate, -
    uint8_t* occ[2];  // [slot], written by k_shadow: 1 = the ray is occluded
};
#define PBRS_STATE_BYTES_PER_PATH (2u * 48u + 16u + 4u + 1u + 16u + 2u * 48u + 48u + 2u + 4u + 8u)  // records above + the nee queue entry + two slow-list entries

struct RenderConst {
    pbrs_camera cam;
    uint32_t x0, y0, w, h;
    uint32_t strata_x, strata_y;
    uint32_t max_depth;
    uint32_t pass_first_sample;  // sample index of k = 0 in this pass
    uint32_t n_pixels;           // P
    uint32_t n_slots;            // P * K of this pass
    uint32_t chunk_pixels;       // slot order of a pass (slot_of_sample): chunks of this many pixels, each with its K samples
    uint32_t tiles8_per_row;     // w / 8 when the pixels of the tile are taken in 8 x 8 blocks (pixel_of_order), else 0
    uint32_t band_rows, band_count, band_index;  // interleaved row bands (pbrs_render_params)
    uint32_t integrator;                         // PBRS_INTEGRATOR_*
    uint64_t seed;
};

// Byte / word columns by path slot (the nee hand-off's occlusion bytes).  Slots stay below 2^28 (check_params).
template <class T>
PD T& at(T* base, uint32_t slot) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 1, "column element size");
    return *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + (size_t)(uint32_t)(slot * (uint32_t)sizeof(T)));
}
PD float4 pack4(f3 v, uint32_t w) { return make_float4(v.x, v.y, v.z, __uint_as_float(w)); }
PD float4 pack4(f3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
PD f3 xyz(float4 v) { return mk3(v.x, v.y, v.z); }
#define PBRS_SLOT_MASK 0x3fffffffu
// Queue and state records are streams: written by one kernel, read once by the next, tens of GB apart.  k_shade's and k_raygen's record
// traffic and k_extend's hit store are marked non-temporal, so that they do not push scene data out of the L2 (C4 +0.7 %, C2 +0.6 %, C3
// +1 %, c4xl +0.7 % in same-box A/B: profiles/r04r_ab_nontemporal_streams.log).  The rays a traversal kernel fetches stay ordinary loads:
// a walk reads its ray record again when it leaves an instance (traverse.h, reload_world).
PD float4 ld_stream(const float4* p) {
    const pbrs_f4v v = __builtin_nontemporal_load(reinterpret_cast<const pbrs_f4v*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
PD void st_stream(float4* p, float4 v) {
    pbrs_f4v w;
    w.x = v.x, w.y = v.y, w.z = v.z, w.w = v.w;
    __builtin_nontemporal_store(w, reinterpret_cast<pbrs_f4v*>(p));
}

// ---- slot order of a pass ----------------------------------------------------------------------------------------
// The P * K camera samples of a pass, in queue order: the pixels are cut into chunks of C (RenderConst::chunk_pixels; the
// last one shorter), and a chunk's samples come sample index by sample index — slot = chunk * C * K + k * C_chunk + pixel in
// chunk.  A wave still holds 64 neighbouring pixels of one sample index, but the K samples of a chunk now follow each
// other, and since a traversal kernel's blocks take the queue in eight contiguous segments, one per XCD (wave_fetch), an
// XCD works through its own band of the image chunk by chunk: the BVH subtrees and triangles under a chunk are fetched
// into that XCD's L2 once and serve all K samples, instead of once per sample index in all eight L2s (the order K, then
// pixels, that a plain `slot = k * P + pixel` gives).  The radiance of a sample does not depend on its slot.
PD void sample_of_slot(uint32_t slot, uint32_t n_pixels, uint32_t k_count, uint32_t chunk, uint32_t& k, uint32_t& pix) {
    const uint32_t per_chunk = chunk * k_count;
    const uint32_t c = slot / per_chunk, rem = slot - c * per_chunk, first = c * chunk;
    const uint32_t size = n_pixels - first < chunk ? n_pixels - first : chunk;
    k = rem / size;
    pix = first + (rem - k * size);
}
PD uint32_t slot_of_sample(uint32_t k, uint32_t pix, uint32_t n_pixels, uint32_t k_count, uint32_t chunk) {
    const uint32_t c = pix / chunk, first = c * chunk;
    const uint32_t size = n_pixels - first < chunk ? n_pixels - first : chunk;
    return c * chunk * k_count + k * size + (pix - first);
}

// The order the pixels of the tile are taken in: 8 x 8 blocks, row-major over the blocks and inside a block, when the
// tile's width and height are multiples of 8 (tiles8_per_row != 0) — a wave's 64 camera rays then leave through a square of
// the film instead of a 64-pixel line, and walk the same nodes for longer — else row-major.  order <-> pixel (row-major):
PD uint32_t pixel_of_order(uint32_t q, uint32_t w, uint32_t tiles8_per_row) {
    if (tiles8_per_row == 0u) return q;
    const uint32_t t = q >> 6, r = q & 63u;
    const uint32_t ty = t / tiles8_per_row, tx = t - ty * tiles8_per_row;
    return (ty * 8u + (r >> 3)) * w + tx * 8u + (r & 7u);
}
PD uint32_t order_of_pixel(uint32_t pix, uint32_t w, uint32_t tiles8_per_row) {
    if (tiles8_per_row == 0u) return pix;
    const uint32_t y = pix / w, x = pix - y * w;
    return (((y >> 3) * tiles8_per_row + (x >> 3)) << 6) + ((y & 7u) << 3) + (x & 7u);
}

// ---- raygen --------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_raygen(PathState st, RenderConst rc) {
    uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= rc.n_slots) return;
    uint32_t k, pix;
    sample_of_slot(slot, rc.n_pixels, rc.n_slots / rc.n_pixels, rc.chunk_pixels, k, pix);
    pix = pixel_of_order(pix, rc.w, rc.tiles8_per_row);
    uint32_t col = rc.x0 + pix % rc.w, vrow = pix / rc.w;
    uint32_t row = rc.y0 + (rc.band_count > 1 ? ((vrow / rc.band_rows) * rc.band_count + rc.band_index) * rc.band_rows + vrow % rc.band_rows : vrow);
    uint32_t i = rc.pass_first_sample + k;
    uint64_t rng = pn_rng_init(rc.seed, row * rc.cam.width + col, i);
    float r0 = pn_rng_f32(&rng), r1 = pn_rng_f32(&rng);
    // src/main.rs:198-201 with msaa -> (strata_x, strata_y): i / msaa, i % msaa
    float jx = ((float)(i / rc.strata_y) + r0) / (float)rc.strata_x;
    float jy = ((float)(i % rc.strata_y) + r1) / (float)rc.strata_y;
    if (rc.integrator >= PBRS_INTEGRATOR_MATERIALS) jx = jy = 0.0f;  // the visualisers: `shoot_ray(row, col, (0.0, 0.0))`, src/main.rs:170
    // Camera::shoot_ray, camera.rs:65-77
    float x = (float)col + pn_fract(jx);
    float y = (float)row + pn_fract(jy);
    f3 dir = ld3(rc.cam.c) + ld3(rc.cam.a) * x + ld3(rc.cam.b) * y;
    // bounce 0: the queue position of a path is its slot
    st_stream(&st.q[0][0][slot], pack4(ld3(rc.cam.center), slot));
    st_stream(&st.q[0][1][slot], pack4(dir, (uint32_t)rng));
    st_stream(&st.q[0][2][slot], pack4(gray(1.0f), (uint32_t)(rng >> 32)));
    st_stream(&st.L[slot], make_float4(0.0f, 0.0f, 0.0f, 0.0f));
}

struct GlobalCounters {  // instrumented variant only
    unsigned long long tlas_nodes, blas_nodes, instances, instance_hits, triangles, tri_shading, spheres, quads, cuboids, disks, rays, hits;
};
template <bool STATS>
PD void flush_counters(const Cnt<STATS>& cnt, GlobalCounters* g, bool valid, uint32_t rays, uint32_t hits) {
    if (!STATS) return;
    const WorkCounters& c = cnt.c;
    uint32_t v[12] = {c.tlas_nodes, c.blas_nodes, c.instances, c.instance_hits, c.triangles, c.tri_shading,
                      c.spheres,    c.quads,      c.cuboids,   c.disks,         rays,        hits};
    unsigned long long* out = reinterpret_cast<unsigned long long*>(g);
    for (int k = 0; k < 12; ++k) {
        uint32_t x = valid ? v[k] : 0u;
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63u) == 0 && x) atomicAdd(out + k, (unsigned long long)x);
    }
}

// ---- extend ----------------------------------------------------------------------------------------------------
// A wave takes new rays when fewer than DevScene::refill_below of its lanes are walking (pbrs_upload_scene): short walks
// (a small TLAS scanned by the wave, BLASes of a few nodes) favour late, large refills — the shared scan of the new rays
// fills its windows and the step kernels run on fuller waves less often; long walks (C4's 18-level BLAS) favour early ones.
#ifndef PBRS_REFILL_BELOW_SHORT
#define PBRS_REFILL_BELOW_SHORT 20u  // C2 extend 9.23 / 8.92 / 9.04 ms per 16 spp at 40 / 24 / 16
#endif
#ifndef PBRS_REFILL_BELOW_LONG
#define PBRS_REFILL_BELOW_LONG 48u   // C4 extend 25.9 / 24.5 / 23.8 ms per 16 spp at 24 / 40 / 48; per frame 398.5 / 392.4 / 418.3 / 452.5 ms at 40 / 48 / 56 / 60 (round 4)
#endif
#ifndef PBRS_REFILL_BELOW_LONG_SHADOW
#define PBRS_REFILL_BELOW_LONG_SHADOW 40u  // C4 shadow 205.3 / 209.9 / 231.5 ms per frame at 40 / 48 / 56 (profiles/r04j_ab_refill_thresholds_c4.log)
#endif
#define PBRS_LONG_WALK_HEIGHT 12u    // a mesh whose BLAS is at least this high makes the scene's walks "long"
#ifndef PBRS_NODE_STEPS_LONG
#define PBRS_NODE_STEPS_LONG 3u
#endif
#ifndef PBRS_CHUNK_MAX
#define PBRS_CHUNK_MAX 512u
#endif
#ifndef PBRS_TRAV_WAVES  // min waves per SIMD asked of the register allocator for k_extend / k_shadow
#define PBRS_TRAV_WAVES 5
#endif
#ifndef PBRS_LEAN_EXTEND_WAVES  // ... and for the k_extend variants without the per-candidate shading check: 80 VGPRs, two
#define PBRS_LEAN_EXTEND_WAVES 6  // dwords of the shared leaf step spilled — C2 extend 13.6 -> 12.8 ms per 16 spp over 5 waves
#endif
#ifndef PBRS_SHADOW_WAVES  // ... and for k_shadow: 76-80 VGPRs without a spill (C4 shadow 347 -> 317 ms, C3 198 -> 180 ms over 5 waves)
#define PBRS_SHADOW_WAVES 6
#endif
#ifndef PBRS_SHADE_WAVES  // min waves per SIMD asked of the register allocator for k_shade (2nd arg of __launch_bounds__)
#define PBRS_SHADE_WAVES 3
#endif
#ifndef PBRS_FOURIER_SHADE_WAVES  // ... and for its variants that carry the Fourier lobe (device/fourier.h).  At three waves they keep 168
#define PBRS_FOURIER_SHADE_WAVES 3  // registers and 428 bytes of scratch per lane, at two 255 registers and none — and are SLOWER: a 960 x 640,
#endif                              // 64-spp frame of tests/fourier_scenes.py shades in 124 / 228 ms at three and 147 / 285 ms at two
                                    // (tools/fourier_bench.py).  The spill code sits outside the series loops (one reload in 7 300
                                    // instructions at loop depth 2); what the lobe costs is the recomputed series itself.

// One round of the traversal loop ("if-if"): lanes in the node state take a node step, then the wave tests the primitives
// of its held leaves (leaf_wave: triangle tests shared out over all 64 lanes), so lanes in different phases of their walks
// share the instruction stream.  Lanes waiting at an instance boundary cross it together once PBRS_XFER_MIN of them wait,
// leaves are tested once PBRS_LEAF_MIN lanes hold one — or when no lane of the wave can do anything else.
// The thresholds differ per kernel (measured, C2 / C4): closest-hit walks gain from batching both (a triangle test with
// its four divisions is the longest step); any-hit walks end at the first hit, where waiting at a boundary costs more
// than it saves.
#ifndef PBRS_EXT_XFER_MIN
#define PBRS_EXT_XFER_MIN 8
#endif
#ifndef PBRS_EXT_LEAF_MIN
#define PBRS_EXT_LEAF_MIN 10
#endif
#ifndef PBRS_SHD_XFER_MIN
#define PBRS_SHD_XFER_MIN 2
#endif
#ifndef PBRS_SHD_LEAF_MIN
#define PBRS_SHD_LEAF_MIN 8
#endif
// Scenes with long walks (a BLAS of PBRS_LONG_WALK_HEIGHT levels or more: PBRS_FEAT_LONG_WALKS) take PBRS_NODE_STEPS_LONG
// node steps per loop round: most rounds of a deep walk are node steps, and the checks around them (who waits at a
// boundary, who holds a leaf, who is done) then run a third as often.  C4 (23 levels), ms per frame extend / shadow at
// 1 / 2 / 3 / 4 / 6 steps: 562 / 530 / 523 / 523 / 537 and 321 / 301 / 295 / 295 / 301; short walks lose (two steps: C2 -2 %,
// C3 -1 %: the later steps run at few lanes) and keep one.
// A round's further node steps are LEAN ones (traverse.h, node_step_fast: pop, test, push; a lane at the floor of its tree, or on the
// literal divisions, waits for the round's first step): C4 +2 %, c4xl +2.6 % (profiles/r04p_ab_c4xl_fast_box_test_lean_vs_full.log).  A scene
// whose node coordinates leave the guarded range of the division-free box test walks EVERY ray on the literal divisions, which the
// lean steps do not carry — they would all sit idle: such a scene gets FULL further steps (PBRS_FEAT_FULL_STEPS, pbrs_upload_scene).
// (Round 3 measured "lean steps cost the out-of-cache scene 11 %" on c4xl: that scene had lost the division-free test over three
// vertex heights below 2^-20 — a bound since lowered to 2^-60, pbrs_gpu.hip — and with it everything built on it.)
#define PBRS_MORE_NODE_STEPS(walk, S, stk, cnt, NSTEPS, FULL)                      \
    do {                                                                           \
        _Pragma("unroll") for (uint32_t k_ = 1; k_ < (NSTEPS); ++k_) {             \
            PBRS_KP_LANE(5 + (k_ < 2u ? k_ : 2u), walk.mode == PBRS_WALK_NODE);    \
            if (walk.mode == PBRS_WALK_NODE) {                                     \
                if constexpr (FULL) walk.node_step(S, stk, cnt);                   \
                else walk.node_step_fast(S, stk, cnt);                             \
            }                                                                      \
        }                                                                          \
    } while (0)
#define PBRS_STEP_WALK(walk, S, stk, cnt, XFER_MIN, LEAF_MIN, NSTEPS, FULL)                                                \
    do {                                                                                                       \
        PBRS_KP_WAVE(0);                                                                                       \
        PBRS_KP_LANE(10, walk.mode == PBRS_WALK_NODE || walk.mode == PBRS_WALK_LEAF || walk.mode == PBRS_WALK_XFER); \
        const uint32_t nx = (uint32_t)__popcll(__ballot(walk.mode == PBRS_WALK_XFER));                         \
        if (nx) {                                                                                              \
            if (nx >= XFER_MIN || __ballot(walk.mode == PBRS_WALK_NODE || walk.mode == PBRS_WALK_LEAF) == 0) { \
                PBRS_PROBE_XFER_COUNT(cnt);                                                                    \
                PBRS_KP_WAVE(3);                                                                               \
                PBRS_KP_LANE(4, walk.mode == PBRS_WALK_XFER);                                                  \
                if (walk.mode == PBRS_WALK_XFER) walk.xfer_step(S, stk, cnt);                                   \
            }                                                                                                  \
        }                                                                                                      \
        PBRS_TT(1);                                                                                            \
        PBRS_PROBE_UTIL_COUNT(walk, cnt);                                                                      \
        PBRS_KP_LANE(5, walk.mode == PBRS_WALK_NODE);                                                          \
        if (__ballot(walk.mode == PBRS_WALK_NODE)) PBRS_KP_WAVE(11);                                           \
        if (walk.mode == PBRS_WALK_NODE) walk.node_step(S, stk, cnt);                                          \
        PBRS_MORE_NODE_STEPS(walk, S, stk, cnt, NSTEPS, FULL);                                                            \
        PBRS_TT(2);                                                                                            \
        const uint32_t nl = (uint32_t)__popcll(__ballot(walk.mode == PBRS_WALK_LEAF));                         \
        if (nl && (nl >= LEAF_MIN || __ballot(walk.mode == PBRS_WALK_NODE) == 0)) {                            \
            PBRS_PROBE_LEAF_COUNT(cnt);                                                                        \
            PBRS_KP_WAVE(8);                                                                                   \
            PBRS_KP_LANE(9, walk.mode == PBRS_WALK_LEAF);                                                      \
            walk.leaf_wave(S, cnt);                                                                            \
        }                                                                                                      \
        PBRS_TT(3);                                                                                            \
    } while (0)
// Work fetch of the persistent traversal kernels.  A wave owns a private range [cur, end) of queue items and hands
// them to its idle lanes without touching memory; only when the range is empty does its first idle lane take a new
// chunk with one atomicAdd.  A single device-wide head word serialises at ~10 ns per atomic, which capped the kernels
// at chunk / 10 ns items per second, so the queue is cut into PBRS_WORK_HEADS contiguous segments, each with its own
// head word on its own 128-byte line.  A block starts on segment blockIdx.x % 8 — blocks are dealt round-robin to
// the 8 XCDs, so neighbouring rays stay within one XCD's L2 — and moves on to the next segment when its own is empty.
// All values are wave-uniform.
#ifndef PBRS_WORK_HEADS
#define PBRS_WORK_HEADS 8u
#endif
#define PBRS_WORK_HEAD_STRIDE 32u  // u32 words between head words
struct WaveWork {
    uint32_t cur, end, chunk;
    uint32_t head, tried;  // segment in use; segments found empty so far
    PD bool left() const { return tried < PBRS_WORK_HEADS || cur < end; }
};
PD WaveWork wave_work_init(uint32_t n) {
    // small queues (late bounces) get small chunks so that every wave still finds work
    uint32_t waves = gridDim.x * (blockDim.x >> 6);
    uint32_t chunk = n / (waves * 4u);
    chunk = chunk < 64u ? 64u : (chunk > PBRS_CHUNK_MAX ? PBRS_CHUNK_MAX : (chunk & ~63u));
    return WaveWork{0u, 0u, chunk, blockIdx.x % PBRS_WORK_HEADS, 0u};
}
PD uint32_t work_segment(uint32_t n, uint32_t h) {
    return h >= PBRS_WORK_HEADS ? n : (uint32_t)(((unsigned long long)n * h / PBRS_WORK_HEADS) & ~63ull);
}
// Lanes with `need` get an item index < n (returned; 0xffffffff = none).  Items left in the wave's range are handed
// out first; if they do not cover every idle lane a new chunk is taken in the same call, so a refill never leaves
// lanes idle while the queue still holds work.
PD uint32_t wave_fetch(WaveWork& w, bool need, uint32_t* heads, uint32_t n) {
    const uint64_t mask = __ballot(need);
    const uint32_t want = (uint32_t)__popcll(mask), rank = lane_prefix(mask);
    const uint32_t avail = w.end - w.cur;
    uint32_t idx = (need && rank < avail) ? w.cur + rank : 0xffffffffu;
    if (want <= avail) {
        w.cur += want;
        return idx;
    }
    w.cur = w.end;
    const int leader = __ffsll((unsigned long long)mask) - 1;
    while (w.tried < PBRS_WORK_HEADS) {  // ends: every pass either returns or retires one segment
        const uint32_t seg_lo = work_segment(n, w.head), seg_hi = work_segment(n, w.head + 1);
        uint32_t base = 0;
        if ((int)(threadIdx.x & 63u) == leader) base = atomicAdd(heads + w.head * PBRS_WORK_HEAD_STRIDE, w.chunk);
        base = __shfl(base, leader, 64);
        const uint32_t len = seg_hi - seg_lo;
        const uint32_t lo = seg_lo + (base < len ? base : len), hi = seg_lo + (base + w.chunk < len ? base + w.chunk : len);
        if (base + w.chunk >= len) {  // nothing beyond this chunk in the segment
            w.head = (w.head + 1u) % PBRS_WORK_HEADS;
            w.tried++;
        }
        if (lo == hi) continue;
        const uint32_t rest = want - avail, got = hi - lo;
        if (need && rank >= avail && rank - avail < got) idx = lo + (rank - avail);
        w.cur = lo + (rest < got ? rest : got);
        w.end = hi;
        break;
    }
    return idx;
}
// Rays a wide-walk kernel cannot take (traverse.h, PBRS_WALK_SLOW) go, whole, into a list that the binary-walk kernel of the
// same stage works off right after it: one atomic per wave and batch.
PD void wave_append_slow(bool slow, uint32_t item, uint32_t* list, uint32_t* count) {
    const uint64_t m = __ballot(slow);
    if (m == 0) return;
    const int leader = __ffsll((unsigned long long)m) - 1;
    uint32_t base = 0;
    if ((int)(threadIdx.x & 63u) == leader) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = __shfl(base, leader, 64);
    if (slow) list[base + lane_prefix(m)] = item;
}
#ifndef PBRS_WIDE_EXTEND_WAVES  // min waves per SIMD asked of the register allocator for the wide-walk kernels
#define PBRS_WIDE_EXTEND_WAVES 5
#endif
#ifndef PBRS_WIDE_SHADOW_WAVES
#define PBRS_WIDE_SHADOW_WAVES 5
#endif
#ifndef PBRS_WIDE_NODE_STEPS  // node steps per loop round of the wide walks in scenes with long walks
#define PBRS_WIDE_NODE_STEPS 2u
#endif
template <uint32_t ARITY, bool STATS, uint32_t FEAT>  // ARITY 0: the binary walks; 4: the walks over four-wide nodes (device/wide.h)
struct ClosestSel {
#ifdef PBRS_DEV_OVERRIDES
    typedef ClosestWalkW<FEAT> type;  // device/experimental/closest_wide.h
#endif
};
template <bool STATS, uint32_t FEAT>
struct ClosestSel<0u, STATS, FEAT> {
    typedef ClosestWalk<STATS, (FEAT & (PBRS_FEAT_ALL | PBRS_FEAT_LDS_TOP | PBRS_FEAT_EXTENT))> type;
};
template <uint32_t ARITY, bool STATS, uint32_t FEAT>
struct AnySel {
    typedef AnyWalkW<FEAT> type;
};
template <bool STATS, uint32_t FEAT>
struct AnySel<0u, STATS, FEAT> {
    typedef AnyWalk<STATS, (FEAT & (PBRS_FEAT_ALL | PBRS_FEAT_LDS_TOP))> type;
};
#define PBRS_WALK_ARITY(STATS, FEAT) ((STATS) ? 0u : ((FEAT) & PBRS_FEAT_WIDE) ? 4u : 0u)
#ifndef PBRS_NODE_STEPS_SHORT  // node steps per round in scenes with short walks: one (two, the second a lean one: C2 -2.3 %, C3 -3.2 %)
#define PBRS_NODE_STEPS_SHORT 1u
#endif
#define PBRS_WALK_NSTEPS(FEAT, ARITY) \
    (!((FEAT) & PBRS_FEAT_LONG_WALKS) ? ((ARITY) == 0u ? PBRS_NODE_STEPS_SHORT : 1u) : (ARITY) == 4u ? PBRS_WIDE_NODE_STEPS : PBRS_NODE_STEPS_LONG)
// PBRS_FEAT_LDS_SCENE: a random 16-byte read costs the CU's texture path 2 cycles per lane (a 32-byte node 129 cycles per wave, a
// 48-byte triangle ~190) and the LDS 0.37 (a node 46-58, a triangle 36-50: tools/microbench/gather_lds_coop.hip,
// profiles/r04_microbench_gather_lds_coop.log), at a third of the latency.  Where the arrays a walk reads fit next to the block's stack
// rows (pbrs_upload_scene: C2 / C3, 8 KB) every block copies them in once and the walk's scene view points into the copy: node,
// triangle, instance and shape fetches become ds_read_b128; what stays on the texture path is a ray's own record (fetched at its
// start, read again when it leaves an instance) and its result.
PD DevScene stage_scene(const DevScene& G, uint32_t* lds_base) {
    DevScene S = G;
    uint4* dst = reinterpret_cast<uint4*>(__builtin_assume_aligned(lds_base + G.lds_off_words, 16));
    const uint32_t n_nodes = G.lds_nodes * (uint32_t)(sizeof(pbrs_node) / 16), n_tv = G.lds_tris * (uint32_t)(sizeof(pbrs_tri_verts) / 16),
                   n_inst = G.lds_inst * (uint32_t)(sizeof(pbrs_instance) / 16), n_shapes = G.lds_shapes * (uint32_t)(sizeof(pbrs_shape) / 16);
    const uint4* src = reinterpret_cast<const uint4*>(G.nodes);
    for (uint32_t i = threadIdx.x; i < n_nodes; i += PBRS_TRAVERSAL_BLOCK) dst[i] = src[i];
    S.nodes = reinterpret_cast<const pbrs_node*>(dst);
    dst += n_nodes;
    src = reinterpret_cast<const uint4*>(G.tv);
    for (uint32_t i = threadIdx.x; i < n_tv; i += PBRS_TRAVERSAL_BLOCK) dst[i] = src[i];
    S.tv = reinterpret_cast<const pbrs_tri_verts*>(dst);
    dst += n_tv;
    src = reinterpret_cast<const uint4*>(G.inst);
    for (uint32_t i = threadIdx.x; i < n_inst; i += PBRS_TRAVERSAL_BLOCK) dst[i] = src[i];
    S.inst = reinterpret_cast<const pbrs_instance*>(dst);
    dst += n_inst;
    src = reinterpret_cast<const uint4*>(G.shapes);
    for (uint32_t i = threadIdx.x; i < n_shapes; i += PBRS_TRAVERSAL_BLOCK) dst[i] = src[i];
    S.shapes = reinterpret_cast<const pbrs_shape*>(dst);
    __syncthreads();
    return S;
}
static_assert(sizeof(pbrs_node) % 16 == 0 && sizeof(pbrs_tri_verts) % 16 == 0 && sizeof(pbrs_instance) % 16 == 0 && sizeof(pbrs_shape) % 16 == 0, "stage_scene copies 16-byte pieces");

// PBRS_FEAT_LDS_TOP: scenes whose arrays do not fit as a whole but whose TLAS does (C5: 130 instances, 259 nodes, 8 KB) get the head of the
// node array staged — a ray of such a scene tests some thirty TLAS boxes, five of a BLAS.
PD DevScene stage_top(const DevScene& G, uint32_t* lds_base) {
    DevScene S = G;
    uint4* dst = reinterpret_cast<uint4*>(__builtin_assume_aligned(lds_base + G.lds_off_words, 16));
    const uint4* src = reinterpret_cast<const uint4*>(G.nodes);
    const uint32_t n16 = G.lds_nodes * (uint32_t)(sizeof(pbrs_node) / 16);
    for (uint32_t i = threadIdx.x; i < n16; i += PBRS_TRAVERSAL_BLOCK) dst[i] = src[i];
    S.nodes_top = reinterpret_cast<const pbrs_node*>(dst);
    __syncthreads();
    return S;
}

// Persistent: every wave keeps pulling rays from the queue until it is empty; a lane whose walk ends is
// handed a new ray at the next refill, the walks of the other lanes continue where they were.
// `indirect` (binary-walk kernels working off a slow list): the queue positions to trace, `count` of them.
template <bool STATS, uint32_t FEAT>
__global__ void __launch_bounds__(256, STATS ? 3 : (FEAT & PBRS_FEAT_WIDE) ? PBRS_WIDE_EXTEND_WAVES : (FEAT & PBRS_FEAT_SHADING_CHECK) ? PBRS_TRAV_WAVES : PBRS_LEAN_EXTEND_WAVES)
    k_extend(DevScene G, PathState st, uint32_t set, const uint32_t* count, uint32_t n_direct, uint32_t* next, GlobalCounters* gc, const uint32_t* indirect,
             uint32_t* slow_list, uint32_t* slow_count, uint32_t split) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    const DevScene S = (!STATS && (FEAT & PBRS_FEAT_LDS_SCENE)) ? stage_scene(G, lds_stack) : (!STATS && (FEAT & PBRS_FEAT_LDS_TOP)) ? stage_top(G, lds_stack) : G;
    constexpr uint32_t ARITY = PBRS_WALK_ARITY(STATS, FEAT);
    constexpr bool WIDE = ARITY != 0u;
    const uint32_t n = count ? *count : n_direct;
    const float4* q0 = st.q[set][0];
    const float4* q1 = st.q[set][1];
    LaneStack stk{lds_stack + threadIdx.x, q0, q1, 0u};
    Cnt<STATS> cnt;
    cnt.init();
    uint32_t nrays = 0, nhit = 0;
    typename ClosestSel<ARITY, STATS, (FEAT & (PBRS_FEAT_ALL | PBRS_FEAT_LDS_TOP | PBRS_FEAT_EXTENT))>::type walk;
    walk.mode = PBRS_WALK_IDLE;
    PBRS_KP_DECL(walk);
    PBRS_TT_DECL;
    uint32_t item = 0;  // queue position of the lane's ray; bit 31: the path record's "after a specular bounce" flag (queues stay below 2^28)
    WaveWork work = wave_work_init(n);
    for (;;) {
        uint64_t live = __ballot(walk.mode == PBRS_WALK_NODE || walk.mode == PBRS_WALK_LEAF || walk.mode == PBRS_WALK_XFER);
        if ((uint32_t)__popcll(live) < S.refill_below) {
            PBRS_KP_WAVE(1);
            if constexpr (WIDE) {
                wave_append_slow(walk.mode == PBRS_WALK_SLOW, item & 0x7fffffffu, slow_list, slow_count);
                if (walk.mode == PBRS_WALK_SLOW) walk.mode = PBRS_WALK_IDLE;
            }
            if (walk.mode == PBRS_WALK_DONE) {  // finished walks are retired in batches, at refill time
                walk.finish(cnt);  // a walk that ended inside an instance: its candidate meets the best hit here
                const Hit& h = walk.best;
                nhit += h.inst != 0xffffffffu ? 1u : 0u;
                // k_shade rebuilds the Interaction from (t, inst, prim): the barycentrics are recomputed there, as the
                // reference's intersect does for the winning primitive
                // the 4th word names the hit's shading class (pbrs_upload_scene: DevScene::inst_class), 0 for a miss
                uint32_t cls = (S.n_classes > 1u && h.inst != 0xffffffffu) ? S.inst[h.inst].pad[0] : 0u;
                if (split) {
                    // The path integrator's queue of a scene with one shading class, split by what src/pathintegrator.rs does with
                    // the result (class 1 = kept for k_shade, 0 = dropped; k_class_count / _scan / _scatter then list the kept
                    // positions, in queue order).  A hit is kept: a surface with lobes is shaded (:31-71), an emitter adds its
                    // emission and the empty light estimate and ends.  A miss that sees the environment (the first bounce or after a
                    // specular one, :19-22) is kept: it adds the environment and ends.  Any other miss adds nothing (`break` at
                    // :25-27 with nothing before it) and is dropped — in an open scene most of a bounce's queue; so is a miss of the
                    // first bounce under a black environment: L = 0 + 1 * 0 there, which is what k_raygen left in L.  (After a
                    // specular bounce under a black environment beta may be infinite: beta * 0 is kept for k_shade to add.)
                    cls = 1u;
                    if (h.inst == 0xffffffffu) {
                        // (the path record's flag rides in `item` since the ray was fetched: reading the record again here put a
                        // memory round trip into every refill of an open scene)
                        const bool specular = (item >> 31) != 0u;
                        cls = (specular || ((split & 2u) && S.has_env != 0u)) ? 1u : 0u;
                    }
                }
                const uint32_t pos = item & 0x7fffffffu;
                if (!split || cls) st_stream(&st.hit[pos], make_float4(h.t, __uint_as_float(h.inst), __uint_as_float(h.prim), __uint_as_float(split ? 0u : cls)));
                if (S.n_classes > 1u || split) st.cls[pos] = (uint8_t)cls;  // what the class sort reads: 1 byte per path instead of a 16-byte record
                walk.mode = PBRS_WALK_IDLE;
            }
            if (work.left()) {
                uint32_t idx = wave_fetch(work, walk.mode == PBRS_WALK_IDLE, next, n);
                if (idx != 0xffffffffu) {
                    if (indirect) idx = indirect[idx];
                    stk.item = idx;
                    const float4 a = q0[idx], b = q1[idx];
                    item = idx | (__float_as_uint(a.w) & 0x80000000u);
                    walk.start(S, xyz(a), xyz(b), pn_inf(), stk);
                    nrays++;
                    PBRS_KP_LANE(2, true);
                }
                if constexpr (WIDE) walk.scan_wave(S, stk);
                else walk.scan_wave(S, cnt);
                live = __ballot(walk.mode == PBRS_WALK_NODE || walk.mode == PBRS_WALK_LEAF || walk.mode == PBRS_WALK_XFER);
            }
            if (live == 0) {
                if constexpr (WIDE) {
                    walk.forget_reciprocals();
                    if (__ballot(walk.mode == PBRS_WALK_SLOW || walk.mode == PBRS_WALK_DONE)) continue;  // rays that ended in the refill itself
                }
                break;
            }
        }
        PBRS_TT(0);
        PBRS_STEP_WALK(walk, S, stk, cnt, PBRS_EXT_XFER_MIN, PBRS_EXT_LEAF_MIN, PBRS_WALK_NSTEPS(FEAT, ARITY), ((FEAT & PBRS_FEAT_FULL_STEPS) != 0u));
        if constexpr (WIDE) walk.forget_reciprocals();
    }
    flush_counters<STATS>(cnt, gc, true, nrays, nhit);
    if (!STATS) PBRS_KP_FLUSH(0, walk);
    if (!STATS) PBRS_TT_FLUSH(0);
}

PD float power_heuristic2(float f_pdf, float g_pdf) {  // src/directlighting.rs:224-232, BETA = 2, nf = ng = 1
    float f = 1.0f * f_pdf;
    float g = 1.0f * g_pdf;
    return pn_powi(f, 2) / (pn_powi(f, 2) + pn_powi(g, 2));
}

// ---- shade -----------------------------------------------------------------------------------------------------
// SPEC: what the scene's materials and lights allow the stage to leave out (derived at upload, pbrs_upload_scene):
//   PBRS_SHADE_LAMBERT        every lobe is an untextured Lambertian DiffuseReflect, at most one per material
//   PBRS_SHADE_LIGHT_SPHERE / _TRIANGLE   every area light has that shape
// Code a scene cannot reach still costs the loads that decide not to take it (a lobe's kind, a light's shape kind), the
// registers of its longest path and the instructions around it: C2 (Lambert + triangle lights) shades in 87.7 instead of
// 110.5 ms per frame, C4 (Lambert + sphere lights: 96 VGPRs, five waves per SIMD) in 116.7 instead of 150.6.
#define PBRS_SHADE_LAMBERT 1u
#define PBRS_SHADE_LIGHT_SPHERE 2u
#define PBRS_SHADE_LIGHT_TRIANGLE 4u
//   PBRS_SHADE_FOURIER        the other way round: some material is a Fourier BSDF (device/fourier.h), whose code only the
//                             kernels with this bit contain (its f64 series sums and Newton loops are long and register-hungry)
#define PBRS_SHADE_FOURIER 8u
//   PBRS_SHADE_FOURIER_ONLY   (with PBRS_SHADE_FOURIER) every vertex the launch meets is on a Fourier material, whose one lobe is the
//                             Fourier BSDF: the launch over that class of a class-major queue (pbrs_gpu.hip)
#define PBRS_SHADE_FOURIER_ONLY 16u
//   PBRS_SHADE_LDS_RECORDS    the scene's instance records, analytic shapes, materials, lobes and lights are copied into the block's LDS
//   PBRS_SHADE_LDS_TRIS       ... and its triangle vertex and shading records (scenes of a few KB)
// at kernel start (stage_shade_scene): a vertex's ~25 record fetches — the instance's two matrices, the triangle's seven vectors, material,
// lobe and light — are gathers that cost the CU's texture path a cycle or two per lane each (k_shade's texture data unit was 0.93-0.95
// busy on C2 / C3) and the LDS a third of that, at a third of the latency (tools/microbench/gather_lds_coop.hip).  Chosen per scene by what fits
// (pbrs_upload_scene): C2 / C3 both, C4 (a million triangles, six instances) the records.
#define PBRS_SHADE_LDS_RECORDS 32u
#define PBRS_SHADE_LDS_TRIS 64u
#ifndef PBRS_FOURIER_AK_ROWS
#define PBRS_FOURIER_AK_ROWS 48u  // terms of a luminance series kept in LDS per lane; longer series are recomputed where they are consumed
#endif
PD uint4* stage_array(uint4* dst, const void* src_, uint32_t n16) {
    const uint4* src = reinterpret_cast<const uint4*>(src_);
    for (uint32_t i = threadIdx.x; i < n16; i += 256u) dst[i] = src[i];
    return dst + n16;
}
template <uint32_t SPEC>
PD DevScene stage_shade_scene(const DevScene& G, uint32_t* lds_base) {
    DevScene S = G;
    uint4* dst = reinterpret_cast<uint4*>(__builtin_assume_aligned(lds_base, 16));
    if (SPEC & PBRS_SHADE_LDS_RECORDS) {
        S.inst = reinterpret_cast<const pbrs_instance*>(dst);
        dst = stage_array(dst, G.inst, G.n_inst * (uint32_t)(sizeof(pbrs_instance) / 16));
        S.shapes = reinterpret_cast<const pbrs_shape*>(dst);
        dst = stage_array(dst, G.shapes, G.n_shapes * (uint32_t)(sizeof(pbrs_shape) / 16));
        S.mats = reinterpret_cast<const pbrs_material*>(dst);
        dst = stage_array(dst, G.mats, G.n_mats * (uint32_t)(sizeof(pbrs_material) / 16));
        S.bxdfs = reinterpret_cast<const pbrs_bxdf*>(dst);
        dst = stage_array(dst, G.bxdfs, G.n_bxdfs * (uint32_t)(sizeof(pbrs_bxdf) / 16));
        S.alights = reinterpret_cast<const pbrs_area_light*>(dst);
        dst = stage_array(dst, G.alights, G.n_area * (uint32_t)(sizeof(pbrs_area_light) / 16));
        S.dlights = reinterpret_cast<const pbrs_delta_light*>(dst);
        dst = stage_array(dst, G.dlights, G.n_delta * (uint32_t)(sizeof(pbrs_delta_light) / 16));
    }
    if (SPEC & PBRS_SHADE_LDS_TRIS) {
        S.tv = reinterpret_cast<const pbrs_tri_verts*>(dst);
        dst = stage_array(dst, G.tv, G.n_tris * (uint32_t)(sizeof(pbrs_tri_verts) / 16));
        S.ts = reinterpret_cast<const pbrs_tri_shade*>(dst);
        dst = stage_array(dst, G.ts, G.n_tris * (uint32_t)(sizeof(pbrs_tri_shade) / 16));
    }
    __syncthreads();
    return S;
}
static_assert(sizeof(pbrs_material) % 16 == 0 && sizeof(pbrs_bxdf) % 16 == 0 && sizeof(pbrs_area_light) % 16 == 0 && sizeof(pbrs_delta_light) % 16 == 0 &&
                  sizeof(pbrs_tri_shade) % 16 == 0,
              "stage_shade_scene copies 16-byte pieces");
template <uint32_t INTEG, bool TEX, uint32_t SPEC>
__global__ void __launch_bounds__(256, (SPEC & 8u) ? PBRS_FOURIER_SHADE_WAVES : PBRS_SHADE_WAVES) k_shade(DevScene G, PathState st, RenderConst rc, uint32_t bounce, const uint32_t* count, uint32_t n_direct,
                                              uint32_t* count_out, uint32_t* nee_queue, unsigned long long* nee_shadow_count, uint32_t sorted, const uint2* range) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_shade_scene[];
    // per-hit lobe lists of textured materials (Bsdf::hit_lobe / hit_albedo); absent from the untextured instantiation
    __shared__ uint32_t s_hit_lobe[TEX ? PBRS_MAX_BXDFS * 256 : 1];
    __shared__ float s_hit_albedo[TEX ? 3 * PBRS_MAX_BXDFS * 256 : 1];
    // the luminance series of a sampled Fourier direction pair, one column per lane (device/fourier.h, sample_fourier): 48 KB — three
    // blocks of the Fourier-only variants, which carry no texture lists, still fit a CU's LDS
    __shared__ float s_fourier_ak[(SPEC & 16u) ? PBRS_FOURIER_AK_ROWS * 256 : 1];
#ifdef PBRS_PROBE_SHADE
    unsigned long long probe_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long probe_t = __builtin_amdgcn_s_memtime();
#endif
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n = count ? *count : n_direct;
    if (range) {  // one launch per group of shading classes: this one's lanes of the class-major permutation (k_class_scatter)
        const uint2 r = *range;
        i += r.x;
        n = r.y;
        if (blockIdx.x * blockDim.x + r.x >= n) return;  // the grid is sized for the whole queue
    } else if (blockIdx.x * blockDim.x >= n) {
        return;  // ... of the pass: a block beyond the bounce's queue has nothing to shade, nothing to compact
    }
    const DevScene S = (SPEC & (PBRS_SHADE_LDS_RECORDS | PBRS_SHADE_LDS_TRIS)) ? stage_shade_scene<SPEC>(G, lds_shade_scene) : G;
    bool valid = i < n;
    bool alive = false, cast0 = false, cast1 = false;
    uint32_t slot = 0;
    ShadowRay sr0, sr1;  // the rays to cast, kept until their queue positions are known
    sr0.o = sr0.d = sr1.o = sr1.d = gray(0.0f);
    sr0.t_max = sr1.t_max = -1.0f;
    f3 L_vis = gray(0.0f);  // a lone shadow ray: the path's radiance if it turns out unoccluded
    // the path's next record, kept until the compaction below has given it a queue position
    f3 next_o = gray(0.0f), next_d = gray(0.0f), next_beta = gray(0.0f);
    uint64_t next_rng = 0;
    uint32_t next_spec = 0;
    if (valid) {
        const uint32_t set = bounce & 1u;
        // several shading classes: lanes take the paths in class order (k_class_sort), so that a wave runs one material's code; a
        // queue k_extend split into kept and dropped paths: lanes take the kept positions (class 1 of a class-major order)
        const uint32_t src = sorted ? st.perm[i] : i;
        const float4 r0 = ld_stream(&st.q[set][0][src]), r1 = ld_stream(&st.q[set][1][src]), r2 = ld_stream(&st.q[set][2][src]), rh = ld_stream(&st.hit[src]);
        slot = __float_as_uint(r0.w) & PBRS_SLOT_MASK;
        const float4 rl = st.L[slot];
        f3 o = xyz(r0), d = xyz(r1);
        f3 beta = xyz(r2);
        f3 L = xyz(rl);
        float post_w = rl.w;  // the direct integrator's 1 / mass rides next to the radiance
        const uint64_t rng_in = ((uint64_t)__float_as_uint(r2.w) << 32) | (uint64_t)__float_as_uint(r1.w);
        Hit h;
        h.t = rh.x;
        h.inst = __float_as_uint(rh.y);
        h.prim = __float_as_uint(rh.z);
        bool has_hit = h.inst != 0xffffffffu;
        bool specular_bounce = (__float_as_uint(r0.w) >> 31) != 0;
        const pbrs_material* mat = nullptr;
        if (has_hit) mat = S.mats + S.inst[h.inst].material;
        // The direct-lighting integrator (INTEG 1, src/directlighting.rs:14-56) runs on the same stage: bounce 0 is
        // direct_lighting_integrator, bounce 1 the debug integrator behind one specular lobe.  Its result is
        // direct + (S * f) * (1 / mass): f travels in the beta columns, 1 / mass in the flags column (as float bits).
        bool emitter_hit = false;
        float post = 1.0f;
        if (INTEG == PBRS_INTEGRATOR_MATERIALS) {  // material_visualizer, src/directlighting.rs:234-271
            if (has_hit) {
                float r = 0.0f, g = 0.0f, b = 0.0f;  // pal[index], :235-246; Color::rgb = u8 / 255.0 (color.rs:51-53)
                switch (mat->vis_class) {
                    case 0: r = 232.0f, g = 207.0f, b = 59.0f; break;
                    case 1: r = 124.0f, g = 188.0f, b = 126.0f; break;
                    case 2: r = 30.0f, g = 68.0f, b = 176.0f; break;
                    case 3: r = 15.0f, g = 142.0f, b = 205.0f; break;
                    case 4: r = 44.0f, g = 180.0f, b = 172.0f; break;
                    case 5: r = 216.0f, g = 39.0f, b = 252.0f; break;
                    case 6: r = 143.0f, g = 112.0f, b = 252.0f; break;
                    default: break;
                }
                L = mk3(r / 255.0f, g / 255.0f, b / 255.0f);
                if (mat->vis_class == 7u) L = gray(0.3f);
                if (mat->vis_class == 8u) L = gray(0.9f);
            } else {
                const int parity = (int)((uint32_t)pn_f32_to_i32(pn_floor(d.x * 50.0f)) + (uint32_t)pn_f32_to_i32(pn_floor(d.y * 50.0f)));
                L = gray(parity % 2 == 0 ? 0.9f : 0.7f);
            }
            emitter_hit = true;  // nothing below runs for this lane
        } else if (INTEG == PBRS_INTEGRATOR_NORMALS) {  // normal_visualizer, src/directlighting.rs:273-289
            if (!has_hit) {
                L = env_eval(S, d);
            } else {
                const Isect is = reconstruct_isect(S, h, o, d);
                const pbrs_bxdf& lobe0 = S.bxdfs[mat->first_bxdf];
                const pbrs_bxdf& vis = S.bxdfs[mat->vis_bxdf - 1u];
                f3 albedo = gray(0.0f);  // `let (_, albedo) = mtl.scatter(-ray.dir, &hit)`; black where scatter is todo!()
                switch (mat->vis_class) {
                    case 8:  // Lambertian, material/src/lib.rs:163-177
                        albedo = vis.tex ? tex_value(S, (vis.tex & ~PBRS_BXDF_TEX_DROP_IF_BLACK) - 1u, is.u, is.v, is.pos) : ld3(vis.albedo);
                        break;
                    case 7:  // Metal :192-199: Fresnel::conductor(eta, k).eval(|n . wi|), wi = -ray.dir as it is
                        albedo = fresnel_eval(lobe0, pn_abs(dot(is.normal, -d)));
                        break;
                    case 5:  // Mirror :224-228
                    case 0:  // Plastic :427-432
                        albedo = ld3(vis.albedo);
                        break;
                    case 4: {  // Dielectric :246-264
                        const float ior = lobe0.eta[1];
                        const f3 wi = hat(-d);
                        const float ndw = dot(is.normal, wi);
                        const bool inside = ndw < 0.0f;
                        const f3 outward = inside ? -is.normal : is.normal;
                        const float ratio = inside ? ior : 1.0f / ior;
                        const float cosine = inside ? -ndw : ndw;
                        f3 wt;
                        const bool transmits = refract3(outward, wi, ratio, wt);
                        float reflect_pr = 1.0f;
                        albedo = ld3(lobe0.albedo);  // self.reflect
                        if (transmits) {
                            float r0 = (1.0f - ior) / (1.0f + ior);  // schlick, :477-481
                            r0 = r0 * r0;
                            reflect_pr = r0 + (1.0f - r0) * pn_powi(1.0f - cosine, 5);
                            albedo = ld3(vis.albedo);  // self.transmit
                        }
                        uint64_t rng = rng_in;
                        if (pn_rng_f32(&rng) < reflect_pr) albedo = ld3(lobe0.albedo);
                        break;
                    }
                    default: break;  // DiffuseLight: black (:282-286); Glossy / Uber / Substrate / Fourier: todo!()
                }
                L = (albedo + is.normal) * 0.5f;
            }
            emitter_hit = true;
        } else if (INTEG == PBRS_INTEGRATOR_PATH) {
            if (bounce == 0 || specular_bounce) {  // pathintegrator.rs:19-22
                f3 e = has_hit ? ld3(mat->emission) : env_eval(S, d);
                L = L + cmul(beta, e);
            }
        } else if (bounce == 0) {
            if (!has_hit) {
                L = env_eval(S, d);  // scene.eval_env_light(ray), directlighting.rs:45
            } else if (!is_black(ld3(mat->emission))) {
                L = ld3(mat->emission);  // :27-28
                emitter_hit = true;
            }
        } else {
            post = post_w;
            if (!has_hit) L = L + cmul(env_eval(S, d), beta) * post;  // :54, then spec_refl * f * pr.mass().weak_recip() (:37)
        }
        PBRS_SHADE_MARK(0);  // queue + state loads, emission
        if (has_hit && !emitter_hit) {
            uint64_t rng = rng_in;
#ifdef PBRS_ABL_NO_RECON  // timing-only ablation build: skips the Interaction rebuild, results are wrong
            Isect is;
            is.pos = o + h.t * d;
            is.normal = mk3(0.0f, 1.0f, 0.0f);
            is.wo = -d;
            is.tangent = mk3(1.0f, 0.0f, 0.0f);
#else
            Isect is = reconstruct_isect(S, h, o, d);
#endif
            Bsdf bs = bsdf_new_frame(is, S.bxdfs + mat->first_bxdf, mat->n_bxdfs);
            bs.lam = (SPEC & PBRS_SHADE_LAMBERT) != 0u;
            if ((SPEC & PBRS_SHADE_LAMBERT) && mat->n_bxdfs) bs.a0 = ld3(S.bxdfs[mat->first_bxdf].albedo);
            const FourierView fourier_view{S.fourier, S.tex_floats, S.tex_words, (SPEC & PBRS_SHADE_FOURIER_ONLY) ? s_fourier_ak + threadIdx.x : nullptr,
                                           (SPEC & PBRS_SHADE_FOURIER_ONLY) ? PBRS_FOURIER_AK_ROWS : 0u, (SPEC & PBRS_SHADE_FOURIER_ONLY) != 0u};
            bs.fourier = (SPEC & PBRS_SHADE_FOURIER) ? &fourier_view : nullptr;
            if (TEX && (mat->flags & PBRS_MATERIAL_TEXTURED)) {
                // `mtl.bxdfs_at(&hit)` with non-Solid textures (material/src/lib.rs:180-184, :317-365): evaluate each
                // lobe's colour at (uv, pos) once, keep the lobes the material pushes for this hit
                uint32_t* hl = s_hit_lobe + threadIdx.x;
                float* ha = s_hit_albedo + threadIdx.x;
                uint32_t kept = 0;
                for (uint32_t k = 0; k < mat->n_bxdfs; ++k) {
                    const pbrs_bxdf& lb = bs.lobes[k];
                    f3 colour = ld3(lb.albedo);
                    if (lb.tex) {
                        colour = tex_value(S, (lb.tex & ~PBRS_BXDF_TEX_DROP_IF_BLACK) - 1u, is.u, is.v, is.pos);
                        if ((lb.tex & PBRS_BXDF_TEX_DROP_IF_BLACK) && is_black(colour)) continue;
                    }
                    hl[kept * 256u] = k;
                    ha[(3u * kept) * 256u] = colour.x;
                    ha[(3u * kept + 1u) * 256u] = colour.y;
                    ha[(3u * kept + 2u) * 256u] = colour.z;
                    ++kept;
                }
                bs.n = kept;
                bs.hit_lobe = hl;
                bs.hit_albedo = ha;
            }
            PBRS_SHADE_MARK(1);  // interaction rebuild + frame

            // uniform_sample_one_light, directlighting.rs:58-99
            uint32_t num_lights = S.n_delta + S.n_area + S.has_env;
#ifdef PBRS_ABL_NO_NEE  // timing-only ablation build (tools/ablate.sh): skips next-event estimation, results are wrong
            if (num_lights > 0) {
                for (int k = 0; k < 5; ++k) pn_rng_f32(&rng);
                num_lights = 0;
            }
#endif
            if (num_lights > 0) {
                float light_pdf = 1.0f / (float)num_lights;
                float uidx = pn_rng_f32(&rng);
                uint32_t chosen = (uint32_t)(uidx * (float)num_lights);
                if (chosen > num_lights - 1) chosen = num_lights - 1;
                float lu = pn_rng_f32(&rng), lv = pn_rng_f32(&rng);
                float su = pn_rng_f32(&rng), sv = pn_rng_f32(&rng);
                float scale = 1.0f / light_pdf;
                const f3 wo_l = world_to_local(bs, is.wo);  // once for eval, pdf and the MIS sample below
                f3 c1 = gray(0.0f), c2 = gray(0.0f);
                ShadowRay v1, v2;
                v1.t_max = -1.0f;
                v2.t_max = -1.0f;
                v1.o = v1.d = v2.o = v2.d = gray(0.0f);
                uint32_t mode;
                if (chosen < S.n_delta) {  // estimate_direct_delta_light :101-153
                    mode = 1;
                    f3 li, wi;
                    ShadowRay vis;
                    delta_sample_incident(S.dlights[chosen], is, li, wi, vis);
                    f3 bv = bsdf_eval_l(bs, wo_l, wi) * pn_abs(dot(is.normal, wi));
                    if (!(is_black(li) || is_black(bv))) {
                        c1 = cmul(bv, li) * 1.0f * pn_weak_recip(1.0f);
                        v1 = vis;
                    }
                } else if (chosen >= S.n_delta && chosen < S.n_area) {  // Q6 guard; estimate_direct_area_light :155-222
                    mode = 0;
                    const pbrs_area_light& Lt = S.alights[chosen - S.n_delta];
                    const uint32_t lkind = (SPEC & PBRS_SHADE_LIGHT_SPHERE) ? (uint32_t)PBRS_SHAPE_SPHERE
                                           : (SPEC & PBRS_SHADE_LIGHT_TRIANGLE) ? (uint32_t)PBRS_SHAPE_TRIANGLE : Lt.shape_kind;
                    f3 li, wi;
                    float lpdf;
                    ShadowRay vis;
                    area_sample_incident(Lt, lkind, is, lu, lv, li, wi, lpdf, vis);
                    PBRS_SHADE_MARK(2);  // NEE: draws + light sample + its pdf
                    if (lpdf > 0.0f && !is_black(li)) {
                        f3 bv = bsdf_eval_l(bs, wo_l, wi) * pn_abs(dot(is.normal, wi));
                        float spdf = bsdf_pdf_l(bs, wo_l, wi);
                        if (!is_black(bv) && spdf > 0.0f) {
                            float weight = power_heuristic2(lpdf, spdf);
                            c1 = cmul(bv, li) * weight * pn_weak_recip(lpdf);
                            v1 = vis;
                        }
                    }
                    PBRS_SHADE_MARK(3);  // NEE term 1: BSDF eval + pdf + MIS weight
                    f3 f2, wi2;
                    ProbD pr2;
                    bsdf_sample_l(bs, wo_l, su, sv, f2, wi2, pr2);
                    f2 = f2 * pn_abs(dot(is.normal, wi2));
                    if (!(is_black(f2) || !(pr2.v > 0.0f))) {
                        f3 le;
                        float lpdf2;
                        ShadowRay vis2;
                        if (area_radiance_to(Lt, lkind, is, wi2, le, lpdf2, vis2)) {
                            if (!(is_black(le) || lpdf2 <= 0.0f)) {
                                float weight = pr2.is_mass ? 1.0f : power_heuristic2(pr2.v, lpdf2);
                                c2 = (weight * cmul(f2, le)) * pn_weak_recip(pr2.v);
                                v2 = vis2;
                            }
                        }
                    }
                } else {  // :80-96 environment
                    mode = 2;
                    f3 f2, wi2;
                    ProbD pr2;
                    bsdf_sample_l(bs, wo_l, su, sv, f2, wi2, pr2);
                    spawn_ray(is, wi2, v1.o, v1.d);
                    v1.t_max = pn_inf();
                    float ac = pn_abs(dot(wi2, is.normal));
                    float wr = pn_weak_recip(pr2.v);
                    c1 = cmul(env_eval(S, v1.d), f2) * ac * wr;  // scene.eval_env_light(incident_ray), :90-95
                    c2 = cmul(gray(0.0f), f2) * ac * wr;  // Color::black() * f * ..., the occluded arm
                }
                PBRS_SHADE_MARK(4);  // NEE term 2: BSDF sample + light intersection + pdf
                cast0 = v1.t_max >= 0.0f;
                cast1 = v2.t_max >= 0.0f;
                if (cast0 && cast1) {
                    // two rays (area light, both MIS terms alive): directlighting.rs:193 and :219 add up in k_nee_resolve
                    st.nee[0][slot] = pack4(c1, scale);
                    st.nee[1][slot] = pack4(c2, post);
                    st.nee[2][slot] = pack4(beta, 0.0f);
                } else if (cast0 || cast1) {
                    // one ray: the lane that traces it finishes the estimate (directlighting.rs:193 / :219 / :90-96, then
                    // :98 and pathintegrator.rs:35), so both outcomes are evaluated here with the reference's operations
                    f3 cr = cast0 ? c1 : c2;
                    f3 one_v, one_o;
                    if (mode == 0) {
                        one_o = gray(0.0f);
                        one_v = gray(0.0f) + cr;
                    } else if (mode == 1) {
                        one_o = gray(0.0f);
                        one_v = cr;
                    } else {
                        one_o = c2;
                        one_v = cr;
                    }
                    f3 add_v = cmul(beta, one_v * scale);
                    f3 add_o = cmul(beta, one_o * scale);
                    if (INTEG != PBRS_INTEGRATOR_PATH) {
                        add_v = add_v * post;
                        add_o = add_o * post;
                    }
                    // pathintegrator.rs:35 is one f32 add per channel either way: the occluded outcome goes into L here,
                    // the unoccluded one travels with the ray and replaces it (k_shadow) — nothing below touches L
                    L_vis = L + add_v;
                    L = L + add_o;
                } else {
                    // nothing to test: the estimate is black; pathintegrator.rs:35 still adds beta * (black * n)
                    f3 z = cmul(beta, gray(0.0f) * scale);
                    if (INTEG != PBRS_INTEGRATOR_PATH) z = z * post;
                    L = L + z;
                }
                sr0 = v1;
                sr1 = v2;
            }

            if (INTEG != PBRS_INTEGRATOR_PATH) {
                // directlighting.rs:31-41: one level of perfect specular reflection / refraction
                f3 f, wi;
                ProbD pr;
                if (bounce == 0 && bsdf_sample_specular(bs, is.wo, f, wi, pr)) {
                    f3 no, nd;
                    spawn_ray(is, wi, no, nd);
                    alive = true;
                    next_o = no;
                    next_d = nd;
                    next_beta = f;
                    next_rng = rng;
                    post_w = pn_weak_recip(pr.v);
                }
            } else {
            PBRS_SHADE_MARK(5);  // NEE bookkeeping (cast flags, both outcomes)
            // pathintegrator.rs:46-71
            float r0 = pn_rng_f32(&rng), r1 = pn_rng_f32(&rng);
            f3 f, wi;
            ProbD pr;
            bsdf_sample(bs, -d, r0, r1, f, wi, pr);
            if (!(is_black(f) || pr.v == 0.0f)) {
                specular_bounce = pr.is_mass;
                beta = cmul(beta, f) * dot(wi, is.normal) * pn_recip(pr.v);  // Q9: no abs
                f3 no, nd;
                spawn_ray(is, wi, no, nd);
                bool cont = true;
                if (bounce > 3) {
                    float q = pn_max(1.0f - luminance(beta), 0.05f);
                    if (pn_rng_f32(&rng) < q)
                        cont = false;
                    else
                        beta = beta * pn_recip(1.0f - q);
                }
                if (cont && bounce + 1 < rc.max_depth) {
                    alive = true;
                    next_o = no;
                    next_d = nd;
                    next_beta = beta;
                    next_rng = rng;
                    next_spec = specular_bounce ? 0x80000000u : 0u;
                }
            }
            }
        }
        st.L[slot] = pack4(L, post_w);
        PBRS_SHADE_MARK(6);  // bounce: BSDF sample, beta, spawn, roulette, state stores
    }
    // Stream compaction of the three outputs.  A hot queue tail serialises at ~10 ns per atomic on gfx950, so the
    // counts are first summed per block through LDS and the tails are bumped by TWO atomics per block: the
    // next-bounce tail, and one 64-bit add carrying the nee-path count (low half) and the shadow-ray count (high).
    __shared__ uint32_t s_cnt[4][4];   // [wave][alive, nee, cast0, cast1]
    __shared__ uint32_t s_base[4];     // block bases: alive, nee, shadow
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const bool both = cast0 && cast1;
    const uint32_t lone = (!both && (cast0 || cast1)) ? 0x40000000u : 0u;
    const uint64_t m_alive = __ballot(alive), m_nee = __ballot(both), m_c0 = __ballot(cast0), m_c1 = __ballot(cast1);
    if (lane == 0) {
        s_cnt[wave][0] = (uint32_t)__popcll(m_alive);
        s_cnt[wave][1] = (uint32_t)__popcll(m_nee);
        s_cnt[wave][2] = (uint32_t)__popcll(m_c0);
        s_cnt[wave][3] = (uint32_t)__popcll(m_c1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t_alive = 0, t_nee = 0, t_sh = 0;
        for (int w = 0; w < 4; ++w) {
            t_alive += s_cnt[w][0];
            t_nee += s_cnt[w][1];
            t_sh += s_cnt[w][2] + s_cnt[w][3];
        }
        uint32_t a = 0u;
        unsigned long long packed = 0ull;
        if (t_alive | t_nee | t_sh) {  // both in flight before either is waited for: the whole block sits out their latency
            a = atomicAdd(count_out, t_alive);
            packed = atomicAdd(nee_shadow_count, ((unsigned long long)t_sh << 32) | (unsigned long long)t_nee);
        }
        s_base[0] = a;
        s_base[1] = (uint32_t)packed;
        s_base[2] = (uint32_t)(packed >> 32);
    }
    __syncthreads();
    uint32_t b_alive = s_base[0], b_nee = s_base[1], b_sh = s_base[2];
    for (uint32_t w = 0; w < wave; ++w) {
        b_alive += s_cnt[w][0];
        b_nee += s_cnt[w][1];
        b_sh += s_cnt[w][2] + s_cnt[w][3];
    }
    if (alive) {
        const uint32_t j = b_alive + lane_prefix(m_alive), out = (bounce + 1u) & 1u;
        st_stream(&st.q[out][0][j], pack4(next_o, slot | next_spec));
        st_stream(&st.q[out][1][j], pack4(next_d, (uint32_t)next_rng));
        st_stream(&st.q[out][2][j], pack4(next_beta, (uint32_t)(next_rng >> 32)));
    }
    if (both) nee_queue[b_nee + lane_prefix(m_nee)] = slot;
    if (cast0) {
        const size_t j = (size_t)b_sh + lane_prefix(m_c0);
        st_stream(&st.sr[0][j], pack4(sr0.o, sr0.t_max));
        st_stream(&st.sr[1][j], pack4(sr0.d, slot | lone));
        if (lone) st_stream(&st.sr[2][j], pack4(L_vis, 0.0f));
    }
    if (cast1) {
        const size_t j = (size_t)b_sh + s_cnt[wave][2] + lane_prefix(m_c1);
        st_stream(&st.sr[0][j], pack4(sr1.o, sr1.t_max));
        st_stream(&st.sr[1][j], pack4(sr1.d, slot | 0x80000000u | lone));
        if (lone) st_stream(&st.sr[2][j], pack4(L_vis, 0.0f));
    }
#ifdef PBRS_PROBE_SHADE
    PBRS_SHADE_MARK(7);  // compaction + queue / record writes
    if ((threadIdx.x & 63u) == 0 && __ballot(valid) != 0 && (blockIdx.x % 61u) == 0)  // a sample of the blocks
        for (int k = 0; k < 8; ++k) atomicAdd(&g_shade_probe[((SPEC & PBRS_SHADE_LAMBERT) ? 0 : 8) + k], probe_acc[k]);  // [0..7] the Lambert variants, [8..15] the others
#endif
}

// ---- class sort ------------------------------------------------------------------------------------------------
// A wave of k_shade that holds paths on different kinds of material runs every kind's code one after the other with most
// lanes masked (C3: 30 of 64 lanes active on average).  Scenes with several shading classes (materials with the same
// lobe signature, pbrs_upload_scene) therefore get their bounce queues ordered by class before shading: a stable
// counting sort of the hit records' class word within tiles of PBRS_SORT_TILE queue positions (one block per tile), which
// writes the permutation k_shade reads its paths through.  The records themselves stay where they are (k_shade gathers
// 4 x 16 bytes per path; inside a class the order is the queue's, so the gathers stay near-sequential) and tiles keep
// the spatial order of the queue at large.  The result of a path does not depend on which lane shades it.
#define PBRS_SORT_TILE 16384u
#define PBRS_MAX_CLASSES 16u
// Class sizes of the queue positions [base, end) into s_tot[] (zeroed, block-wide).  Sixteen ballots per 64 paths, lane c of a
// wave keeping class c's count, and one LDS add per lane at the end: one LDS atomic per PATH on sixteen words at most was
// what the counting pass spent its time on (0.39 ms per 200 M paths reading bytes, as much as when it read 16-byte records).
template <uint32_t NC>  // classes that occur (16, or 2 for a queue split into kept and dropped paths)
PD void tile_class_histogram(const uint8_t* cls, uint32_t base, uint32_t end, uint32_t* s_tot) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t mine = 0;
    for (uint32_t it = base; it < end; it += 256u) {  // block-uniform trip count
        const uint32_t i = it + threadIdx.x;
        const uint32_t c = i < end ? ((uint32_t)cls[i] & (PBRS_MAX_CLASSES - 1u)) : 0xffffffffu;
#pragma unroll
        for (uint32_t k = 0; k < NC; ++k) {
            const uint32_t nk = (uint32_t)__popcll(__ballot(c == k));
            if (lane == k) mine += nk;
        }
    }
    if (lane < NC && mine) atomicAdd(&s_tot[lane], mine);
}
__global__ void __launch_bounds__(256) k_class_sort(PathState st, const uint32_t* count, uint32_t n_direct) {
    const uint32_t n = count ? *count : n_direct;
    const uint32_t base = blockIdx.x * PBRS_SORT_TILE;
    if (base >= n) return;
    const uint32_t end = base + PBRS_SORT_TILE < n ? base + PBRS_SORT_TILE : n;
    __shared__ uint32_t s_tot[PBRS_MAX_CLASSES];     // class sizes in the tile, then their running write offsets
    __shared__ uint32_t s_wave[4][PBRS_MAX_CLASSES];  // per iteration: class counts of each wave
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (threadIdx.x < PBRS_MAX_CLASSES) s_tot[threadIdx.x] = 0u;
    __syncthreads();
    tile_class_histogram<PBRS_MAX_CLASSES>(st.cls, base, end, s_tot);
    __syncthreads();
    if (threadIdx.x == 0) {  // exclusive prefix over the classes
        uint32_t run = base;
        for (uint32_t c = 0; c < PBRS_MAX_CLASSES; ++c) {
            const uint32_t k = s_tot[c];
            s_tot[c] = run;
            run += k;
        }
    }
    __syncthreads();
    for (uint32_t it = base; it < end; it += 256u) {  // every thread of the block takes part in the barriers
        const uint32_t i = it + threadIdx.x;
        const bool valid = i < end;
        const uint32_t cls = valid ? ((uint32_t)st.cls[i] & (PBRS_MAX_CLASSES - 1u)) : 0xffffffffu;
        uint32_t rank = 0;
        for (uint32_t c = 0; c < PBRS_MAX_CLASSES; ++c) {
            const uint64_t m = __ballot(cls == c);
            if (cls == c) rank = lane_prefix(m);
            if (lane == 0) s_wave[wave][c] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        if (valid) {
            uint32_t pos = s_tot[cls] + rank;
            for (uint32_t w = 0; w < wave; ++w) pos += s_wave[w][cls];
            st.perm[pos] = i;
        }
        __syncthreads();
        if (threadIdx.x < PBRS_MAX_CLASSES) s_tot[threadIdx.x] += s_wave[0][threadIdx.x] + s_wave[1][threadIdx.x] + s_wave[2][threadIdx.x] + s_wave[3][threadIdx.x];
        __syncthreads();
    }
}

// Class-major variant, for scenes whose Lambertian class is shaded by the Lambert variant of k_shade (pbrs_gpu.hip, run_pass):
// the same stable counting sort, over the whole queue — per-tile histograms (k_class_count), their prefix over the tiles per
// class (k_class_scan, one block), then the scatter (k_class_scatter).  Classes come in the order 0, 1, ... with class `last`
// moved to the end, so that "all classes but `last`" is one range of lanes too: class_range[PBRS_MAX_CLASSES].
template <uint32_t NC>
__global__ void __launch_bounds__(256) k_class_count(PathState st, const uint32_t* count, uint32_t n_direct) {
    const uint32_t n = count ? *count : n_direct;
    const uint32_t base = blockIdx.x * PBRS_SORT_TILE;
    if (base >= n) return;
    const uint32_t end = base + PBRS_SORT_TILE < n ? base + PBRS_SORT_TILE : n;
    __shared__ uint32_t s_tot[PBRS_MAX_CLASSES];
    if (threadIdx.x < PBRS_MAX_CLASSES) s_tot[threadIdx.x] = 0u;
    __syncthreads();
    tile_class_histogram<NC>(st.cls, base, end, s_tot);
    __syncthreads();
    if (threadIdx.x < PBRS_MAX_CLASSES) st.tile_hist[blockIdx.x * PBRS_MAX_CLASSES + threadIdx.x] = s_tot[threadIdx.x];
}
// One block of PBRS_MAX_CLASSES waves: wave c turns class c's per-tile counts into offsets inside the class.
// `acc` (optional): adds (paths of class `last`, paths in all) to two counters — what a queue split kept (pbrs_gpu.hip, split_decision).
__global__ void __launch_bounds__(64 * PBRS_MAX_CLASSES) k_class_scan(PathState st, const uint32_t* count, uint32_t n_direct, uint32_t last, unsigned long long* acc) {
    const uint32_t n = count ? *count : n_direct;
    const uint32_t n_tiles = (n + PBRS_SORT_TILE - 1u) / PBRS_SORT_TILE;
    const uint32_t c = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    __shared__ uint32_t s_total[PBRS_MAX_CLASSES];
    uint32_t carry = 0;
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += 64u) {  // wave-uniform trip count
        const uint32_t t = t0 + lane;
        const uint32_t v = t < n_tiles ? st.tile_hist[t * PBRS_MAX_CLASSES + c] : 0u;
        uint32_t x = v;  // inclusive scan over the wave
        for (uint32_t off = 1; off < 64u; off <<= 1) {
            const uint32_t y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        if (t < n_tiles) st.tile_hist[t * PBRS_MAX_CLASSES + c] = carry + x - v;
        carry += __shfl(x, 63, 64);
    }
    if (lane == 0) s_total[c] = carry;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t k = 0; k < PBRS_MAX_CLASSES; ++k) {
            if (k == last) continue;
            st.class_range[k] = make_uint2(run, run + s_total[k]);
            run += s_total[k];
        }
        st.class_range[PBRS_MAX_CLASSES] = make_uint2(0u, run);  // every class but `last`
        if (last < PBRS_MAX_CLASSES) st.class_range[last] = make_uint2(run, run + s_total[last]);
        if (acc && last < PBRS_MAX_CLASSES) {
            atomicAdd(acc, (unsigned long long)s_total[last]);
            atomicAdd(acc + 1, (unsigned long long)n);
        }
    }
}
template <uint32_t NC>
__global__ void __launch_bounds__(256) k_class_scatter(PathState st, const uint32_t* count, uint32_t n_direct) {
    const uint32_t n = count ? *count : n_direct;
    const uint32_t base = blockIdx.x * PBRS_SORT_TILE;
    if (base >= n) return;
    const uint32_t end = base + PBRS_SORT_TILE < n ? base + PBRS_SORT_TILE : n;
    __shared__ uint32_t s_tot[PBRS_MAX_CLASSES];      // running write offset of each class for this tile
    __shared__ uint32_t s_wave[4][PBRS_MAX_CLASSES];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (threadIdx.x < PBRS_MAX_CLASSES) s_tot[threadIdx.x] = st.class_range[threadIdx.x].x + st.tile_hist[blockIdx.x * PBRS_MAX_CLASSES + threadIdx.x];
    __syncthreads();
    for (uint32_t it = base; it < end; it += 256u) {  // every thread of the block takes part in the barriers
        const uint32_t i = it + threadIdx.x;
        const bool valid = i < end;
        const uint32_t cls = valid ? ((uint32_t)st.cls[i] & (PBRS_MAX_CLASSES - 1u)) : 0xffffffffu;
        uint32_t rank = 0;
#pragma unroll
        for (uint32_t c = 0; c < NC; ++c) {
            const uint64_t m = __ballot(cls == c);
            if (cls == c) rank = lane_prefix(m);
            if (lane == 0) s_wave[wave][c] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        if (valid && !(NC == 2u && cls == 0u)) {  // two classes: a queue split into dropped (0) and kept (1) paths, only the kept are listed
            uint32_t pos = s_tot[cls] + rank;
            for (uint32_t w = 0; w < wave; ++w) pos += s_wave[w][cls];
            st.perm[pos] = i;
        }
        __syncthreads();
        if (threadIdx.x < NC) s_tot[threadIdx.x] += s_wave[0][threadIdx.x] + s_wave[1][threadIdx.x] + s_wave[2][threadIdx.x] + s_wave[3][threadIdx.x];
        __syncthreads();
    }
}

// ---- shadow ----------------------------------------------------------------------------------------------------
// One work item per shadow ray (persistent, same refill scheme as k_extend); writes one occlusion byte.
template <bool STATS, uint32_t FEAT>
__global__ void __launch_bounds__(256, STATS ? 3 : (FEAT & PBRS_FEAT_WIDE) ? PBRS_WIDE_SHADOW_WAVES : PBRS_SHADOW_WAVES)
    k_shadow(DevScene G, PathState st, const uint32_t* count, uint32_t* next, GlobalCounters* gc, const uint32_t* indirect, uint32_t* slow_list,
             uint32_t* slow_count) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    const DevScene S = (!STATS && (FEAT & PBRS_FEAT_LDS_SCENE)) ? stage_scene(G, lds_stack) : (!STATS && (FEAT & PBRS_FEAT_LDS_TOP)) ? stage_top(G, lds_stack) : G;
    constexpr uint32_t ARITY = PBRS_WALK_ARITY(STATS, FEAT);
    constexpr bool WIDE = ARITY != 0u;
    const uint32_t n = indirect ? count[0] : count[1];  // a slow list's length, or the high half of the packed (nee paths, shadow rays) counter
    LaneStack stk{lds_stack + threadIdx.x, st.sr[0], st.sr[1], 0u};
    Cnt<STATS> cnt;
    cnt.init();
    uint32_t nrays = 0;
    typename AnySel<ARITY, STATS, (FEAT & (PBRS_FEAT_ALL | PBRS_FEAT_LDS_TOP))>::type walk;
    walk.mode = PBRS_WALK_IDLE;
    PBRS_KP_DECL(walk);
    PBRS_TT_DECL;
    uint32_t item = 0, rec = 0;
    f3 l_vis = gray(0.0f);  // a lone ray's "unoccluded" outcome, fetched with the ray: read at retire time it put a memory round trip into every refill
    WaveWork work = wave_work_init(n);
    for (;;) {
        uint64_t live = __ballot(walk.mode == PBRS_WALK_NODE || walk.mode == PBRS_WALK_LEAF || walk.mode == PBRS_WALK_XFER);
        if ((uint32_t)__popcll(live) < S.refill_below_shadow) {
            PBRS_KP_WAVE(1);
            if constexpr (WIDE) {
                wave_append_slow(walk.mode == PBRS_WALK_SLOW, rec, slow_list, slow_count);
                if (walk.mode == PBRS_WALK_SLOW) walk.mode = PBRS_WALK_IDLE;
            }
            if (walk.mode == PBRS_WALK_DONE) {
                const bool occluded = walk.occluded;
                const uint32_t slot = item & PBRS_SLOT_MASK, r = item >> 31;
                if (item & 0x40000000u) {
                    // the path's only shadow ray: k_shade stored the occluded outcome in L and sent the other one along
                    if (!occluded) {
                        float* l = reinterpret_cast<float*>(st.L + slot);  // .w (the direct integrator's 1 / mass) stays
                        l[0] = l_vis.x;
                        l[1] = l_vis.y;
                        l[2] = l_vis.z;
                    }
                } else {
                    at(st.occ[r], slot) = occluded ? 1 : 0;
                }
                walk.mode = PBRS_WALK_IDLE;
            }
            if (work.left()) {
                uint32_t idx = wave_fetch(work, walk.mode == PBRS_WALK_IDLE, next, n);
                if (idx != 0xffffffffu) {
                    if (indirect) idx = indirect[idx];
                    rec = idx;
                    stk.item = idx;
                    const float4 q0 = st.sr[0][idx], q1 = st.sr[1][idx];
                    item = __float_as_uint(q1.w);
                    if (item & 0x40000000u) l_vis = xyz(st.sr[2][idx]);
                    walk.start(S, mk3(q0.x, q0.y, q0.z), mk3(q1.x, q1.y, q1.z), q0.w, stk);
                    nrays++;
                    PBRS_KP_LANE(2, true);
                }
                if constexpr (WIDE) walk.scan_wave(S, stk);
                else walk.scan_wave(S, cnt);
                live = __ballot(walk.mode == PBRS_WALK_NODE || walk.mode == PBRS_WALK_LEAF || walk.mode == PBRS_WALK_XFER);
            }
            if (live == 0) {
                if constexpr (WIDE) {
                    walk.forget_reciprocals();
                    if (__ballot(walk.mode == PBRS_WALK_SLOW || walk.mode == PBRS_WALK_DONE)) continue;
                }
                break;
            }
        }
        PBRS_TT(0);
        PBRS_STEP_WALK(walk, S, stk, cnt, PBRS_SHD_XFER_MIN, PBRS_SHD_LEAF_MIN, PBRS_WALK_NSTEPS(FEAT, ARITY), ((FEAT & PBRS_FEAT_FULL_STEPS) != 0u));
        if constexpr (WIDE) walk.forget_reciprocals();
    }
    flush_counters<STATS>(cnt, gc, true, nrays, 0u);
    if (!STATS) PBRS_KP_FLUSH(1, walk);
    if (!STATS) PBRS_TT_FLUSH(1);
}

// The radiance add of uniform_sample_one_light / path_integrator for paths whose estimate had to wait for
// visibility: Ld terms in the reference's order (directlighting.rs:193, :219), * n_lights (:98), * beta
// (pathintegrator.rs:35).
__global__ void __launch_bounds__(256) k_nee_resolve(PathState st, const uint32_t* queue, const uint32_t* count) {
  const uint32_t n = count[0];  // low half of the packed (nee paths, shadow rays) counter
  // A capped grid in strides (pbrs_gpu.hip, kStreamGridCap): the number of two-ray estimates is only known on the device — a
  // few % of a bounce's vertices — and a grid sized for the whole pass costs 0.17 ms per launch in blocks that find nothing
  // to do (an empty block is dispatched in ≈0.2 ns): 7 ms per frame of C4.
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    uint32_t slot = queue[i];
    // only area-light estimates cast two rays: Ld = term 1 (light sample, :193) + term 2 (BSDF sample, :219)
    bool occ0 = at(st.occ[0], slot) != 0, occ1 = at(st.occ[1], slot) != 0;
    const float4 n0 = st.nee[0][slot], n1 = st.nee[1][slot], n2 = st.nee[2][slot];
    f3 c1 = xyz(n0), c2 = xyz(n1);
    f3 one = gray(0.0f);
    if (!occ0) one = one + c1;
    if (!occ1) one = one + c2;
    f3 nb = xyz(n2);
    const float4 rl = st.L[slot];
    f3 L = xyz(rl);
    L = L + cmul(nb, one * n0.w) * n1.w;  // the post factor is 1 for the path integrator: x * 1 == x bit for bit
    st.L[slot] = pack4(L, rl.w);
  }
}

// ---- accumulate / finalize -----------------------------------------------------------------------------------------
// color_sum = color_sum + integrator(...) for strictly increasing sample index (src/main.rs:205)
// Also counts the samples whose radiance is not finite (pbrs_stats.invalid_samples): one atomic per wave that saw any.
__global__ void __launch_bounds__(256) k_accumulate(PathState st, float* sum, uint32_t n_pixels, uint32_t k_count, uint32_t chunk, uint32_t w,
                                                     uint32_t tiles8_per_row, unsigned long long* nonfinite) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pixels) return;
    f3 s = mk3(sum[p], sum[n_pixels + p], sum[2 * n_pixels + p]);
    uint32_t bad = 0;
    for (uint32_t k = 0; k < k_count; ++k) {
        const uint32_t slot = slot_of_sample(k, order_of_pixel(p, w, tiles8_per_row), n_pixels, k_count, chunk);
        const f3 l = xyz(st.L[slot]);
        bad += (pn_isfinite(l.x) && pn_isfinite(l.y) && pn_isfinite(l.z)) ? 0u : 1u;
        s = s + l;
    }
    if (__ballot(bad != 0u)) {
        for (int off = 32; off > 0; off >>= 1) bad += __shfl_down(bad, off, 64);
        if ((threadIdx.x & 63u) == 0) atomicAdd(nonfinite, (unsigned long long)bad);
    }
    sum[p] = s.x;
    sum[n_pixels + p] = s.y;
    sum[2 * n_pixels + p] = s.z;
}
// color_sum.scale_down_by(msaa*msaa) (src/main.rs:208, color.rs:90-95): * (1.0 / n as f32); planar -> row-major RGB
__global__ void __launch_bounds__(256) k_finalize(const float* sum, float* rgb, uint32_t n_pixels, float inv_spp) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pixels) return;
    rgb[3 * p] = sum[p] * inv_spp;
    rgb[3 * p + 1] = sum[n_pixels + p] * inv_spp;
    rgb[3 * p + 2] = sum[2 * n_pixels + p] * inv_spp;
}

// Instrumented renders only: the queue sizes of one pass, bounce by bounce, added to the render's totals (pbrs_stats).
// act[b] = paths entering bounce b >= 1 (bounce 0: the pass itself), ns[b] = (shadow rays << 32 | two-ray estimates) of bounce b.
__global__ void k_sum_bounce_counts(const uint32_t* act, const unsigned long long* ns, uint32_t n_pass, uint32_t n_bounces, unsigned long long* acc) {
    const uint32_t b = threadIdx.x;
    if (b >= n_bounces) return;
    const uint32_t at = b < PBRS_STATS_MAX_BOUNCES ? b : PBRS_STATS_MAX_BOUNCES - 1u;
    atomicAdd(acc + at, (unsigned long long)(b == 0 ? n_pass : act[b]));
    atomicAdd(acc + PBRS_STATS_MAX_BOUNCES + at, ns[b] >> 32);
}

// ---- parity-harness kernels --------------------------------------------------------------------------------------------
// The walks the pipeline runs for the scene, chosen per stage exactly as run_pass chooses them: WIDE_ANY = occlusion queries go through
// the four-wide any-hit walk of k_shadow (AnyWalkW, with its hand-off of refused rays to the binary walk); WIDE_CLOSEST = the
// four-wide closest-hit walk (developer builds only).  info[0] / info[1]: rays the wide any-hit / closest-hit walk refused.
template <bool WIDE_CLOSEST, bool WIDE_ANY>
__global__ void __launch_bounds__(256) k_intersect_rays(DevScene S, uint32_t n, const float4* origins, const float4* dirs, const float* tmax,
                                                       pbrs_hit_record* hits, uint8_t* occluded, uint32_t* info) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    LaneStack stk{lds_stack + threadIdx.x, origins, dirs, 0u};
    Cnt<false> cnt;
    // grid <= PBRS_PERSISTENT_BLOCKS; whole blocks stay in the loop together (the walks share work across a wave)
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        const bool active = i < n;
        f3 o = gray(0.0f), d = gray(1.0f);
        float t_max = 0.0f;
        if (active) {
            o = xyz(origins[i]);
            d = xyz(dirs[i]);
            t_max = tmax[i];
            stk.item = i;
        }
        if (hits) {
            Hit h;
            bool slow = false;
#ifdef PBRS_DEV_OVERRIDES
            if constexpr (WIDE_CLOSEST) tlas_closest_wide(S, active, o, d, t_max, stk, h, slow);
            else
#endif
            if (S.exact_extent) tlas_closest<false, PBRS_FEAT_EXTENT>(S, active, o, d, t_max, stk, h, cnt);  // as the scene's k_extend
            else tlas_closest<false>(S, active, o, d, t_max, stk, h, cnt);
            if (active) {
                pbrs_hit_record r;
                r.t = h.t;
                r.inst = h.inst;
                r.prim = 0;
                if (h.inst != 0xffffffffu && S.inst[h.inst].shape_kind == PBRS_SHAPE_MESH) r.prim = S.ts[h.prim].orig;
                r.b1 = h.b1;
                r.b2 = h.b2;
                hits[i] = r;
                if (slow) atomicAdd(info + 1, 1u);
            }
        }
        if (occluded) {
            bool slow = false;
            const bool occ = WIDE_ANY ? tlas_any_wide(S, active, o, d, t_max, stk, slow) : tlas_any<false>(S, active, o, d, t_max, stk, cnt);
            if (active) {
                occluded[i] = occ ? 1 : 0;
                if (slow) atomicAdd(info, 1u);
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_numeric_eval(uint32_t fn, uint32_t n, const float* x, const float* y, float* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = x[i], b = y ? y[i] : 0.0f, r = 0.0f;
    switch (fn) {
        case 0: r = pn_sin(a); break;
        case 1: r = pn_cos(a); break;
        case 2: r = pn_tan(a); break;
        case 3: r = pn_atan(a); break;
        case 4: r = pn_atan2(a, b); break;
        case 5: r = pn_acos(a); break;
        case 6: r = pn_exp(a); break;
        case 7: r = pn_ln(a); break;
        case 8: r = pn_hypot(a, b); break;
        case 9: r = a / b; break;
        case 10: r = pn_sqrt(a); break;
        case 11: r = pn_asin(a); break;
        case 12: r = pn_powi(a, (int)b); break;
        case 13: r = pn_fract(a); break;
        case 14: r = pn_floor(a); break;
        case 15: r = qdiv(-a, b, -(1.0f / b)); break;  // the box test's quotient (traverse.h): must equal a / b in its guarded range
        default: break;
    }
    out[i] = r;
}

__global__ void __launch_bounds__(256) k_export_rays(PathState st, uint32_t n, float* origins, float* dirs) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 o = st.q[0][0][i], d = st.q[0][1][i];  // as k_raygen leaves them: queue position = slot
    origins[3 * i] = o.x; origins[3 * i + 1] = o.y; origins[3 * i + 2] = o.z;
    dirs[3 * i] = d.x; dirs[3 * i + 1] = d.y; dirs[3 * i + 2] = d.z;
}
__global__ void __launch_bounds__(256) k_export_radiance(PathState st, uint32_t n, float* rgb) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 l = st.L[i];
    rgb[3 * i] = l.x; rgb[3 * i + 1] = l.y; rgb[3 * i + 2] = l.z;
}
