// pbrs_gpu.hip — implementation of include/pbrs_gpu.h for gfx950 (MI355X).
//
// Host-side driver of the wavefront pipeline in device/kernels.h: owns the HBM copies of the
// flattened scene, the SoA path state, the slot queues and the per-tile accumulator, and issues
//     raygen -> [extend -> shade -> shadow] x max_depth -> accumulate
// per pass of `samples_per_pass` sample indices, all on one HIP stream with no host round trip
// inside a tile: queue lengths stay on the device (kernels read them; empty blocks exit).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see include/pbrs_numeric.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <utility>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pbrs_gpu.h"
#include "device/kernels.h"

namespace {

constexpr uint32_t kBlock = 256;
constexpr uint32_t kMaxDepth = 64;
constexpr uint32_t kHeadWords = PBRS_WORK_HEADS * PBRS_WORK_HEAD_STRIDE;
// per bounce: act, ns (u64), the slow-list lengths of k_extend and k_shadow; then four sets of work heads (the two stages, and
// the binary-walk launches that work off their slow lists)
constexpr uint32_t kCounterWords = (7 + 4 * kHeadWords) * (kMaxDepth + 2);  // ... and k_extend's split counts (u64)
constexpr uint32_t kSlowGrid = 64;
#ifndef PBRS_SPLIT_KEEP_PERCENT
#define PBRS_SPLIT_KEEP_PERCENT 85u  // k_extend's queue split stays on where it keeps at most this share of a pass's rays for k_shade
#endif  // blocks of a binary-walk launch over a slow list (empty in nearly every launch)
constexpr uint32_t kStreamGridCap = 4096;  // blocks of k_nee_resolve, whose work is counted on the device (kernels.h)
constexpr uint32_t kPersistentBlocks = PBRS_PERSISTENT_BLOCKS;  // 256 CUs x up to 6 resident 256-thread blocks (VGPR/LDS permitting)
constexpr size_t kLdsBytesPerCU = 160 * 1024;

struct StageEvent {
    int stage;  // 0 raygen, 1 extend, 2 shade, 3 shadow, 4 accumulate
    hipEvent_t a, b;
};

// Developer overrides (A/B timing of kernel selection: tools/ab_env.sh) exist only in builds made with -DPBRS_DEV_OVERRIDES.
// The shipped library never reads the environment: a bench line must not depend on the box it ran on.
inline const char* dev_env(const char* name) {
#ifdef PBRS_DEV_OVERRIDES
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

}  // namespace

struct pbrs_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string error;

    // scene
    bool has_scene = false;
    std::vector<void*> scene_allocs;
    DevScene S{};
    uint32_t stack_depth = 0;

    // Two sets of per-pass working memory (path state, queues, counters) and two streams.  Consecutive passes of a render alternate
    // between the sets; a pass runs its bounces below overlap_from on the main stream and the rest — near-empty launches that end with
    // the latency of their longest walks — on a second, high-priority stream, beside the full kernels of the next pass's first bounces
    // (render_common, run_pass).  The fields below (state_mem .. counters) are the view of the set in use; use_pass_set swaps them.
    struct PassSet {
        void* state_mem = nullptr;
        size_t cap_slots = 0;
        PathState st{};
        uint32_t* neeq = nullptr;
        uint32_t* slow = nullptr;
        uint32_t* counters = nullptr;
        hipEvent_t accumulated = nullptr;  // the set's last k_accumulate has run (late stream): the set's memory is free for its next pass
        hipEvent_t late = nullptr;         // the set's pass has run its bounces below pbrs_ctx::overlap_from (main stream): the late stream takes over
        bool in_flight = false;            // `accumulated` has been recorded at least once
    } pass_set[2];
    int cur_set = 0;
    hipStream_t main_stream = nullptr;  // the stream of set 0: the context's own, or the caller's (pbrs_set_stream)
    hipStream_t second_stream = nullptr;  // the late stream: the context's own, high priority, ordered against the main one by events
    bool overlap_passes = true;  // PBRS_OVERLAP_PASSES=0 in developer builds: every pass on the main stream, as in rounds 1-3
    // The bounce from which a pass moves to the late stream (and the next pass starts behind it on the main one).  Same-box A/B lines in
    // profiles/r04m_ab_pass_overlap.log: scenes that live in every XCD's L2 gain most from bounce 2 on (C2 +1.9 %, C3 +1.5 %, C5 +3.3 %
    // against one stream; from 1: +1.3 / +1.2 / +2.7, from 3: 0 / +1.4 / +2.3), a scene that lives in the Infinity Cache from bounce 4 on
    // (C4 +1.0 %; from 3: +0.8, from 2 or 5: +0.2) — two passes' full kernels side by side cost it its cache residency (streams by pass,
    // every bounce overlapped: -2.6 %, profiles/r04l_ab_pass_overlap_streams_by_pass.log).
    uint32_t overlap_from = 2;
    // working set
    size_t cap_slots = 0, cap_pixels = 0;
    void* state_mem = nullptr;    // every per-path array of PathState, carved out of one allocation
    PathState st{};
    uint32_t* neeq = nullptr;     // cap_slots: slots of the paths whose estimate waits for two shadow rays
    uint32_t* counters = nullptr; // kCounterWords: act, ns (u64), extend work heads, shadow work heads
    float* sum = nullptr;         // 3 * cap_pixels, planar
    float* rgb_dev = nullptr;     // 3 * cap_pixels, row-major (for the host-output variant)
    GlobalCounters* gcnt = nullptr;  // [0] extend, [1] shadow
    unsigned long long* nonfinite = nullptr;  // samples of the current render whose radiance is not finite
    unsigned long long* bounce_acc = nullptr; // 2 x PBRS_STATS_MAX_BOUNCES: queue sizes per bounce of the current instrumented render

    // timing
    std::vector<StageEvent> events;
    size_t events_used = 0;
    std::vector<hipEvent_t> total_ev;  // 2
    pbrs_stats pending{};
    bool pending_counters = false, pending_times = false;
    bool textured = false;  // the uploaded scene evaluates non-Solid textures: k_shade<.., true>
    bool fourier = false;   // ... has a Fourier BSDF lobe: k_shade<.., true, PBRS_SHADE_FOURIER>
    uint32_t lambert_class = 0;    // shading class of the Lambert-only materials (0: none)
    uint32_t fourier_class = 0;    // shading class of the Fourier BSDF materials (0: none): k_shade's variants with that lobe run over it alone
    uint32_t light_spec = 0;       // PBRS_SHADE_LIGHT_*: every area light has that shape
    bool split_lambert = true;     // PBRS_SPLIT_LAMBERT=0 in the environment: one general k_shade launch for all classes (A/B timing)
    uint32_t shade_spec = 0;       // PBRS_SHADE_*: what k_shade<PATH> may leave out for this scene
    bool long_walks = false;       // a BLAS of PBRS_LONG_WALK_HEIGHT levels or more: the PBRS_FEAT_LONG_WALKS kernels
    bool full_steps = false;       // ... whose further node steps are full ones (PBRS_FEAT_FULL_STEPS): a scene outside the guarded range of the division-free box test
    uint64_t walk_bytes = 0;       // bytes of the arrays the walks read (nodes, wide nodes, triangle vertices, instances)
    uint32_t shade_lds = 0;        // PBRS_SHADE_LDS_*: what k_shade<PATH>'s untextured variants stage in LDS for this scene (kernels.h)
    size_t shade_lds_bytes = 0;
    bool lds_top = false;          // ... or only the TLAS does (an unscanned one): the PBRS_FEAT_LDS_TOP kernels
    size_t lds_top_bytes = 0;
    bool lds_scene = false;        // ... and they fit next to a block's stack rows: the PBRS_FEAT_LDS_SCENE kernels (S.lds_*)
    size_t lds_scene_bytes = 0;
    bool shadow_flat = false;      // k_shadow scans the TLAS leaves (up to PBRS_FLAT_TLAS_MAX_ANYHIT instances; k_extend: S.features)
    bool split_queue = true;       // k_extend splits the path integrator's queue of one-class scenes (shaded / terminal / dropped); PBRS_SPLIT_QUEUE=0 in developer builds
    // ... which pays where many paths are dropped (an open scene: C4 shades in 84 instead of 117 ms per frame) and costs where
    // none are (a closed box: the gathered records cost C2 4 %).  Decided once per uploaded scene, from the counts of the first
    // pass rendered with the path integrator: 0 = not yet, 1 = split, 2 = do not.  The image does not depend on it.
    int split_decision = 0;
    // The counts travel to the host through a pinned buffer behind an event that later passes poll (hipEventQuery): no call of the
    // render path waits for them, so pbrs_render_tile_device stays asynchronous (also on a caller's stream, pbrs_set_stream).
    unsigned long long* split_host = nullptr;  // pinned: (paths kept, paths in all) of the probed pass
    hipEvent_t split_ev = nullptr;
    bool split_probe_in_flight = false;
    bool wide_extend = false, wide_shadow = false;  // the stage runs the walks over four-wide nodes (device/wide.h), the binary walks after it for what they refuse
    uint32_t* slow = nullptr;      // 2 * cap_slots: queue positions a wide-walk kernel handed to the binary-walk kernel
    bool has_vis_records = false;  // every material names its pbrs_material::vis_bxdf record (normal_visualizer)
    bool sort_classes = true;      // PBRS_SORT_CLASSES=0 in the environment turns the class sort off (A/B timing)
    bool split_fourier = true;     // PBRS_SPLIT_FOURIER=0 in developer builds: one launch of the Fourier variants over every class, as in round 2
    uint64_t pending_closest = 0;
    pbrs_intersect_info last_intersect{};  // which walks the last pbrs_intersect_rays went through
};

namespace {

#define HIPCHK(ctx, expr)                                                                             \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            (ctx)->error = std::string(#expr) + ": " + hipGetErrorString(e_);                         \
            (void)hipGetLastError(); /* reported: must not resurface in a later call's hipGetLastError() */ \
            return PBRS_E_DEVICE;                                                                     \
        }                                                                                             \
    } while (0)

int fail(pbrs_ctx* c, int code, const char* msg) {
    c->error = msg;
    return code;
}

template <class T>
int upload(pbrs_ctx* c, const T* src, size_t n, const T** dst) {
    *dst = nullptr;
    size_t bytes = (n ? n : 1) * sizeof(T);
    void* p = nullptr;
    HIPCHK(c, hipMalloc(&p, bytes));
    c->scene_allocs.push_back(p);
    if (n) HIPCHK(c, hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
    *dst = static_cast<const T*>(p);
    return PBRS_OK;
}

void free_scene(pbrs_ctx* c) {
    for (void* p : c->scene_allocs) (void)hipFree(p);
    c->scene_allocs.clear();
    c->has_scene = false;
}

// The per-pass working memory in use (pbrs_ctx::pass_set): stores the context's view into the set it came from and loads set k,
// whose stream becomes the context's.
void use_pass_set(pbrs_ctx* c, int k) {
    pbrs_ctx::PassSet& o = c->pass_set[c->cur_set];
    o.state_mem = c->state_mem; o.cap_slots = c->cap_slots; o.st = c->st; o.neeq = c->neeq; o.slow = c->slow; o.counters = c->counters;
    c->cur_set = k;
    const pbrs_ctx::PassSet& n = c->pass_set[k];
    c->state_mem = n.state_mem; c->cap_slots = n.cap_slots; c->st = n.st; c->neeq = n.neeq; c->slow = n.slow; c->counters = n.counters;
    c->stream = c->main_stream;
}

void free_work(pbrs_ctx* c) {
    // capacities first: whatever happens below, no later call may take the old pointers for valid
    for (int k = 0; k < 2; ++k) {
        use_pass_set(c, k);
        c->cap_slots = 0;
        c->st = PathState{};
        if (c->state_mem) (void)hipFree(c->state_mem);
        c->state_mem = nullptr;
        c->neeq = nullptr;
        c->slow = nullptr;
    }
    use_pass_set(c, 0);
    c->cap_pixels = 0;
    if (c->sum) (void)hipFree(c->sum);
    if (c->rgb_dev) (void)hipFree(c->rgb_dev);
    c->sum = nullptr;
    c->rgb_dev = nullptr;
}

// Carves the per-path arrays of PathState out of one allocation; every array starts 256-byte aligned.  The new
// capacities are published only after every allocation has succeeded: a failure leaves the context without a working
// set (cap_* = 0, PathState cleared), never with pointers into freed memory.
int ensure_work(pbrs_ctx* c, size_t n_slots, size_t n_pixels) {
    if (n_slots > c->cap_slots) {
        c->cap_slots = 0;
        c->st = PathState{};
        c->neeq = nullptr;
        c->slow = nullptr;
        if (c->state_mem) (void)hipFree(c->state_mem);
        c->state_mem = nullptr;
        auto align = [](size_t b) { return (b + 255) / 256 * 256; };
        const size_t v16 = align(n_slots * sizeof(float4));
        // q[2][3], hit, L, nee[3]: one float4 per path each; sr[3]: two per path; occ: two bytes; nee queue: one word
        const size_t n_tiles = n_slots / PBRS_SORT_TILE + 1;
        const size_t sort_bytes = align(n_tiles * PBRS_MAX_CLASSES * sizeof(uint32_t)) + align((PBRS_MAX_CLASSES + 1) * sizeof(uint2));
        const size_t total = (6 + 1 + 1 + 3) * v16 + 3 * 2 * v16 + align(2 * n_slots) + align(n_slots * sizeof(uint32_t)) + align(n_slots * sizeof(uint32_t)) +
                             sort_bytes + align(n_slots) + align(2 * n_slots * sizeof(uint32_t));
        hipError_t e = hipMalloc(&c->state_mem, total);
        if (e != hipSuccess) {
            c->state_mem = nullptr;
            c->error = std::string("hipMalloc of the path state (") + std::to_string(total >> 20) + " MiB): " + hipGetErrorString(e);
            (void)hipGetLastError();  // reported: must not resurface in a later call's hipGetLastError()
            return PBRS_E_DEVICE;
        }
        char* p = static_cast<char*>(c->state_mem);
        auto take = [&](size_t bytes) { char* r = p; p += bytes; return r; };
        PathState s{};
        for (int set = 0; set < 2; ++set)
            for (int k = 0; k < 3; ++k) s.q[set][k] = reinterpret_cast<float4*>(take(v16));
        s.hit = reinterpret_cast<float4*>(take(v16));
        s.L = reinterpret_cast<float4*>(take(v16));
        for (int k = 0; k < 3; ++k) s.nee[k] = reinterpret_cast<float4*>(take(v16));
        for (int k = 0; k < 3; ++k) s.sr[k] = reinterpret_cast<float4*>(take(2 * v16));
        s.occ[0] = reinterpret_cast<uint8_t*>(take(align(2 * n_slots)));
        s.occ[1] = s.occ[0] + n_slots;
        c->neeq = reinterpret_cast<uint32_t*>(take(align(n_slots * sizeof(uint32_t))));
        s.perm = reinterpret_cast<uint32_t*>(take(align(n_slots * sizeof(uint32_t))));
        s.cls = reinterpret_cast<uint8_t*>(take(align(n_slots)));
        c->slow = reinterpret_cast<uint32_t*>(take(align(2 * n_slots * sizeof(uint32_t))));
        s.tile_hist = reinterpret_cast<uint32_t*>(take(align(n_tiles * PBRS_MAX_CLASSES * sizeof(uint32_t))));
        s.class_range = reinterpret_cast<uint2*>(take(align((PBRS_MAX_CLASSES + 1) * sizeof(uint2))));
        c->st = s;
        c->cap_slots = n_slots;
    }
    if (n_pixels > c->cap_pixels) {
        c->cap_pixels = 0;
        if (c->sum) (void)hipFree(c->sum);
        if (c->rgb_dev) (void)hipFree(c->rgb_dev);
        c->sum = nullptr;
        c->rgb_dev = nullptr;
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->sum), 3 * n_pixels * sizeof(float)));
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->rgb_dev), 3 * n_pixels * sizeof(float)));
        c->cap_pixels = n_pixels;
    }
    return PBRS_OK;
}

// wide-walk kernels: DevScene::wide_cap stack rows, and for closest hit one row per scanned TLAS leaf (entry distances)
size_t lds_bytes_wide(const pbrs_ctx* c, bool closest) {
    return (size_t)(c->S.wide_cap + (closest ? c->S.n_flat : 0u)) * kBlock * sizeof(uint32_t);
}
size_t lds_bytes(const pbrs_ctx* c) {
    size_t b = (size_t)(c->stack_depth) * kBlock * sizeof(uint32_t);
    if (const char* e = dev_env("PBRS_LDS_MIN")) b = b < (size_t)atol(e) ? (size_t)atol(e) : b;  // lowers the occupancy of the traversal kernels
    return b;
}

uint32_t auto_samples_per_pass(const pbrs_ctx* c, const pbrs_render_params* p);

int check_params(pbrs_ctx* c, const pbrs_camera* cam, const pbrs_render_params* p) {
    if (!cam || !p) return fail(c, PBRS_E_INVALID, "null camera or params");
    if (!c->has_scene) return fail(c, PBRS_E_NO_SCENE, "no scene uploaded");
    if (p->w == 0 || p->h == 0) return fail(c, PBRS_E_INVALID, "empty tile");
    if (p->band_count > 1) {
        if (p->band_rows == 0 || p->band_index >= p->band_count) return fail(c, PBRS_E_INVALID, "bad row-band parameters");
        uint64_t vr = p->h - 1;
        uint64_t last = p->y0 + ((vr / p->band_rows) * p->band_count + p->band_index) * (uint64_t)p->band_rows + vr % p->band_rows;
        if (p->x0 + p->w > cam->width || last >= cam->height) return fail(c, PBRS_E_INVALID, "row bands outside the film");
    } else if (p->x0 + p->w > cam->width || p->y0 + p->h > cam->height) {
        return fail(c, PBRS_E_INVALID, "tile outside the film");
    }
    if (p->strata_x == 0 || p->strata_y == 0) return fail(c, PBRS_E_INVALID, "zero strata");
    if (p->max_depth > kMaxDepth) return fail(c, PBRS_E_LIMIT, "max_depth above 64");
    if (p->integrator > PBRS_INTEGRATOR_NORMALS) return fail(c, PBRS_E_INVALID, "unknown integrator");
    if (p->integrator >= PBRS_INTEGRATOR_MATERIALS && (p->strata_x != 1 || p->strata_y != 1))
        return fail(c, PBRS_E_INVALID, "a visualiser takes one un-jittered ray per pixel (strata 1 x 1)");
    if (p->integrator == PBRS_INTEGRATOR_NORMALS && !c->has_vis_records)
        return fail(c, PBRS_E_INVALID, "the scene's materials carry no pbrs_material::vis_bxdf records");
    if ((uint64_t)p->w * p->h > (1ull << 28)) return fail(c, PBRS_E_LIMIT, "tile above 2^28 pixels");
    // records keep two flag bits next to the slot index, and 16-byte records are addressed with 32-bit element indices
    if ((uint64_t)p->w * p->h * auto_samples_per_pass(c, p) >= (1ull << 28)) return fail(c, PBRS_E_LIMIT, "tile x samples_per_pass above 2^28 paths");
    return PBRS_OK;
}

// Four-wide nodes over the binary subtree of inner node x (device/wide.h): the boxes of x's grandchildren — or of a child that is
// a leaf — in left-first order, with the three split axes that order them.  Returns the index of x's wide node in `out`.
// `nodes`: DevScene::nodes as uploaded (absolute links; checked: children come after their parent, so the recursion ends).
uint32_t build_wide(const std::vector<pbrs_node>& nodes, uint32_t x, std::vector<pbrs_wnode>& out, uint32_t level, uint32_t& levels) {
    const uint32_t me = (uint32_t)out.size();
    out.push_back(pbrs_wnode{});
    levels = std::max(levels, level + 1u);
    uint32_t slot_node[4] = {0, 0, 0, 0};
    uint32_t used = 0, info = nodes[x].b & 3u;
    const uint32_t child[2] = {x + 1u, nodes[x].a};
    for (uint32_t s = 0; s < 2; ++s) {
        const pbrs_node& ch = nodes[child[s]];
        if (ch.b & PBRS_LEAF_FLAG) {
            slot_node[2 * s] = child[s];
            used |= 1u << (2 * s);
        } else {
            info |= (ch.b & 3u) << (2 + 2 * s);
            slot_node[2 * s] = child[s] + 1u;
            slot_node[2 * s + 1] = ch.a;
            used |= 3u << (2 * s);
        }
    }
    pbrs_wnode w{};
    for (uint32_t k = 0; k < 4; ++k) {
        w.child[k] = PBRS_WREF_NONE;
        if (!((used >> k) & 1u)) {  // never passes the filter (device/wide.h)
            for (int a = 0; a < 3; ++a) w.lo[a][k] = PBRS_WIDE_UNUSED_PLANE, w.hi[a][k] = -PBRS_WIDE_UNUSED_PLANE;
            continue;
        }
        const pbrs_node& n = nodes[slot_node[k]];
        for (int a = 0; a < 3; ++a) {
            w.lo[a][k] = n.min[a];
            w.hi[a][k] = n.max[a];
        }
        w.child[k] = (n.b & PBRS_LEAF_FLAG) ? (PBRS_WREF_LEAF | slot_node[k]) : build_wide(nodes, slot_node[k], out, level + 1u, levels);
    }
    w.child[0] |= (info & 15u) << PBRS_WREF_AXIS_SHIFT;  // slots 0 and 2 are always in use
    w.child[2] |= ((info >> 4) & 3u) << PBRS_WREF_AXIS_SHIFT;
    out[me] = w;
    return me;
}

struct Timer {
    pbrs_ctx* c;
    bool on;
    int begin(int stage) {
        if (!on) return 0;
        if (c->events_used == c->events.size()) {
            StageEvent e{};
            if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return -1;
            c->events.push_back(e);
        }
        StageEvent& e = c->events[c->events_used];
        e.stage = stage;
        return hipEventRecord(e.a, c->stream) == hipSuccess ? 0 : -1;
    }
    int end() {
        if (!on) return 0;
        StageEvent& e = c->events[c->events_used++];
        return hipEventRecord(e.b, c->stream) == hipSuccess ? 0 : -1;
    }
};

RenderConst make_const(const pbrs_camera* cam, const pbrs_render_params* p) {
    RenderConst rc{};
    rc.cam = *cam;
    rc.x0 = p->x0; rc.y0 = p->y0; rc.w = p->w; rc.h = p->h;
    rc.strata_x = p->strata_x; rc.strata_y = p->strata_y;
    rc.max_depth = p->max_depth;
    rc.n_pixels = p->w * p->h;
    rc.band_rows = p->band_rows; rc.band_count = p->band_count; rc.band_index = p->band_index;
    rc.seed = p->seed;
    rc.integrator = p->integrator;
    // slot order of a pass (kernels.h, sample_of_slot): chunks of 4 K pixels, a multiple of the block and of the wave (C4:
    // 1156 Msamples/s with the sample index outermost, 1208-1211 with chunks of 256 ... 16 K pixels, 1202 with 64 K)
    rc.chunk_pixels = 4096u;
    if (const char* e = dev_env("PBRS_RAYGEN_CHUNK")) {  // developer override (A/B timing): 0 = sample index outermost, as in round 1
        const long v = std::atol(e);
        rc.chunk_pixels = v > 0 ? (uint32_t)v : 0xffffffffu;
    }
    if (rc.chunk_pixels > rc.n_pixels) rc.chunk_pixels = rc.n_pixels;  // one chunk: slot = k * P + pixel
    rc.tiles8_per_row = (p->w % 8u == 0u && p->h % 8u == 0u) ? p->w / 8u : 0u;
    if (const char* e = dev_env("PBRS_RAYGEN_TILES8")) {  // developer override (A/B timing)
        if (std::atoi(e) == 0) rc.tiles8_per_row = 0u;
    }
    return rc;
}

uint32_t auto_samples_per_pass(const pbrs_ctx* c, const pbrs_render_params* p) {
    uint64_t P = (uint64_t)p->w * p->h, spp = (uint64_t)p->strata_x * p->strata_y;
    uint64_t k = p->samples_per_pass;
    if (k == 0) {
        // ~240M paths in flight (~67 GB of the 288 GB HBM for path, hit, radiance, shadow-ray and nee records), less when
        // that would exceed a quarter of the memory currently free.  Every bounce is a handful of launches and a persistent
        // traversal kernel ends with the latency of its longest walks (hundreds of dependent node fetches on a deep BLAS): the
        // fewer, larger launches a frame is cut into, the less of it is spent draining (C2, 256 spp: 617 / 629 / 637
        // Msamples/s at 32 / 64 / 128 samples per pass; C4: 921 / 939 at 64 / 115).  Stays under the 2^28 paths of check_params.
        uint64_t target = 240ull << 20;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const uint64_t per_path = PBRS_STATE_BYTES_PER_PATH;  // path, hit, radiance, shadow-ray and nee records
            // what this context already holds for paths counts as available: the answer must not change between calls
            // (... in both of its pass sets: a set may take a quarter of the memory, the two of them half)
            const uint64_t held = (uint64_t)c->cap_slots + (uint64_t)c->pass_set[c->cur_set ^ 1].cap_slots;
            const uint64_t fit = ((uint64_t)free_b + held * per_path) / 4 / per_path;
            if (fit < target) target = fit < (4ull << 20) ? (4ull << 20) : fit;
        }
        k = P >= target ? 1 : target / P;
        if (k < 1) k = 1;
        if (k < spp) {  // passes of equal size: 256 spp at 240 per pass is 128 + 128, not 240 + 16
            const uint64_t passes = (spp + k - 1) / k;
            k = (spp + passes - 1) / passes;
        }
    }
    if (k > spp) k = spp;
    if (k < 1) k = 1;
    return (uint32_t)k;
}

// The traversal kernels are instantiated per scene-feature set (device/shapes.h PBRS_FEAT_*): one table entry per valid combination,
// filled at compile time; the instrumented variant exists for the full set only.  k_shadow never evaluates shading frames, so
// PBRS_FEAT_SHADING_CHECK does not select it.  `wide`: the walks over four-wide nodes (scenes with a scanned TLAS); `indirect` /
// `slow_*`: see kernels.h.
typedef void (*extend_fn_t)(DevScene, PathState, uint32_t, const uint32_t*, uint32_t, uint32_t*, GlobalCounters*, const uint32_t*, uint32_t*, uint32_t*, uint32_t);
typedef void (*shadow_fn_t)(DevScene, PathState, const uint32_t*, uint32_t*, GlobalCounters*, const uint32_t*, uint32_t*, uint32_t*);
constexpr uint32_t kFeatCombos = 256u;  // PBRS_FEAT_* bits 0 .. 7
constexpr bool extend_feat_ok(uint32_t f) {
    if ((f & PBRS_FEAT_FULL_STEPS) && !(f & PBRS_FEAT_LONG_WALKS)) return false;  // further node steps exist in the long-walk kernels only
    if ((f & PBRS_FEAT_LDS_SCENE) && (f & (PBRS_FEAT_WIDE | PBRS_FEAT_FULL_STEPS))) return false;  // a scene of a few KB
    if ((f & PBRS_FEAT_LDS_TOP) && (f & (PBRS_FEAT_LDS_SCENE | PBRS_FEAT_WIDE | PBRS_FEAT_FULL_STEPS | PBRS_FEAT_FLAT_TLAS))) return false;  // a TLAS too large to scan
    if (f & PBRS_FEAT_WIDE) {
#ifdef PBRS_DEV_OVERRIDES  // the four-wide closest-hit walk (device/experimental/closest_wide.h)
        return (f & PBRS_FEAT_FLAT_TLAS) && !(f & PBRS_FEAT_FULL_STEPS);
#else
        return false;
#endif
    }
    return true;
}
constexpr bool shadow_feat_ok(uint32_t f) {
    if (f & PBRS_FEAT_SHADING_CHECK) return false;
    if ((f & PBRS_FEAT_FULL_STEPS) && !(f & PBRS_FEAT_LONG_WALKS)) return false;
    if ((f & PBRS_FEAT_LDS_SCENE) && (f & (PBRS_FEAT_WIDE | PBRS_FEAT_FULL_STEPS))) return false;
    if ((f & PBRS_FEAT_LDS_TOP) && (f & (PBRS_FEAT_LDS_SCENE | PBRS_FEAT_WIDE | PBRS_FEAT_FULL_STEPS | PBRS_FEAT_FLAT_TLAS))) return false;
    if ((f & PBRS_FEAT_WIDE) && !(f & PBRS_FEAT_FLAT_TLAS)) return false;
    return true;
}
template <uint32_t F>
constexpr extend_fn_t extend_fn_of() {
    if constexpr (extend_feat_ok(F)) return &k_extend<false, F>;
    else return nullptr;
}
template <uint32_t F>
constexpr shadow_fn_t shadow_fn_of() {
    if constexpr (shadow_feat_ok(F)) return &k_shadow<false, F>;
    else return nullptr;
}
template <uint32_t... F>
const extend_fn_t* extend_table(std::integer_sequence<uint32_t, F...>) {
    static const extend_fn_t t[sizeof...(F)] = {extend_fn_of<F>()...};
    return t;
}
template <uint32_t... F>
const shadow_fn_t* shadow_table(std::integer_sequence<uint32_t, F...>) {
    static const shadow_fn_t t[sizeof...(F)] = {shadow_fn_of<F>()...};
    return t;
}
const extend_fn_t* extend_fns() { return extend_table(std::make_integer_sequence<uint32_t, kFeatCombos>{}); }
const shadow_fn_t* shadow_fns() { return shadow_table(std::make_integer_sequence<uint32_t, kFeatCombos>{}); }

int launch_extend(pbrs_ctx* c, bool stats, bool wide, uint32_t grid, size_t lds, uint32_t set, const uint32_t* count, uint32_t n_direct, uint32_t* heads,
                  const uint32_t* indirect, uint32_t* slow_list, uint32_t* slow_count, uint32_t split) {
    // k_extend scans the TLAS leaves only up to PBRS_FLAT_TLAS_MAX instances (S.features); the leaf copies may exist for
    // k_shadow alone, and the instrumented variant, which carries every feature, must then walk the tree like the timed one
    DevScene S = c->S;
    if (!(S.features & PBRS_FEAT_FLAT_TLAS)) S.n_flat = 0u;
    if (S.exact_extent) {  // the extent follows the tree: no leaf scan, no staging, one kernel each way (scenes with a ParallelQuad: no benchmark holds one)
        constexpr uint32_t kF = PBRS_FEAT_ANALYTIC | PBRS_FEAT_SHADING_CHECK | PBRS_FEAT_EXTENT;
        if (stats)
            hipLaunchKernelGGL((k_extend<true, kF>), dim3(grid), dim3(kBlock), lds, c->stream, S, c->st, set, count, n_direct, heads, c->gcnt, indirect, slow_list, slow_count, split);
        else
            hipLaunchKernelGGL((k_extend<false, kF>), dim3(grid), dim3(kBlock), lds, c->stream, S, c->st, set, count, n_direct, heads, c->gcnt, indirect, slow_list, slow_count, split);
        c->pending.kernel_features_extend = kF | (stats ? 0x80000000u : 0u);
        return PBRS_OK;
    }
    if (stats) {
        hipLaunchKernelGGL((k_extend<true, PBRS_FEAT_ALL>), dim3(grid), dim3(kBlock), lds, c->stream, S, c->st, set, count, n_direct, heads, c->gcnt, indirect, slow_list,
                           slow_count, split);
        c->pending.kernel_features_extend = PBRS_FEAT_ALL | 0x80000000u;
        return PBRS_OK;
    }
    uint32_t feat = (c->S.features & PBRS_FEAT_ALL) | (c->long_walks ? PBRS_FEAT_LONG_WALKS : 0u) | ((c->long_walks && c->full_steps) ? PBRS_FEAT_FULL_STEPS : 0u);
    if (wide) feat = (feat & ~PBRS_FEAT_FULL_STEPS) | PBRS_FEAT_WIDE;  // (developer builds; PBRS_FEAT_FLAT_TLAS is set: pbrs_upload_scene)
    if (c->lds_scene && !wide) {
        feat |= PBRS_FEAT_LDS_SCENE;
        lds += c->lds_scene_bytes;
    } else if (c->lds_top && !wide && !(feat & (PBRS_FEAT_FLAT_TLAS | PBRS_FEAT_FULL_STEPS))) {
        feat |= PBRS_FEAT_LDS_TOP;
        lds += c->lds_top_bytes;
    }
    const extend_fn_t fn = extend_fns()[feat];
    if (!fn) return fail(c, PBRS_E_DEVICE, "no k_extend instantiation for this scene's feature set");
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kBlock), lds, c->stream, S, c->st, set, count, n_direct, heads, c->gcnt, indirect, slow_list, slow_count, split);
    if (!indirect) c->pending.kernel_features_extend = feat;  // (a launch over a slow list is the stage's second kernel)
    return PBRS_OK;
}
int launch_shadow(pbrs_ctx* c, bool stats, bool wide, uint32_t grid, size_t lds, const uint32_t* count, uint32_t* heads, const uint32_t* indirect,
                  uint32_t* slow_list, uint32_t* slow_count) {
    if (stats) {
        hipLaunchKernelGGL((k_shadow<true, PBRS_FEAT_ANALYTIC | PBRS_FEAT_FLAT_TLAS>), dim3(grid), dim3(kBlock), lds, c->stream, c->S, c->st, count, heads, c->gcnt + 1,
                           indirect, slow_list, slow_count);
        c->pending.kernel_features_shadow = PBRS_FEAT_ANALYTIC | PBRS_FEAT_FLAT_TLAS | 0x80000000u;
        return PBRS_OK;
    }
    uint32_t feat = (c->S.features & PBRS_FEAT_ANALYTIC) | (c->shadow_flat ? PBRS_FEAT_FLAT_TLAS : 0u) | (c->long_walks ? PBRS_FEAT_LONG_WALKS : 0u) |
                    ((c->long_walks && c->full_steps) ? PBRS_FEAT_FULL_STEPS : 0u);
    if (wide) feat |= PBRS_FEAT_WIDE;  // (PBRS_FEAT_FLAT_TLAS is set: pbrs_upload_scene)
    if (c->lds_scene && !wide) {
        feat |= PBRS_FEAT_LDS_SCENE;
        lds += c->lds_scene_bytes;
    } else if (c->lds_top && !wide && !(feat & (PBRS_FEAT_FLAT_TLAS | PBRS_FEAT_FULL_STEPS))) {
        feat |= PBRS_FEAT_LDS_TOP;
        lds += c->lds_top_bytes;
    }
    const shadow_fn_t fn = shadow_fns()[feat];
    if (!fn) return fail(c, PBRS_E_DEVICE, "no k_shadow instantiation for this scene's feature set");
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kBlock), lds, c->stream, c->S, c->st, count, heads, c->gcnt + 1, indirect, slow_list, slow_count);
    if (!indirect) c->pending.kernel_features_shadow = feat;
    return PBRS_OK;
}

// The split probe of an earlier pass (run_pass), if its counts have arrived: keep the queue split where it keeps at most
// PBRS_SPLIT_KEEP_PERCENT of a pass's rays for k_shade.  Never waits.
void poll_split_probe(pbrs_ctx* c) {
    if (!c->split_probe_in_flight || hipEventQuery(c->split_ev) != hipSuccess) {
        (void)hipGetLastError();  // hipErrorNotReady is not an error of this call
        return;
    }
    c->split_probe_in_flight = false;
    c->split_decision = (c->split_host[0] * 100ull <= c->split_host[1] * PBRS_SPLIT_KEEP_PERCENT) ? 1 : 2;
}

// One pass: kc sample indices starting at `first` for every pixel of the tile.
// `handoff`: the pass moves to the late stream at bounce pbrs_ctx::overlap_from (at the latest for its k_accumulate: the late stream runs
// the passes' accumulations in pass order, src/main.rs:205) and leaves the main stream to the next pass, which works in the other pass set.
int run_pass(pbrs_ctx* c, RenderConst rc, uint32_t first, uint32_t kc, bool stats, Timer& tm, bool handoff = false) {
    pbrs_ctx::PassSet& set = c->pass_set[c->cur_set];
    // the set's memory is free once the pass that used it last has accumulated (two passes back, on the late stream)
    if (handoff && set.in_flight) HIPCHK(c, hipStreamWaitEvent(c->stream, set.accumulated, 0));
    auto to_late_stream = [&]() -> hipError_t {
        hipError_t e = hipEventRecord(set.late, c->stream);
        c->stream = c->second_stream;
        return e != hipSuccess ? e : hipStreamWaitEvent(c->stream, set.late, 0);
    };
    const uint32_t P = rc.n_pixels;
    const uint32_t N = P * kc;
    rc.pass_first_sample = first;
    rc.n_slots = N;
    const uint32_t grid = (N + kBlock - 1) / kBlock;
    const uint32_t sgrid = grid < kStreamGridCap ? grid : kStreamGridCap;
    // persistent traversal kernels: enough blocks to fill the chip, each pulls work until the queue is empty
    const uint32_t pgrid = grid < kPersistentBlocks ? grid : kPersistentBlocks;
    const uint32_t stride = kMaxDepth + 2;
    uint32_t* act = c->counters;                 // act[b]: paths entering bounce b (b >= 1)
    // ns[b]: one 64-bit word per bounce: low half = paths whose light estimate waits for visibility, high half = shadow rays
    unsigned long long* ns = reinterpret_cast<unsigned long long*>(c->counters + stride);
    // work-fetch heads of k_extend / k_shadow: kHeadWords words per bounce (one head per queue segment, kernels.h)
    uint32_t* slowx = c->counters + 3 * stride;  // slow-list lengths per bounce: k_extend's, k_shadow's
    uint32_t* slows = slowx + stride;
    uint32_t* xhead = c->counters + 7 * stride;
    uint32_t* shead = xhead + stride * kHeadWords;
    uint32_t* xhead2 = shead + stride * kHeadWords;  // the binary-walk launches over the slow lists
    uint32_t* shead2 = xhead2 + stride * kHeadWords;
    const bool wide_x = c->wide_extend && !stats, wide_s = c->wide_shadow && !stats;
    const size_t lds_wx = lds_bytes_wide(c, true), lds_ws = lds_bytes_wide(c, false);
    uint32_t* neeq = c->neeq;
    HIPCHK(c, hipMemsetAsync(c->counters, 0, kCounterWords * sizeof(uint32_t), c->stream));
    if (tm.begin(0)) return fail(c, PBRS_E_DEVICE, "event record failed");
    hipLaunchKernelGGL(k_raygen, dim3(grid), dim3(kBlock), 0, c->stream, c->st, rc);
    tm.end();
    const size_t lds = lds_bytes(c);
    poll_split_probe(c);
    // the direct-lighting integrator is at most two rays deep whatever `depth` says (directlighting.rs:15-17, :36, :49)
    const uint32_t n_bounces = rc.integrator == PBRS_INTEGRATOR_DIRECT      ? (rc.max_depth ? 2u : 0u)
                               : rc.integrator >= PBRS_INTEGRATOR_MATERIALS ? 1u  // the visualisers: one cast, no lights
                                                                            : rc.max_depth;
    // this pass counts what k_extend's queue split keeps (the first path-integrator pass of an uploaded one-class scene)
    const bool probe_split = c->split_decision == 0 && !c->split_probe_in_flight && rc.integrator == PBRS_INTEGRATOR_PATH && c->S.n_classes <= 1u && c->split_queue && n_bounces > 0;
    for (uint32_t b = 0; b < n_bounces; ++b) {
        if (handoff && b == c->overlap_from) HIPCHK(c, to_late_stream());
        // bounce b reads the path records of set b & 1 (k_raygen wrote set 0) and k_shade writes set (b + 1) & 1; the
        // queue length of bounce 0 is the pass size, later ones are counted on the device
        const uint32_t* cnt_in = b == 0 ? nullptr : act + b;
        if (tm.begin(1)) return fail(c, PBRS_E_DEVICE, "event record failed");
        // the path integrator on a scene with one shading class (no class sort): k_extend splits its queue into the hits k_shade
        // shades, the paths that only end (emitter hits, misses that see the environment) and the misses nothing happens to
        const uint32_t qsplit = (rc.integrator == PBRS_INTEGRATOR_PATH && c->S.n_classes <= 1u && c->split_queue && c->split_decision != 2) ? (1u | (b == 0 ? 2u : 0u)) : 0u;
        int lrc = launch_extend(c, stats, wide_x, pgrid, wide_x ? lds_wx : lds, b & 1u, cnt_in, N, xhead + b * kHeadWords, nullptr, c->slow, slowx + b, qsplit);
        if (!lrc && wide_x)  // what the wide walks refused (rays outside the guarded range of the division-free box test, overlong stacks)
            lrc = launch_extend(c, false, false, pgrid < kSlowGrid ? pgrid : kSlowGrid, lds, b & 1u, slowx + b, 0u, xhead2 + b * kHeadWords, c->slow, nullptr, nullptr, qsplit);
        if (lrc) return lrc;
        tm.end();
        if (tm.begin(2)) return fail(c, PBRS_E_DEVICE, "event record failed");
        // several shading classes (and an integrator that shades): order the queue by class first; counted as shade time
        const uint32_t sorted = (c->S.n_classes > 1u && rc.integrator <= PBRS_INTEGRATOR_DIRECT && c->sort_classes) ? 1u : 0u;
        // ... and where one of the classes is Lambertian (and the integrator has a Lambert variant), class-major over the whole
        // queue, so that the class gets a launch of that variant and the other classes one of the general kernel
        const bool split = sorted && c->lambert_class && c->split_lambert && !c->textured && !c->fourier && rc.integrator == PBRS_INTEGRATOR_PATH;
        // ... or a Fourier BSDF: its lobe's code (168 registers and scratch in k_shade's variants that carry it) then runs over
        // the vertices on such a material only, the other classes take the kernels without it
        const bool fsplit = sorted && c->fourier && c->fourier_class && c->split_fourier && rc.integrator <= PBRS_INTEGRATOR_DIRECT;
        const uint32_t n_tiles = (N + PBRS_SORT_TILE - 1) / PBRS_SORT_TILE;
        if (split || qsplit || fsplit) {  // class-major over the whole queue; a queue k_extend split: class 1 = the kept paths, last
            if (qsplit) hipLaunchKernelGGL(k_class_count<2u>, dim3(n_tiles), dim3(kBlock), 0, c->stream, c->st, cnt_in, N);
            else hipLaunchKernelGGL(k_class_count<PBRS_MAX_CLASSES>, dim3(n_tiles), dim3(kBlock), 0, c->stream, c->st, cnt_in, N);
            hipLaunchKernelGGL(k_class_scan, dim3(1), dim3(64 * PBRS_MAX_CLASSES), 0, c->stream, c->st, cnt_in, N, qsplit ? 1u : fsplit ? c->fourier_class : c->lambert_class,
                               (qsplit && probe_split) ? c->bounce_acc + 2 * PBRS_STATS_MAX_BOUNCES : nullptr);
            if (qsplit) hipLaunchKernelGGL(k_class_scatter<2u>, dim3(n_tiles), dim3(kBlock), 0, c->stream, c->st, cnt_in, N);
            else hipLaunchKernelGGL(k_class_scatter<PBRS_MAX_CLASSES>, dim3(n_tiles), dim3(kBlock), 0, c->stream, c->st, cnt_in, N);
        } else if (sorted) {
            hipLaunchKernelGGL(k_class_sort, dim3(n_tiles), dim3(kBlock), 0, c->stream, c->st, cnt_in, N);
        }
        {
#define PBRS_LAUNCH_SHADE_LDS(I, T, SP, LDS)                                                                                              \
    hipLaunchKernelGGL((k_shade<I, T, SP>), dim3(shade_grid), dim3(kBlock), LDS, c->stream, c->S, c->st, rc, b, shade_count, N, act + b + 1, neeq, ns + b, \
                       sorted | split_sorted, shade_range)
#define PBRS_LAUNCH_SHADE(I, T, SP) PBRS_LAUNCH_SHADE_LDS(I, T, SP, 0)
// the path integrator's untextured variants: with the scene's shading records (and triangle records) staged in LDS where they fit
#define PBRS_LAUNCH_SHADE_PATH(SP)                                                                                                               \
    do {                                                                                                                                          \
        if (c->shade_lds == (PBRS_SHADE_LDS_RECORDS | PBRS_SHADE_LDS_TRIS))                                                                        \
            PBRS_LAUNCH_SHADE_LDS(PBRS_INTEGRATOR_PATH, false, (SP) | PBRS_SHADE_LDS_RECORDS | PBRS_SHADE_LDS_TRIS, c->shade_lds_bytes);           \
        else if (c->shade_lds == PBRS_SHADE_LDS_RECORDS)                                                                                           \
            PBRS_LAUNCH_SHADE_LDS(PBRS_INTEGRATOR_PATH, false, (SP) | PBRS_SHADE_LDS_RECORDS, c->shade_lds_bytes);                                 \
        else                                                                                                                                       \
            PBRS_LAUNCH_SHADE(PBRS_INTEGRATOR_PATH, false, SP);                                                                                    \
    } while (0)
            const uint32_t shade_grid = grid;
            const uint2* shade_range = qsplit ? c->st.class_range + 1 : nullptr;  // a split queue: the kept paths
            const uint32_t split_sorted = qsplit ? 1u : 0u;
            const uint32_t* shade_count = cnt_in;
            {
            const bool direct = rc.integrator == PBRS_INTEGRATOR_DIRECT;
            if (rc.integrator == PBRS_INTEGRATOR_MATERIALS) {
                PBRS_LAUNCH_SHADE(PBRS_INTEGRATOR_MATERIALS, false, 0u);
            } else if (rc.integrator == PBRS_INTEGRATOR_NORMALS) {
                PBRS_LAUNCH_SHADE(PBRS_INTEGRATOR_NORMALS, false, 0u);
            } else if (c->fourier && fsplit) {  // the Fourier materials' class under the kernels that carry the lobe, the rest without
                shade_range = c->st.class_range + c->fourier_class;  // (one untextured Fourier lobe per material: the variant cut down to it)
                if (direct) PBRS_LAUNCH_SHADE(PBRS_INTEGRATOR_DIRECT, false, PBRS_SHADE_FOURIER | PBRS_SHADE_FOURIER_ONLY);
                else PBRS_LAUNCH_SHADE(PBRS_INTEGRATOR_PATH, false, PBRS_SHADE_FOURIER | PBRS_SHADE_FOURIER_ONLY);
                shade_range = c->st.class_range + PBRS_MAX_CLASSES;
                if (c->textured) {
                    if (direct) PBRS_LAUNCH_SHADE(PBRS_INTEGRATOR_DIRECT, true, 0u);
                    else PBRS_LAUNCH_SHADE(PBRS_INTEGRATOR_PATH, true, 0u);
                } else {
                    if (direct) PBRS_LAUNCH_SHADE(PBRS_INTEGRATOR_DIRECT, false, 0u);
                    else PBRS_LAUNCH_SHADE(PBRS_INTEGRATOR_PATH, false, 0u);
                }
            } else if (c->fourier && c->fourier_class && c->S.n_classes == 1u) {  // every material with lobes is a Fourier BSDF
                if (direct) PBRS_LAUNCH_SHADE(PBRS_INTEGRATOR_DIRECT, false, PBRS_SHADE_FOURIER | PBRS_SHADE_FOURIER_ONLY);
                else PBRS_LAUNCH_SHADE(PBRS_INTEGRATOR_PATH, false, PBRS_SHADE_FOURIER | PBRS_SHADE_FOURIER_ONLY);
            } else if (c->fourier) {  // some material is a Fourier BSDF: the kernels that carry the lobe (and textures)
                if (direct) PBRS_LAUNCH_SHADE(PBRS_INTEGRATOR_DIRECT, true, PBRS_SHADE_FOURIER);
                else PBRS_LAUNCH_SHADE(PBRS_INTEGRATOR_PATH, true, PBRS_SHADE_FOURIER);
            } else if (c->textured) {  // some material evaluates a non-Solid texture per hit
                if (direct) PBRS_LAUNCH_SHADE(PBRS_INTEGRATOR_DIRECT, true, 0u);
                else PBRS_LAUNCH_SHADE(PBRS_INTEGRATOR_PATH, true, 0u);
            } else if (direct) {
                PBRS_LAUNCH_SHADE(PBRS_INTEGRATOR_DIRECT, false, 0u);
            } else if (split) {
                shade_range = c->st.class_range + c->lambert_class;
                switch (c->light_spec) {
                    case PBRS_SHADE_LIGHT_SPHERE: PBRS_LAUNCH_SHADE_PATH(PBRS_SHADE_LAMBERT | PBRS_SHADE_LIGHT_SPHERE); break;
                    case PBRS_SHADE_LIGHT_TRIANGLE: PBRS_LAUNCH_SHADE_PATH(PBRS_SHADE_LAMBERT | PBRS_SHADE_LIGHT_TRIANGLE); break;
                    default: PBRS_LAUNCH_SHADE_PATH(PBRS_SHADE_LAMBERT); break;
                }
                shade_range = c->st.class_range + PBRS_MAX_CLASSES;
                PBRS_LAUNCH_SHADE_PATH(0u);
            } else {  // the path integrator, specialised on what the scene's materials and lights are (c->shade_spec)
                switch (c->shade_spec) {
                    case PBRS_SHADE_LAMBERT: PBRS_LAUNCH_SHADE_PATH(PBRS_SHADE_LAMBERT); break;
                    case PBRS_SHADE_LAMBERT | PBRS_SHADE_LIGHT_SPHERE: PBRS_LAUNCH_SHADE_PATH(PBRS_SHADE_LAMBERT | PBRS_SHADE_LIGHT_SPHERE); break;
                    case PBRS_SHADE_LAMBERT | PBRS_SHADE_LIGHT_TRIANGLE: PBRS_LAUNCH_SHADE_PATH(PBRS_SHADE_LAMBERT | PBRS_SHADE_LIGHT_TRIANGLE); break;
                    default: PBRS_LAUNCH_SHADE_PATH(0u); break;
                }
            }
            }
#undef PBRS_LAUNCH_SHADE_PATH
#undef PBRS_LAUNCH_SHADE
#undef PBRS_LAUNCH_SHADE_LDS
        }
        tm.end();
        if (tm.begin(3)) return fail(c, PBRS_E_DEVICE, "event record failed");
        lrc = launch_shadow(c, stats, wide_s, pgrid, wide_s ? lds_ws : lds, reinterpret_cast<const uint32_t*>(ns + b), shead + b * kHeadWords, nullptr, c->slow, slows + b);
        if (!lrc && wide_s)
            lrc = launch_shadow(c, false, false, pgrid < kSlowGrid ? pgrid : kSlowGrid, lds, slows + b, shead2 + b * kHeadWords, c->slow, nullptr, nullptr);
        if (lrc) return lrc;
        hipLaunchKernelGGL(k_nee_resolve, dim3(sgrid), dim3(kBlock), 0, c->stream, c->st, neeq, reinterpret_cast<const uint32_t*>(ns + b));
        tm.end();
    }
    if (probe_split) {
        // the first pass of this scene through the path integrator: how much of its queues did the split keep for k_shade?  The
        // counts are copied out behind an event; a later pass that finds the event complete takes the decision (poll_split_probe).
        // No synchronisation; the image does not depend on the answer.
        unsigned long long* acc = c->bounce_acc + 2 * PBRS_STATS_MAX_BOUNCES;
        HIPCHK(c, hipMemcpyAsync(c->split_host, acc, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipEventRecord(c->split_ev, c->stream));
        HIPCHK(c, hipMemsetAsync(acc, 0, 2 * sizeof(unsigned long long), c->stream));
        c->split_probe_in_flight = true;
    }
    if (stats)  // queue sizes of this pass, bounce by bounce (the counters are cleared at the start of every pass)
        hipLaunchKernelGGL(k_sum_bounce_counts, dim3(1), dim3(64), 0, c->stream, act, ns, N, n_bounces, c->bounce_acc);
    if (handoff && c->stream != c->second_stream) HIPCHK(c, to_late_stream());
    if (tm.begin(4)) return fail(c, PBRS_E_DEVICE, "event record failed");
    hipLaunchKernelGGL(k_accumulate, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, c->st, c->sum, P, kc, rc.chunk_pixels, rc.w, rc.tiles8_per_row, c->nonfinite);
    tm.end();
    if (handoff) {
        HIPCHK(c, hipEventRecord(set.accumulated, c->stream));
        set.in_flight = true;
    }
    HIPCHK(c, hipGetLastError());
    return PBRS_OK;
}

int render_common(pbrs_ctx* c, const pbrs_camera* cam, const pbrs_render_params* p, float* rgb_device) {
    HIPCHK(c, hipSetDevice(c->device));  // before check_params: the automatic pass size reads THIS device's free memory
    int rcode = check_params(c, cam, p);
    if (rcode) return rcode;
    const uint32_t P = p->w * p->h;
    const uint32_t spp = p->strata_x * p->strata_y;
    const uint32_t K = auto_samples_per_pass(c, p);
    rcode = ensure_work(c, (size_t)P * K, P);
    if (rcode) return rcode;
    RenderConst rc = make_const(cam, p);
    const bool stats = p->collect_counters != 0;
    Timer tm{c, p->time_stages != 0};
    c->events_used = 0;
    c->pending = pbrs_stats{};
    c->pending_counters = stats;
    c->pending_times = p->time_stages != 0;
    c->pending.samples = (uint64_t)P * spp;
    if (stats) HIPCHK(c, hipMemsetAsync(c->gcnt, 0, 2 * sizeof(GlobalCounters), c->stream));
    if (stats) HIPCHK(c, hipMemsetAsync(c->bounce_acc, 0, 2 * PBRS_STATS_MAX_BOUNCES * sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipMemsetAsync(c->nonfinite, 0, sizeof(unsigned long long), c->stream));
    if (c->pending_times) HIPCHK(c, hipEventRecord(c->total_ev[0], c->stream));
    HIPCHK(c, hipMemsetAsync(c->sum, 0, 3 * (size_t)P * sizeof(float), c->stream));
    uint32_t passes = 0;
    // Where the render has more than one pass, passes alternate between the two pass sets and hand their late bounces to the second
    // stream: those are near-empty launches that end with the latency of their longest walks (C4: 47 ms per frame in kernels that leave
    // most of the chip idle, profiles/r04k_trace_gaps_c4.log) — the next pass's first bounces, queued behind the hand-over on the main
    // stream, fill it.  The instrumented render keeps one stream (its counters are per pass).
    bool two = c->overlap_passes && !stats && spp > K;
    if (two) {
        use_pass_set(c, 1);
        rcode = ensure_work(c, (size_t)P * K, P);
        use_pass_set(c, 0);
        if (rcode) {  // no memory for the second set (a device shared with other processes): every pass on the main stream, as before
            two = false;
            c->error.clear();
        }
    }
    for (uint32_t first = 0; first < spp; first += K) {
        uint32_t kc = spp - first < K ? spp - first : K;
        if (two) use_pass_set(c, (int)(passes & 1u));  // (also: back to the main stream)
        rcode = run_pass(c, rc, first, kc, stats, tm, two);
        if (rcode) {
            use_pass_set(c, 0);
            return rcode;
        }
        ++passes;
    }
    if (two) {
        const hipEvent_t last = c->pass_set[c->cur_set].accumulated;
        use_pass_set(c, 0);
        HIPCHK(c, hipStreamWaitEvent(c->main_stream, last, 0));  // the late stream has run every pass's accumulation, in order
    }
    hipLaunchKernelGGL(k_finalize, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, c->sum, rgb_device, P, 1.0f / (float)spp);
    if (c->pending_times) HIPCHK(c, hipEventRecord(c->total_ev[1], c->stream));
    HIPCHK(c, hipGetLastError());
    c->pending.passes = passes;
    c->pending.launches_extend = c->pending.launches_shade = c->pending.launches_shadow =
        passes * (p->integrator == PBRS_INTEGRATOR_DIRECT ? (p->max_depth ? 2u : 0u) : p->integrator >= PBRS_INTEGRATOR_MATERIALS ? 1u : p->max_depth);
    return PBRS_OK;
}

int collect(pbrs_ctx* c, pbrs_stats* out) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    pbrs_stats s = c->pending;
    {
        unsigned long long bad = 0;
        HIPCHK(c, hipMemcpy(&bad, c->nonfinite, sizeof bad, hipMemcpyDeviceToHost));
        s.invalid_samples = bad;
    }
    if (c->pending_counters) {
        GlobalCounters g[2];
        HIPCHK(c, hipMemcpy(g, c->gcnt, sizeof g, hipMemcpyDeviceToHost));
        s.closest_rays = g[0].rays;
        s.shade_events = g[0].hits;
        s.tlas_nodes = g[0].tlas_nodes; s.blas_nodes = g[0].blas_nodes; s.instances = g[0].instances;
        s.instance_hits = g[0].instance_hits; s.triangles = g[0].triangles; s.tri_shading = g[0].tri_shading;
        s.spheres = g[0].spheres; s.quads = g[0].quads; s.cuboids = g[0].cuboids; s.disks = g[0].disks;
        s.shadow_rays = g[1].rays;
        s.shadow_tlas_nodes = g[1].tlas_nodes; s.shadow_blas_nodes = g[1].blas_nodes; s.shadow_instances = g[1].instances;
        s.shadow_triangles = g[1].triangles;
        s.shadow_prims = g[1].spheres + g[1].quads + g[1].cuboids + g[1].disks;
        unsigned long long per_bounce[2 * PBRS_STATS_MAX_BOUNCES];
        HIPCHK(c, hipMemcpy(per_bounce, c->bounce_acc, sizeof per_bounce, hipMemcpyDeviceToHost));
        for (uint32_t b = 0; b < PBRS_STATS_MAX_BOUNCES; ++b) {
            s.paths_at_bounce[b] = per_bounce[b];
            s.shadow_rays_at_bounce[b] = per_bounce[PBRS_STATS_MAX_BOUNCES + b];
        }
    }
    if (c->pending_times) {
        float* acc[5] = {&s.ms_raygen, &s.ms_extend, &s.ms_shade, &s.ms_shadow, &s.ms_accumulate};
        for (size_t i = 0; i < c->events_used; ++i) {
            float ms = 0.0f;
            HIPCHK(c, hipEventElapsedTime(&ms, c->events[i].a, c->events[i].b));
            *acc[c->events[i].stage] += ms;
        }
        HIPCHK(c, hipEventElapsedTime(&s.ms_total, c->total_ev[0], c->total_ev[1]));
    }
    if (out) *out = s;
    return PBRS_OK;
}

// The traversal kernels take their per-lane stacks from dynamic LDS.  The limit a kernel may ask for is per-function state
// of the PROCESS (hipFuncSetAttribute), not of a context: it is raised once per device, to the most any scene may need
// (pbrs_upload_scene refuses stacks above kLdsBytesPerCU / 2), so that contexts holding scenes with different stack depths
// can render side by side — rewriting it per upload let the last upload decide for every context of the process.
std::mutex g_kernel_cfg_mutex;
bool g_kernel_cfg_done[64] = {};
int configure_kernels(pbrs_ctx* c) {
    std::lock_guard<std::mutex> lock(g_kernel_cfg_mutex);
    if (c->device < 64 && g_kernel_cfg_done[c->device]) return PBRS_OK;
    const int cap = (int)(kLdsBytesPerCU / 2);
    std::vector<const void*> traversal_kernels = {reinterpret_cast<const void*>(&k_extend<true, PBRS_FEAT_ALL>),
                                                  reinterpret_cast<const void*>(&k_shadow<true, PBRS_FEAT_ANALYTIC | PBRS_FEAT_FLAT_TLAS>),
                                                  reinterpret_cast<const void*>(&k_intersect_rays<false, false>),
                                                  reinterpret_cast<const void*>(&k_intersect_rays<false, true>),
#ifdef PBRS_DEV_OVERRIDES
                                                  reinterpret_cast<const void*>(&k_intersect_rays<true, false>),
                                                  reinterpret_cast<const void*>(&k_intersect_rays<true, true>),
#endif
    };
    for (uint32_t f = 0; f < kFeatCombos; ++f) {
        if (extend_fns()[f]) traversal_kernels.push_back(reinterpret_cast<const void*>(extend_fns()[f]));
        if (shadow_fns()[f]) traversal_kernels.push_back(reinterpret_cast<const void*>(shadow_fns()[f]));
    }
    for (const void* k : traversal_kernels) HIPCHK(c, hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    if (c->device < 64) g_kernel_cfg_done[c->device] = true;
    return PBRS_OK;
}

}  // namespace

extern "C" {

#ifdef PBRS_PROBE_SHADE
// developer probe: reads and clears the k_shade region cycle sums
int pbrs_debug_shade_probe(unsigned long long* out16) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_shade_probe), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long zero[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_shade_probe), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
#endif

#ifdef PBRS_PROBE_TIME
// developer probe: reads and clears the traversal loops' cycle sums by region ([0]: k_extend, [1]: k_shadow; kernels.h)
int pbrs_debug_trav_time(unsigned long long* out16) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_trav_time), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long zero[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_trav_time), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
#endif
#ifdef PBRS_PROBE_TRAV
// developer probe: reads and clears the traversal loops' event counts ([0]: k_extend, [1]: k_shadow; kernels.h)
int pbrs_debug_trav_probe(unsigned long long* out48) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out48, HIP_SYMBOL(g_trav_probe), 48 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long zero[48] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_trav_probe), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
#endif

int pbrs_create(int device_ordinal, pbrs_ctx** out) {
    if (!out) return PBRS_E_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_ordinal < 0 || device_ordinal >= n) return PBRS_E_DEVICE;
    pbrs_ctx* c = new pbrs_ctx();
    c->device = device_ordinal;
    // every failure below leaves through pbrs_destroy, which releases whatever had been created by then
    // a non-blocking stream: work another library queues on the legacy default stream (torch's copies in bench.py) neither
    // waits for the frames queued here nor holds them up
    bool ok = hipSetDevice(device_ordinal) == hipSuccess && hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) == hipSuccess;
    c->stream = c->main_stream = c->own_stream;
    {
        int least = 0, greatest = 0;  // (numerically lower = higher priority)
        ok = ok && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess &&
             hipStreamCreateWithPriority(&c->second_stream, hipStreamNonBlocking, greatest) == hipSuccess;
    }
    ok = ok && hipEventCreateWithFlags(&c->pass_set[0].accumulated, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c->pass_set[1].accumulated, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c->pass_set[0].late, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c->pass_set[1].late, hipEventDisableTiming) == hipSuccess;
    if (const char* e = dev_env("PBRS_OVERLAP_PASSES")) c->overlap_passes = std::atoi(e) != 0;
    if (const char* e = dev_env("PBRS_SORT_CLASSES")) c->sort_classes = std::atoi(e) != 0;
    if (const char* e = dev_env("PBRS_SPLIT_LAMBERT")) c->split_lambert = std::atoi(e) != 0;
    if (const char* e = dev_env("PBRS_SPLIT_FOURIER")) c->split_fourier = std::atoi(e) != 0;
    if (const char* e = dev_env("PBRS_SPLIT_QUEUE")) c->split_queue = std::atoi(e) != 0;
    for (int k = 0; ok && k < 2; ++k) {
        hipEvent_t ev = nullptr;
        ok = hipEventCreate(&ev) == hipSuccess;
        if (ok) c->total_ev.push_back(ev);
    }
    ok = ok && hipMalloc(reinterpret_cast<void**>(&c->pass_set[1].counters), kCounterWords * sizeof(uint32_t)) == hipSuccess &&
         hipMalloc(reinterpret_cast<void**>(&c->counters), kCounterWords * sizeof(uint32_t)) == hipSuccess &&
         hipMalloc(reinterpret_cast<void**>(&c->gcnt), 2 * sizeof(GlobalCounters)) == hipSuccess &&
         hipMalloc(reinterpret_cast<void**>(&c->nonfinite), sizeof(unsigned long long)) == hipSuccess &&
         hipMalloc(reinterpret_cast<void**>(&c->bounce_acc), (2 * PBRS_STATS_MAX_BOUNCES + 2) * sizeof(unsigned long long)) == hipSuccess &&
         hipMemset(c->bounce_acc, 0, (2 * PBRS_STATS_MAX_BOUNCES + 2) * sizeof(unsigned long long)) == hipSuccess &&
         hipMemset(c->nonfinite, 0, sizeof(unsigned long long)) == hipSuccess &&
         hipHostMalloc(reinterpret_cast<void**>(&c->split_host), 2 * sizeof(unsigned long long), hipHostMallocDefault) == hipSuccess &&
         hipEventCreateWithFlags(&c->split_ev, hipEventDisableTiming) == hipSuccess;
    ok = ok && configure_kernels(c) == PBRS_OK;
    if (!ok) {
        pbrs_destroy(c);
        return PBRS_E_DEVICE;
    }
    *out = c;
    return PBRS_OK;
}

void pbrs_destroy(pbrs_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->second_stream) (void)hipStreamSynchronize(c->second_stream);
    free_scene(c);
    free_work(c);  // (leaves pass set 0 in use)
    if (c->counters) (void)hipFree(c->counters);
    if (c->pass_set[1].counters) (void)hipFree(c->pass_set[1].counters);
    for (int k = 0; k < 2; ++k)
        if (c->pass_set[k].accumulated) (void)hipEventDestroy(c->pass_set[k].accumulated);
    for (int k = 0; k < 2; ++k)
        if (c->pass_set[k].late) (void)hipEventDestroy(c->pass_set[k].late);
    if (c->second_stream) (void)hipStreamDestroy(c->second_stream);
    if (c->gcnt) (void)hipFree(c->gcnt);
    if (c->nonfinite) (void)hipFree(c->nonfinite);
    if (c->bounce_acc) (void)hipFree(c->bounce_acc);
    if (c->split_host) (void)hipHostFree(c->split_host);
    if (c->split_ev) (void)hipEventDestroy(c->split_ev);
    for (auto& e : c->events) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    for (auto& e : c->total_ev) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

const char* pbrs_last_error(const pbrs_ctx* c) { return c ? c->error.c_str() : "null context"; }

int pbrs_set_stream(pbrs_ctx* c, void* hip_stream) {
    if (!c) return PBRS_E_INVALID;
    c->main_stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    c->stream = c->main_stream;  // (pass set 0 is the one in use between calls)
    return PBRS_OK;
}

int pbrs_set_pass_overlap(pbrs_ctx* c, int enabled) {
    if (!c) return PBRS_E_INVALID;
    c->overlap_passes = enabled != 0;
    return PBRS_OK;
}

int pbrs_upload_scene(pbrs_ctx* c, const pbrs_scene_desc* d) {
    if (!c || !d) return PBRS_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (d->n_tlas_nodes == 0 || d->n_instances == 0) return fail(c, PBRS_E_INVALID, "scene without instances");
    // Host-side shape checks: every index the kernels dereference must be in range before any launch.
    for (uint32_t i = 0; i < d->n_tlas_nodes; ++i) {
        const pbrs_node& n = d->tlas_nodes[i];
        if (n.b & PBRS_LEAF_FLAG) {
            if (n.a >= d->n_instances) return fail(c, PBRS_E_INVALID, "tlas leaf references a missing instance");
        } else if (n.a >= d->n_tlas_nodes || i + 1 >= d->n_tlas_nodes) {
            return fail(c, PBRS_E_INVALID, "tlas child out of range");
        }
    }
    for (uint32_t i = 0; i < d->n_blas_nodes; ++i) {
        const pbrs_node& n = d->blas_nodes[i];
        if (n.b & PBRS_LEAF_FLAG) {
            uint32_t cnt = n.b & ~PBRS_LEAF_FLAG;
            if ((uint64_t)n.a + cnt > d->n_triangles) return fail(c, PBRS_E_INVALID, "blas leaf range out of range");
        } else if (n.a >= d->n_blas_nodes || i + 1 >= d->n_blas_nodes || (n.b & 3u) > 2u) {
            return fail(c, PBRS_E_INVALID, "blas child out of range");
        }
    }
    uint32_t max_blas_height = 0;
    for (uint32_t i = 0; i < d->n_meshes; ++i) {
        if (d->meshes[i].root >= d->n_blas_nodes) return fail(c, PBRS_E_INVALID, "mesh root out of range");
        if (d->meshes[i].height > max_blas_height) max_blas_height = d->meshes[i].height;
    }
    for (uint32_t i = 0; i < d->n_instances; ++i) {
        const pbrs_instance& in = d->instances[i];
        if (in.material >= d->n_materials) return fail(c, PBRS_E_INVALID, "instance material out of range");
        if (in.shape_kind > PBRS_SHAPE_MESH) return fail(c, PBRS_E_INVALID, "unknown shape kind");
        if (in.shape_kind == PBRS_SHAPE_MESH ? in.shape_index >= d->n_meshes : in.shape_index >= d->n_shapes)
            return fail(c, PBRS_E_INVALID, "instance shape out of range");
        if (in.shape_kind == PBRS_SHAPE_MESH ? in.blas_root >= d->n_blas_nodes : (in.shape_kind == PBRS_SHAPE_TRIANGLE && in.blas_root >= d->n_triangles))
            return fail(c, PBRS_E_INVALID, "instance blas_root out of range");
    }
    bool vis_records = d->n_materials > 0;
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const pbrs_material& m = d->materials[i];
        if (m.n_bxdfs > PBRS_MAX_BXDFS || (uint64_t)m.first_bxdf + m.n_bxdfs > d->n_bxdfs) return fail(c, PBRS_E_INVALID, "material lobes out of range");
        if (m.vis_bxdf > d->n_bxdfs) return fail(c, PBRS_E_INVALID, "material visualiser record out of range");
        if (m.vis_bxdf == 0) vis_records = false;
    }
    bool textured = false, fourier = false;
    for (uint32_t i = 0; i < d->n_bxdfs; ++i) {
        const uint32_t t = d->bxdfs[i].tex & ~PBRS_BXDF_TEX_DROP_IF_BLACK;
        if (t > d->n_textures) return fail(c, PBRS_E_INVALID, "lobe texture out of range");
        textured = textured || t != 0;
        if (d->bxdfs[i].kind > PBRS_BXDF_FOURIER) return fail(c, PBRS_E_INVALID, "unknown lobe kind");
        if (d->bxdfs[i].kind == PBRS_BXDF_FOURIER) {
            if (d->bxdfs[i].intrusion >= d->n_fourier_tables) return fail(c, PBRS_E_INVALID, "Fourier lobe table out of range");
            fourier = true;
        }
    }
    // Fourier tables (geometry/src/fourier.rs:99-151): every array inside the pools, every series inside the coefficients;
    // with finite, strictly ascending nodes the interpolation weights are finite for every direction the lobe accepts (a NaN
    // direction is refused there), and the lobe skips neighbours outside the table: no lane indexes outside the pools
    for (uint32_t i = 0; i < d->n_fourier_tables; ++i) {
        const pbrs_fourier_table& t = d->fourier_tables[i];
        const uint64_t n = t.n_mu, nn = n * n, nf = d->n_tex_floats, nw = d->n_tex_words;
        if (n < 3 || (t.n_channels != 1 && t.n_channels != 3)) return fail(c, PBRS_E_INVALID, "Fourier table: sizes");
        if (t.mu + n > nf || t.cdf + nn > nf || t.a0 + nn > nf || (uint64_t)t.a + t.n_coeffs > nf || (uint64_t)t.recip + t.m_max > nf ||
            t.a_offset + nn > nw || t.m_lookup + nn > nw)
            return fail(c, PBRS_E_INVALID, "Fourier table: arrays out of range");
        for (uint64_t k = 0; k < n; ++k) {  // finite, strictly ascending nodes: no interval of zero width, no NaN weight (device/fourier.h)
            const float m0 = d->tex_floats[t.mu + k];
            if (!pn_isfinite(m0) || (k + 1 < n && !(m0 < d->tex_floats[t.mu + k + 1]))) return fail(c, PBRS_E_INVALID, "Fourier table: mu is not finite and strictly ascending");
        }
        for (uint64_t k = 0; k < nn; ++k) {
            const uint64_t off = d->tex_words[t.a_offset + k], len = d->tex_words[t.m_lookup + k];
            if (len > t.m_max || off + len * t.n_channels > t.n_coeffs) return fail(c, PBRS_E_INVALID, "Fourier table: series out of range");
        }
    }
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        const pbrs_texture& t = d->textures[i];
        if (t.kind == PBRS_TEX_PERLIN) {
            if ((uint64_t)t.data + 768 > d->n_tex_floats || (uint64_t)t.perm + 768 > d->n_tex_words) return fail(c, PBRS_E_INVALID, "perlin tables out of range");
            for (uint32_t k = 0; k < 768; ++k)
                if (d->tex_words[t.perm + k] > 255u) return fail(c, PBRS_E_INVALID, "perlin permutation entry above 255");
        } else if (t.kind == PBRS_TEX_IMAGE) {
            if (t.width == 0 || t.height == 0 || (uint64_t)t.data + 3ull * t.width * t.height > d->n_tex_floats)
                return fail(c, PBRS_E_INVALID, "image texels out of range");
        } else if (t.kind != PBRS_TEX_CHECKER) {
            return fail(c, PBRS_E_INVALID, "unknown texture kind");
        }
    }
    if (d->env_kind > PBRS_ENV_DUSK) return fail(c, PBRS_E_INVALID, "unknown environment kind");
    if (d->env_kind == PBRS_ENV_IMAGE && (d->env_texture >= d->n_textures || d->textures[d->env_texture].kind != PBRS_TEX_IMAGE))
        return fail(c, PBRS_E_INVALID, "environment map is not an image texture");
    for (uint32_t i = 0; i < d->n_area_lights; ++i) {
        uint32_t k = d->area_lights[i].shape_kind;
        if (!(k == PBRS_SHAPE_SPHERE || k == PBRS_SHAPE_DISK || k == PBRS_SHAPE_TRIANGLE || k == PBRS_SHAPE_QUAD))
            return fail(c, PBRS_E_INVALID, "area light shape kind");
    }
    // Depth of the per-lane stack, from the trees themselves (the heights in the description are not trusted: an entry
    // too few would let a lane write into its neighbour's LDS).  h = levels of a tree (a lone leaf: 1).  A walk pops a node
    // of level l with l - 1 entries pending and pushes two: at most h_tlas entries in the TLAS, h_tlas - 1 pending below an
    // instance, and h_blas more inside it — max(h_tlas, h_tlas - 1 + h_blas) entries.  One level matters: C4's 4 + 23 levels
    // need 26 KB per block, six blocks per CU instead of five.
    auto tree_heights = [](const pbrs_node* nodes, uint32_t n, std::vector<uint32_t>& h) {
        h.assign(n, 1u);  // children come after their parent (left = i + 1, right = a > i, checked above): one reverse pass
        for (uint32_t i = n; i-- > 0;)
            if (!(nodes[i].b & PBRS_LEAF_FLAG)) h[i] = std::max(h[i + 1], h[nodes[i].a]) + 1u;
    };
    for (uint32_t i = 0; i < d->n_tlas_nodes; ++i)
        if (!(d->tlas_nodes[i].b & PBRS_LEAF_FLAG) && d->tlas_nodes[i].a <= i) return fail(c, PBRS_E_INVALID, "tlas nodes are not in pre-order");
    for (uint32_t i = 0; i < d->n_blas_nodes; ++i)
        if (!(d->blas_nodes[i].b & PBRS_LEAF_FLAG) && d->blas_nodes[i].a <= i) return fail(c, PBRS_E_INVALID, "blas nodes are not in pre-order");
    std::vector<uint32_t> th, bh;
    tree_heights(d->tlas_nodes, d->n_tlas_nodes, th);
    tree_heights(d->blas_nodes, d->n_blas_nodes, bh);
    const uint32_t tlas_levels = th[0];
    max_blas_height = 0;
    for (uint32_t i = 0; i < d->n_meshes; ++i) max_blas_height = std::max(max_blas_height, bh[d->meshes[i].root]);
    // a walk enters a BLAS through the instance's own blas_root (range-checked above), which need not be a listed mesh root
    for (uint32_t i = 0; i < d->n_instances; ++i)
        if (d->instances[i].shape_kind == PBRS_SHAPE_MESH) max_blas_height = std::max(max_blas_height, bh[d->instances[i].blas_root]);
    uint32_t depth = std::max(tlas_levels, tlas_levels - 1u + max_blas_height);
    // A ParallelQuad reports hits in the mirrored quadrants of its plane, outside its own box (D1); a mesh may return a hit beyond the
    // extent it was given, which RAISES ray.t_max when its subtree is a left one (bvh.rs:84-88).  Together they make the rise visible
    // (a box the best hit would have pruned is entered and holds a nearer hit: fuzz seed 211699), so the closest-hit walks of such a
    // scene follow ray.t_max to the letter (PBRS_FEAT_EXTENT): a pending TLAS entry then takes two stack words.
    bool has_quad = false, has_mesh = false;
    for (uint32_t i = 0; i < d->n_instances; ++i) {
        has_quad = has_quad || d->instances[i].shape_kind == PBRS_SHAPE_QUAD;
        has_mesh = has_mesh || d->instances[i].shape_kind == PBRS_SHAPE_MESH;
    }
    const bool exact_extent = has_quad && has_mesh;
    if (exact_extent) depth += tlas_levels + 1u;
    if ((size_t)depth * kBlock * sizeof(uint32_t) > kLdsBytesPerCU / 2) return fail(c, PBRS_E_LIMIT, "traversal stack exceeds the LDS budget");
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->second_stream);
    free_scene(c);
    DevScene S{};
    int rc;
    // The division-free box test (device/traverse.h) is exact when every node coordinate b is finite, |b| <= 2^40 and (b == 0 or
    // |b| >= 2^-60) — the range of a ray's origin components (origin_in_range); otherwise every lane uses the literal divisions.
    // With o and b both zero or at least 2^-60 the numerator RN(o - b) is zero or at least 2^-83, its first quotient q0 = nn nr at least
    // 2^-123 (normal: rounded at full precision), the residual e = d q0 + nn a multiple of 2^-131 (exact, if subnormal: the kernels run
    // with f32 denormals on, .amdhsa_float_denorm_mode_32 3) and the result normal: the three instructions return RN(n / d) as they
    // do at any other scale (tools/microbench/div_exhaustive.hip).  Rounds 1-3 asked 2^-20 of the box coordinates — a bound of the
    // f64 route of rounds 1-2 that the f32 quotient inherited: c4xl's 8.4 M vertices hold three heights below it (1.6e-7, 7.4e-7,
    // -9.8e-8), and the WHOLE scene walked on the literal divisions, its lean node steps sitting idle (round 3's "-11 % out of cache").
    {
        auto coord_ok = [](float b) {
            uint32_t u = pn_bits(b) & 0x7fffffffu, e = u >> 23;
            return u == 0u || (e >= 127u - 60u && e <= 127u + 40u);
        };
        bool ok = true;
        for (uint32_t i = 0; i < d->n_tlas_nodes && ok; ++i)
            for (int a = 0; a < 3; ++a) ok = ok && coord_ok(d->tlas_nodes[i].min[a]) && coord_ok(d->tlas_nodes[i].max[a]);
        for (uint32_t i = 0; i < d->n_blas_nodes && ok; ++i)
            for (int a = 0; a < 3; ++a) ok = ok && coord_ok(d->blas_nodes[i].min[a]) && coord_ok(d->blas_nodes[i].max[a]);
        S.fast_slab = ok ? 1u : 0u;
    }
    S.exact_extent = exact_extent ? 1u : 0u;
    uint32_t wide_levels = 0;  // wide nodes on the longest way down a BLAS (0: no wide nodes were built)
    uint64_t walk_bytes = 0;   // what the walks read: nodes, wide nodes, triangle vertices, instance records
    size_t n_scene_nodes = 0;  // DevScene::nodes: TLAS + its leaf copies + every BLAS
    {
        // DevScene::nodes: the TLAS, then its leaves alone in pre-order when the TLAS is small (the shared scan), then every
        // BLAS, in one array with absolute links — a walk reads nodes + index whatever tree it is in.
        const bool scan = d->n_instances >= PBRS_FLAT_TLAS_MIN && d->n_instances <= PBRS_FLAT_TLAS_MAX_ANYHIT;
        std::vector<pbrs_node> nodes(d->tlas_nodes, d->tlas_nodes + d->n_tlas_nodes);
        S.flat_off = (uint32_t)nodes.size();
        if (scan)
            for (uint32_t i = 0; i < d->n_tlas_nodes; ++i)
                if (d->tlas_nodes[i].b & PBRS_LEAF_FLAG) nodes.push_back(d->tlas_nodes[i]);
        S.n_flat = (uint32_t)nodes.size() - S.flat_off;
        const uint64_t blas_off = nodes.size();
        if (blas_off + d->n_blas_nodes > 0x7fffffffull) return fail(c, PBRS_E_LIMIT, "too many BVH nodes");
        nodes.insert(nodes.end(), d->blas_nodes, d->blas_nodes + d->n_blas_nodes);
        for (size_t i = blas_off; i < nodes.size(); ++i)
            if (!(nodes[i].b & PBRS_LEAF_FLAG)) nodes[i].a += (uint32_t)blas_off;  // right child; the left one is i + 1
        if (nodes.size() * sizeof(pbrs_node) >= (1ull << 32)) return fail(c, PBRS_E_LIMIT, "too many BVH nodes (the walks address them with 32-bit byte offsets)");
        if ((rc = upload(c, nodes.data(), nodes.size(), &S.nodes))) return rc;
        n_scene_nodes = nodes.size();
        walk_bytes = nodes.size() * sizeof(pbrs_node) + (uint64_t)d->n_triangles * sizeof(pbrs_tri_verts) + (uint64_t)d->n_instances * sizeof(pbrs_instance);
        std::vector<pbrs_instance> inst(d->instances, d->instances + d->n_instances);
        // Shading classes: one per distinct lobe signature among the materials (class 0: no lobes — emitters — and misses)
        std::vector<uint32_t> mat_class(d->n_materials, 0u);
        {
            std::vector<std::string> sigs;
            for (uint32_t m = 0; m < d->n_materials; ++m) {
                const pbrs_material& mt = d->materials[m];
                if (mt.n_bxdfs == 0) continue;
                std::string sig;
                for (uint32_t k = 0; k < mt.n_bxdfs; ++k) {
                    const pbrs_bxdf& bx = d->bxdfs[mt.first_bxdf + k];
                    sig += (char)('a' + bx.kind);
                    sig += (char)('a' + (bx.kind == PBRS_BXDF_SPECULAR ? bx.intrusion : 0u));
                    sig += (char)('a' + (bx.kind == PBRS_BXDF_DIFFUSE ? bx.oren_nayar : bx.fresnel));
                    sig += (char)('a' + (bx.kind == PBRS_BXDF_MICROFACET && bx.alpha_x != bx.alpha_y ? 1 : 0));
                    sig += bx.tex ? 't' : '-';
                }
                size_t at = 0;
                while (at < sigs.size() && sigs[at] != sig) ++at;
                if (at == sigs.size()) sigs.push_back(sig);
                mat_class[m] = (uint32_t)std::min<size_t>(at + 1, PBRS_MAX_CLASSES - 1);
            }
            S.n_classes = (uint32_t)std::min<size_t>(sigs.size(), PBRS_MAX_CLASSES - 1);
            // the class of the materials that are one untextured Lambertian DiffuseReflect (signature: kind 1, not Oren-Nayar)
            c->lambert_class = 0;
            const std::string lam_sig = {(char)('a' + PBRS_BXDF_DIFFUSE), 'a', 'a', 'a', '-'};
            for (size_t at = 0; at < sigs.size() && at + 1 < PBRS_MAX_CLASSES - 1; ++at)
                if (sigs[at] == lam_sig) c->lambert_class = (uint32_t)at + 1;
            // ... and of the materials that are one Fourier BSDF (material/src/lib.rs:451-475: whatever their tables, one signature)
            c->fourier_class = 0;
            const std::string fou_sig = {(char)('a' + PBRS_BXDF_FOURIER), 'a', 'a', 'a', '-'};
            for (size_t at = 0; at < sigs.size() && at + 1 < PBRS_MAX_CLASSES - 1; ++at)
                if (sigs[at] == fou_sig) c->fourier_class = (uint32_t)at + 1;
        }
        for (pbrs_instance& in : inst) {
            in.pad[0] = mat_class[in.material];
            if (in.shape_kind == PBRS_SHAPE_MESH) in.blas_root += (uint32_t)blas_off;
            bool linear_identity = true;  // bit patterns: -0.0 would not do
            for (int r = 0; r < 3; ++r)
                for (int k = 0; k < 3; ++k) linear_identity = linear_identity && pn_bits(in.inv[r][k]) == pn_bits(r == k ? 1.0f : 0.0f);
            in.flags &= ~PBRS_INSTANCE_TRANSLATION;
            if (linear_identity) in.flags |= PBRS_INSTANCE_TRANSLATION;
        }
        // Four-wide nodes over every BLAS a mesh instance enters (device/wide.h): pad[1] of the device copy of the instance is
        // the wide node of its root, PBRS_WREF_NONE where the mesh is a single leaf.  Built and uploaded only for scenes whose
        // k_shadow can walk them: a scanned TLAS and coordinates inside the guarded range of the division-free box test (the
        // deciding PBRS_WIDE_MIN_LEVELS is known once they are built).  A wide array of 4 GiB or more (32-bit byte offsets) is not
        // an error: the scene keeps the binary walks.
        for (pbrs_instance& in : inst) in.pad[1] = PBRS_WREF_NONE;
        S.wide_cap = 4u;
        if (scan && S.fast_slab != 0u) {
            std::vector<pbrs_wnode> wide;
            std::map<uint32_t, uint32_t> wide_of_root;
            uint32_t levels = 0;
            for (pbrs_instance& in : inst) {
                if (in.shape_kind != PBRS_SHAPE_MESH || (nodes[in.blas_root].b & PBRS_LEAF_FLAG)) continue;
                auto it = wide_of_root.find(in.blas_root);
                if (it == wide_of_root.end()) it = wide_of_root.emplace(in.blas_root, build_wide(nodes, in.blas_root, wide, 0u, levels)).first;
                in.pad[1] = it->second;
            }
            const bool use = levels >= PBRS_WIDE_MIN_LEVELS && wide.size() * sizeof(pbrs_wnode) < (1ull << 32);
            if (use) {
                const pbrs_wnode* dev = nullptr;
                if ((rc = upload(c, wide.data(), wide.size(), &dev))) return rc;
                S.wnodes = dev;
                walk_bytes += wide.size() * sizeof(pbrs_wnode);
                // a node step pushes up to three survivors per level; deeper stacks than PBRS_WIDE_STACK_MAX entries are not given LDS:
                // a ray that would need one (none on the BASELINE scenes) is traced by the binary-walk kernel instead
                S.wide_cap = std::max(4u, std::min(3u * levels + 1u, (uint32_t)PBRS_WIDE_STACK_MAX));
                wide_levels = levels;
            } else {
                for (pbrs_instance& in : inst) in.pad[1] = PBRS_WREF_NONE;
            }
        }
        if ((rc = upload(c, inst.data(), inst.size(), &S.inst))) return rc;
    }
    if ((rc = upload(c, d->shapes, d->n_shapes, &S.shapes))) return rc;
    if ((rc = upload(c, d->meshes, d->n_meshes, &S.meshes))) return rc;
    if ((rc = upload(c, d->tri_verts, d->n_triangles, &S.tv))) return rc;
    if ((rc = upload(c, d->tri_shade, d->n_triangles, &S.ts))) return rc;
    if ((rc = upload(c, d->materials, d->n_materials, &S.mats))) return rc;
    if ((rc = upload(c, d->bxdfs, d->n_bxdfs, &S.bxdfs))) return rc;
    if ((rc = upload(c, d->area_lights, d->n_area_lights, &S.alights))) return rc;
    if ((rc = upload(c, d->delta_lights, d->n_delta_lights, &S.dlights))) return rc;
    S.n_area = d->n_area_lights;
    S.n_delta = d->n_delta_lights;
    if ((rc = upload(c, d->textures, d->n_textures, &S.textures))) return rc;
    if ((rc = upload(c, d->tex_floats, d->n_tex_floats, &S.tex_floats))) return rc;
    if ((rc = upload(c, d->tex_words, d->n_tex_words, &S.tex_words))) return rc;
    if ((rc = upload(c, d->fourier_tables, d->n_fourier_tables, &S.fourier))) return rc;
    S.env_kind = d->env_kind;
    S.env_texture = d->env_texture;
    std::memcpy(S.env_scale, d->env_scale, sizeof S.env_scale);
    std::memcpy(S.env, d->env_constant, sizeof S.env);
    // Scene::has_env_light for EnvLight::Constant (scene/src/lib.rs:96-102): !c.is_black()
    S.has_env = (d->env_kind != PBRS_ENV_CONSTANT || !(S.env[0] <= 0.0f && S.env[1] <= 0.0f && S.env[2] <= 0.0f)) ? 1u : 0u;
    S.refill_below = max_blas_height >= PBRS_LONG_WALK_HEIGHT ? PBRS_REFILL_BELOW_LONG : PBRS_REFILL_BELOW_SHORT;
    S.refill_below_shadow = max_blas_height >= PBRS_LONG_WALK_HEIGHT ? PBRS_REFILL_BELOW_LONG_SHADOW : PBRS_REFILL_BELOW_SHORT;
    if (const char* e = dev_env("PBRS_REFILL_BELOW")) S.refill_below = S.refill_below_shadow = (uint32_t)std::atoi(e);  // developer override (A/B timing)
    // long walks: the levels a ray actually walks — the deepest BLAS, plus the TLAS where it is not scanned
    c->long_walks = (S.n_flat ? 0u : tlas_levels) + max_blas_height >= PBRS_LONG_WALK_HEIGHT;
    if (const char* e = dev_env("PBRS_LONG_WALKS")) c->long_walks = std::atoi(e) != 0;  // developer override (A/B timing)
    // lean further node steps for rays on the division-free box test; a scene whose coordinates leave its guarded range walks every
    // ray on the literal divisions, which the lean steps do not carry: full steps (kernels.h)
    c->walk_bytes = walk_bytes;
    c->full_steps = S.fast_slab == 0u;
    if (const char* e = dev_env("PBRS_FULL_STEPS")) c->full_steps = std::atoi(e) != 0;  // developer override (A/B timing)
    c->overlap_from = walk_bytes <= (4ull << 20) ? 2u : 4u;  // one XCD's L2 holds the arrays the walks read, or not (pbrs_ctx::overlap_from)
    if (const char* e = dev_env("PBRS_OVERLAP_FROM")) c->overlap_from = (uint32_t)std::atoi(e);  // developer override (A/B timing)
    // the leaf copies serve k_shadow up to PBRS_FLAT_TLAS_MAX_ANYHIT instances, k_extend up to PBRS_FLAT_TLAS_MAX
    c->shadow_flat = S.n_flat != 0u;
    const uint32_t flat_feature = (S.n_flat != 0u && d->n_instances <= PBRS_FLAT_TLAS_MAX) ? PBRS_FEAT_FLAT_TLAS : 0u;
    S.features = flat_feature;
    // the walks over four-wide nodes: scenes whose TLAS the stage scans and whose coordinates admit the division-free box test
    // ... and that have a BLAS deep enough for it to matter (PBRS_WIDE_MIN_LEVELS wide nodes on the way down: meshes of a few
    // triangles are a leaf or two, where the binary walks at their six waves per SIMD are faster — C2: 105 against 140 ms)
    // k_shadow gains (C4: 250 -> 236 ms per frame at five waves per SIMD); k_extend, whose wide walk needs 117 registers (four
    // waves per SIMD, or 72 bytes of spills at five), loses against the binary walk at six (459 -> 506 ms) and keeps the binary
    // walk: its wide kernels exist in developer builds only (PBRS_WIDE bit 0).
    const bool wide_ok = S.fast_slab != 0u && S.wnodes != nullptr && wide_levels >= PBRS_WIDE_MIN_LEVELS;
    c->wide_extend = false;
    c->wide_shadow = c->shadow_flat && wide_ok;
#ifdef PBRS_DEV_OVERRIDES
    if (const char* e = dev_env("PBRS_WIDE")) {  // developer override (A/B timing): bit 0 k_extend, bit 1 k_shadow
        c->wide_extend = flat_feature != 0u && wide_ok && (std::atoi(e) & 1);
        c->wide_shadow = c->wide_shadow && (std::atoi(e) & 2);
    }
#endif
    for (uint32_t i = 0; i < d->n_instances; ++i) {
        const pbrs_instance& in = d->instances[i];
        if (in.shape_kind == PBRS_SHAPE_MESH) {
            if (!(in.mesh_flags & PBRS_MESH_SHADING_OK_MASK)) S.features |= PBRS_FEAT_SHADING_CHECK;
        } else if (in.shape_kind != PBRS_SHAPE_TRIANGLE) {  // isolated triangles go through the triangle-record path
            S.features |= PBRS_FEAT_ANALYTIC;
        }
    }
    // k_shade specialisation: every lobe an untextured Lambertian DiffuseReflect (at most one per material); every area light
    // of one shape
    {
        bool lambert = !textured;
        for (uint32_t i = 0; i < d->n_materials && lambert; ++i) {
            const pbrs_material& m = d->materials[i];  // its lobes only: the array also holds the visualisers' records
            lambert = m.n_bxdfs <= 1;
            for (uint32_t k = 0; k < m.n_bxdfs && lambert; ++k) {
                const pbrs_bxdf& bx = d->bxdfs[m.first_bxdf + k];
                lambert = bx.kind == PBRS_BXDF_DIFFUSE && bx.oren_nayar == 0 && bx.tex == 0;
            }
        }
        uint32_t light_spec = 0u;
        if (d->n_area_lights) {
            const uint32_t k0 = d->area_lights[0].shape_kind;
            bool same = true;
            for (uint32_t i = 1; i < d->n_area_lights; ++i) same = same && d->area_lights[i].shape_kind == k0;
            if (same && k0 == PBRS_SHAPE_SPHERE) light_spec = PBRS_SHADE_LIGHT_SPHERE;
            if (same && k0 == PBRS_SHAPE_TRIANGLE) light_spec = PBRS_SHADE_LIGHT_TRIANGLE;
        }
        c->light_spec = light_spec;
        // the light shape alone does not pay: without the Lambert cut the kernel grows to 135-141 VGPRs, three waves per SIMD
        // (C2 shade 110.5 -> 117.0 ms, C4 150.6 -> 169.3)
        uint32_t spec = lambert ? (PBRS_SHADE_LAMBERT | light_spec) : 0u;
        if (const char* e = dev_env("PBRS_SHADE_SPEC")) spec &= (uint32_t)std::atoi(e);  // developer override (A/B timing): a mask
        c->shade_spec = spec;
    }
    // The arrays the walks read, staged in every block's LDS (kernels.h, stage_scene) where they fit next to the stack rows with
    // eight blocks to a CU: scenes of a few KB whose walks are short (no wide nodes, lean-step choice irrelevant).
    {
        const size_t stack_bytes = (size_t)depth * kBlock * sizeof(uint32_t);
        const size_t scene_bytes = n_scene_nodes * sizeof(pbrs_node) + (size_t)d->n_triangles * sizeof(pbrs_tri_verts) + (size_t)d->n_instances * sizeof(pbrs_instance) +
                                   (size_t)d->n_shapes * sizeof(pbrs_shape);
        c->lds_scene = !c->wide_shadow && !c->wide_extend && !c->full_steps && stack_bytes + scene_bytes <= kLdsBytesPerCU / 8;
        if (const char* e = dev_env("PBRS_LDS_SCENE")) c->lds_scene = c->lds_scene && std::atoi(e) != 0;  // developer override (A/B timing)
        c->lds_scene_bytes = c->lds_scene ? scene_bytes : 0;
        // ... or the TLAS alone, where it is too large for the wave's shared scan (no leaf copies) and fits with seven blocks to a CU
        const size_t top_bytes = (size_t)d->n_tlas_nodes * sizeof(pbrs_node);
        c->lds_top = !c->lds_scene && S.n_flat == 0u && !c->full_steps && stack_bytes + top_bytes + 512 <= kLdsBytesPerCU / 7;
        if (const char* e = dev_env("PBRS_LDS_TOP")) c->lds_top = c->lds_top && std::atoi(e) != 0;  // developer override (A/B timing)
        c->lds_top_bytes = c->lds_top ? top_bytes : 0;
        S.lds_off_words = depth * kBlock;
        S.lds_nodes = c->lds_scene ? (uint32_t)n_scene_nodes : c->lds_top ? d->n_tlas_nodes : 0u;
        S.lds_tris = c->lds_scene ? d->n_triangles : 0u;
        S.lds_inst = c->lds_scene ? d->n_instances : 0u;
        S.lds_shapes = c->lds_scene ? d->n_shapes : 0u;
    }
    // k_shade: the shading records (instances, shapes, materials, lobes, lights) in LDS where they are a few KB, the triangle records
    // too where everything is (kernels.h, stage_shade_scene); five blocks of the Lambert variants share a CU's 160 KB with the rest
    {
        S.n_inst = d->n_instances; S.n_shapes = d->n_shapes; S.n_tris = d->n_triangles; S.n_mats = d->n_materials; S.n_bxdfs = d->n_bxdfs;
        const size_t rec = (size_t)d->n_instances * sizeof(pbrs_instance) + (size_t)d->n_shapes * sizeof(pbrs_shape) + (size_t)d->n_materials * sizeof(pbrs_material) +
                           (size_t)d->n_bxdfs * sizeof(pbrs_bxdf) + (size_t)d->n_area_lights * sizeof(pbrs_area_light) + (size_t)d->n_delta_lights * sizeof(pbrs_delta_light);
        const size_t tris = (size_t)d->n_triangles * (sizeof(pbrs_tri_verts) + sizeof(pbrs_tri_shade));
        const size_t budget = 16u << 10;
        c->shade_lds = rec + tris <= budget ? (PBRS_SHADE_LDS_RECORDS | PBRS_SHADE_LDS_TRIS) : rec <= budget ? PBRS_SHADE_LDS_RECORDS : 0u;
        if (const char* e = dev_env("PBRS_SHADE_LDS")) c->shade_lds &= (uint32_t)std::atoi(e);  // developer override (A/B timing): a mask
        c->shade_lds_bytes = (c->shade_lds & PBRS_SHADE_LDS_TRIS) ? rec + tris : c->shade_lds ? rec : 0;
    }
    c->S = S;
    c->textured = textured;
    c->fourier = fourier;
    c->has_vis_records = vis_records;
    c->stack_depth = depth;
    c->has_scene = true;
    c->split_decision = 0;  // (the stream was synchronised above: no probe of the previous scene is in flight)
    c->split_probe_in_flight = false;
    return PBRS_OK;
}

int pbrs_render_tile_device(pbrs_ctx* c, const pbrs_camera* cam, const pbrs_render_params* p, float* rgb_out_device, pbrs_stats* stats_out) {
    if (!c) return PBRS_E_INVALID;
    if (!rgb_out_device) return fail(c, PBRS_E_INVALID, "null output");
    int rc = render_common(c, cam, p, rgb_out_device);
    if (rc) return rc;
    if (stats_out) return collect(c, stats_out);
    return PBRS_OK;
}

int pbrs_render_tile(pbrs_ctx* c, const pbrs_camera* cam, const pbrs_render_params* p, float* rgb_out_host, pbrs_stats* stats_out) {
    if (!c) return PBRS_E_INVALID;
    if (!rgb_out_host) return fail(c, PBRS_E_INVALID, "null output");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = check_params(c, cam, p);
    if (rc) return rc;
    rc = ensure_work(c, (size_t)p->w * p->h * auto_samples_per_pass(c, p), (size_t)p->w * p->h);
    if (rc) return rc;
    rc = render_common(c, cam, p, c->rgb_dev);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(rgb_out_host, c->rgb_dev, 3 * (size_t)p->w * p->h * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    return collect(c, stats_out);
}

int pbrs_collect_stats(pbrs_ctx* c, pbrs_stats* stats_out) {
    if (!c) return PBRS_E_INVALID;
    return collect(c, stats_out);
}

int pbrs_intersect_rays(pbrs_ctx* c, uint32_t n, const float* origins, const float* dirs, const float* tmax, pbrs_hit_record* hits_out,
                        uint8_t* occluded_out) {
    if (!c) return PBRS_E_INVALID;
    if (!c->has_scene) return fail(c, PBRS_E_NO_SCENE, "no scene uploaded");
    if (n == 0) return PBRS_OK;
    if (!origins || !dirs || !tmax) return fail(c, PBRS_E_INVALID, "null ray arrays");
    HIPCHK(c, hipSetDevice(c->device));
    float4 *d_o = nullptr, *d_d = nullptr;
    float* d_t = nullptr;
    pbrs_hit_record* d_h = nullptr;
    uint8_t* d_occ = nullptr;
    uint32_t* d_info = nullptr;
    int rc = PBRS_OK;
    auto cleanup = [&]() {
        (void)hipFree(d_o); (void)hipFree(d_d); (void)hipFree(d_t); (void)hipFree(d_h); (void)hipFree(d_occ); (void)hipFree(d_info);
    };
#define TRY(expr)                                                                  \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) {                                                    \
            c->error = std::string(#expr) + ": " + hipGetErrorString(e_);          \
            (void)hipGetLastError();                                               \
            cleanup();                                                             \
            return PBRS_E_DEVICE;                                                  \
        }                                                                          \
    } while (0)
    // rays travel as 16-byte records, the form the walks re-read their ray in (traverse.h, reload_world)
    std::vector<float4> h_o(n), h_d(n);
    for (uint32_t i = 0; i < n; ++i) {
        h_o[i] = make_float4(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2], 0.0f);
        h_d[i] = make_float4(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], 0.0f);
    }
    TRY(hipMalloc(reinterpret_cast<void**>(&d_o), (size_t)n * 16));
    TRY(hipMalloc(reinterpret_cast<void**>(&d_d), (size_t)n * 16));
    TRY(hipMalloc(reinterpret_cast<void**>(&d_t), (size_t)n * 4));
    if (hits_out) TRY(hipMalloc(reinterpret_cast<void**>(&d_h), (size_t)n * sizeof(pbrs_hit_record)));
    if (occluded_out) TRY(hipMalloc(reinterpret_cast<void**>(&d_occ), (size_t)n));
    TRY(hipMalloc(reinterpret_cast<void**>(&d_info), 2 * sizeof(uint32_t)));
    TRY(hipMemsetAsync(d_info, 0, 2 * sizeof(uint32_t), c->stream));
    TRY(hipMemcpyAsync(d_o, h_o.data(), (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
    TRY(hipMemcpyAsync(d_d, h_d.data(), (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
    TRY(hipMemcpyAsync(d_t, tmax, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    {
        // the walks the pipeline runs for this scene: over four-wide nodes where the TLAS is scanned (the binary walks take what
        // those refuse, as in the pipeline), else the binary walks
        const uint32_t rows = std::max(c->stack_depth, c->S.wide_cap);
        const size_t lds = (size_t)(rows + c->S.n_flat) * kBlock * sizeof(uint32_t);
        const dim3 grid(std::min<uint32_t>((n + kBlock - 1) / kBlock, kPersistentBlocks));
        // each query through the walk its stage runs in the pipeline (run_pass): occlusion through the four-wide any-hit walk whenever
        // k_shadow takes it, closest hits through the binary walk (the four-wide closest walk exists in developer builds only)
        c->last_intersect = pbrs_intersect_info{c->wide_shadow ? 1u : 0u, c->wide_extend ? 1u : 0u, 0u, 0u};
#define PBRS_LAUNCH_RAYS(WC, WA) hipLaunchKernelGGL((k_intersect_rays<WC, WA>), grid, dim3(kBlock), lds, c->stream, c->S, n, d_o, d_d, d_t, d_h, d_occ, d_info)
#ifdef PBRS_DEV_OVERRIDES
        if (c->wide_extend && c->wide_shadow) PBRS_LAUNCH_RAYS(true, true);
        else if (c->wide_extend) PBRS_LAUNCH_RAYS(true, false);
        else
#endif
        if (c->wide_shadow) PBRS_LAUNCH_RAYS(false, true);
        else PBRS_LAUNCH_RAYS(false, false);
#undef PBRS_LAUNCH_RAYS
    }
    TRY(hipGetLastError());
    if (hits_out) TRY(hipMemcpyAsync(hits_out, d_h, (size_t)n * sizeof(pbrs_hit_record), hipMemcpyDeviceToHost, c->stream));
    if (occluded_out) TRY(hipMemcpyAsync(occluded_out, d_occ, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    uint32_t slow[2] = {0u, 0u};
    TRY(hipMemcpyAsync(slow, d_info, sizeof slow, hipMemcpyDeviceToHost, c->stream));
    TRY(hipStreamSynchronize(c->stream));
    c->last_intersect.slow_any = slow[0];
    c->last_intersect.slow_closest = slow[1];
    cleanup();
    return rc;
}

int pbrs_last_intersect_info(const pbrs_ctx* c, pbrs_intersect_info* out) {
    if (!c || !out) return PBRS_E_INVALID;
    *out = c->last_intersect;
    return PBRS_OK;
}

int pbrs_camera_rays(pbrs_ctx* c, const pbrs_camera* cam, const pbrs_render_params* p, uint32_t sample_index, float* origins_out, float* dirs_out) {
    if (!c) return PBRS_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = check_params(c, cam, p);
    if (rc) return rc;
    if (!origins_out || !dirs_out) return fail(c, PBRS_E_INVALID, "null output");
    const uint32_t P = p->w * p->h;
    rc = ensure_work(c, P, P);
    if (rc) return rc;
    RenderConst k = make_const(cam, p);
    k.tiles8_per_row = 0u;  // one sample index, exported by pixel: slot = pixel
    k.pass_first_sample = sample_index;
    k.n_slots = P;
    hipLaunchKernelGGL(k_raygen, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, c->st, k);
    float *d_o = nullptr, *d_d = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_o); (void)hipFree(d_d); };
    TRY(hipMalloc(reinterpret_cast<void**>(&d_o), (size_t)P * 12));
    TRY(hipMalloc(reinterpret_cast<void**>(&d_d), (size_t)P * 12));
    hipLaunchKernelGGL(k_export_rays, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, c->st, P, d_o, d_d);
    TRY(hipGetLastError());
    TRY(hipMemcpyAsync(origins_out, d_o, (size_t)P * 12, hipMemcpyDeviceToHost, c->stream));
    TRY(hipMemcpyAsync(dirs_out, d_d, (size_t)P * 12, hipMemcpyDeviceToHost, c->stream));
    TRY(hipStreamSynchronize(c->stream));
    cleanup();
    return PBRS_OK;
}

int pbrs_numeric_eval(pbrs_ctx* c, uint32_t fn, uint32_t n, const float* x, const float* y, float* out) {
    if (!c) return PBRS_E_INVALID;
    if (n == 0) return PBRS_OK;
    if (!x || !out || fn > 15) return fail(c, PBRS_E_INVALID, "bad numeric_eval arguments");
    HIPCHK(c, hipSetDevice(c->device));
    float *d_x = nullptr, *d_y = nullptr, *d_r = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_x); (void)hipFree(d_y); (void)hipFree(d_r); };
    TRY(hipMalloc(reinterpret_cast<void**>(&d_x), (size_t)n * 4));
    TRY(hipMalloc(reinterpret_cast<void**>(&d_r), (size_t)n * 4));
    TRY(hipMemcpyAsync(d_x, x, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    if (y) {
        TRY(hipMalloc(reinterpret_cast<void**>(&d_y), (size_t)n * 4));
        TRY(hipMemcpyAsync(d_y, y, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    }
    hipLaunchKernelGGL(k_numeric_eval, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, fn, n, d_x, d_y, d_r);
    TRY(hipGetLastError());
    TRY(hipMemcpyAsync(out, d_r, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    TRY(hipStreamSynchronize(c->stream));
    cleanup();
    return PBRS_OK;
}

int pbrs_render_sample_radiance(pbrs_ctx* c, const pbrs_camera* cam, const pbrs_render_params* p, uint32_t sample_index, float* rgb_out_host) {
    if (!c) return PBRS_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = check_params(c, cam, p);
    if (rc) return rc;
    if (!rgb_out_host) return fail(c, PBRS_E_INVALID, "null output");
    const uint32_t P = p->w * p->h;
    rc = ensure_work(c, P, P);
    if (rc) return rc;
    RenderConst k = make_const(cam, p);
    k.tiles8_per_row = 0u;  // one sample index, exported by pixel: slot = pixel
    Timer tm{c, false};
    HIPCHK(c, hipMemsetAsync(c->sum, 0, 3 * (size_t)P * sizeof(float), c->stream));
    rc = run_pass(c, k, sample_index, 1, false, tm);
    if (rc) return rc;
    hipLaunchKernelGGL(k_export_radiance, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, c->st, P, c->rgb_dev);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(rgb_out_host, c->rgb_dev, 3 * (size_t)P * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PBRS_OK;
}

}  // extern "C"
