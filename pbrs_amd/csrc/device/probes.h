// device/probes.h — developer probes of the traversal and shading kernels.  Every macro below expands to nothing in the shipped
// library; the instrumented builds are made by tools/trav_probe.py (-DPBRS_PROBE_TRAV), tools/trav_time.py (-DPBRS_PROBE_TIME),
// tools/util_probe.py / util_probe2.py (-DPBRS_PROBE_UTIL[2]) and tools/shade_probe.py (-DPBRS_PROBE_SHADE).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Developer probe (tools/trav_probe.py; -DPBRS_PROBE_TRAV builds only): per-lane event counts of a walk — 0 node steps, 1 box
// tests (a wide node's four count once), 2 of them failed, 3 BLAS leaves that came up, 4 of them with their own box passing,
// 5 triangle tests run as a helper, 6 / 7 boundary steps in / out.
#ifdef PBRS_PROBE_TRAV
#define PBRS_TP_N 8
#define PBRS_TP_FIELDS uint32_t pr[PBRS_TP_N];
#define PBRS_TP(i) (this->pr[i]++)
#else
#define PBRS_TP_FIELDS
#define PBRS_TP(i) \
    do {           \
    } while (0)
#endif

// Developer probe (tools/trav_probe.py; -DPBRS_PROBE_TRAV builds only): what the traversal loops execute, summed over a launch's
// waves.  kp[]: 0 loop rounds, 1 refills, 2 rays started, 3 boundary-step executions, 4 lanes in them, 5 / 6 / 7 lanes in a round's first /
// second / third node step, 8 leaf-step executions, 9 lanes holding a leaf in them, 10 lanes with a walk at the start of a round,
// 11 rounds whose first node step had a lane; then the walks' own eight counters (traverse.h).  [0]: k_extend, [1]: k_shadow.
// ... and where a wave's cycles go (-DPBRS_PROBE_TIME, tools/trav_time.py): s_memtime at the boundaries of the loop's regions —
// 0 refill (retire, fetch, start, scan), 1 boundary step (with its ballots), 2 node steps, 3 leaf step — summed over the waves.
#ifdef PBRS_PROBE_TIME
__device__ unsigned long long g_trav_time[2][8];
#define PBRS_TT_DECL                                       \
    unsigned long long tt[4] = {0ull, 0ull, 0ull, 0ull};   \
    unsigned long long tprev = __builtin_amdgcn_s_memtime()
#define PBRS_TT(i)                                                      \
    do {                                                                \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
        tt[i] += now_ - tprev;                                          \
        tprev = now_;                                                   \
    } while (0)
#define PBRS_TT_FLUSH(which)                                                             \
    do {                                                                                 \
        if ((threadIdx.x & 63u) == 0u)                                                   \
            for (int k_ = 0; k_ < 4; ++k_) atomicAdd(&g_trav_time[which][k_], tt[k_]);   \
    } while (0)
#else
#define PBRS_TT_DECL \
    do {             \
    } while (0)
#define PBRS_TT(i) \
    do {           \
    } while (0)
#define PBRS_TT_FLUSH(which) \
    do {                     \
    } while (0)
#endif
#ifdef PBRS_PROBE_TRAV
__device__ unsigned long long g_trav_probe[2][24];
#define PBRS_KP_DECL(walk)      \
    uint32_t kp[16];            \
    for (int k_ = 0; k_ < 16; ++k_) kp[k_] = 0; \
    for (int k_ = 0; k_ < PBRS_TP_N; ++k_) walk.pr[k_] = 0
#define PBRS_KP_LANE(i, cond)  \
    do {                       \
        if (cond) kp[i]++;     \
    } while (0)
#define PBRS_KP_WAVE(i) PBRS_KP_LANE(i, (threadIdx.x & 63u) == 0u)  /* wave-uniform control flow only */
PD void trav_probe_flush(int which, const uint32_t* kp, const uint32_t* pr) {
    for (int k = 0; k < 24; ++k) {
        uint32_t v = k < 16 ? kp[k] : pr[k - 16];
        unsigned long long t = v;
        for (int o = 32; o; o >>= 1) t += __shfl_xor(t, o, 64);
        if ((threadIdx.x & 63u) == 0u && t) atomicAdd(&g_trav_probe[which][k], t);
    }
}
#define PBRS_KP_FLUSH(which, walk) trav_probe_flush(which, kp, walk.pr)
#else
#define PBRS_KP_DECL(walk) \
    do {                   \
    } while (0)
#define PBRS_KP_LANE(i, cond) \
    do {                      \
    } while (0)
#define PBRS_KP_WAVE(i) \
    do {                \
    } while (0)
#define PBRS_KP_FLUSH(which, walk) \
    do {                           \
    } while (0)
#endif
// developer probe (tools/util_probe.py, instrumented variant only): wave-level executions of the node and leaf steps,
// stashed in the cuboid / disk counters of a scene that has neither
#ifdef PBRS_PROBE_UTIL
#define PBRS_PROBE_ONE(cond, field)                                                                            \
    do {                                                                                                       \
        const uint64_t pm = __ballot(cond);                                                                    \
        if (pm && (threadIdx.x & 63u) == (uint32_t)(__ffsll((unsigned long long)pm) - 1)) cnt.c.field++;       \
    } while (0)
#ifdef PBRS_PROBE_UTIL2  /* lanes per mode at the start of a round, summed by the wave's first lane */
#define PBRS_PROBE_UTIL_COUNT(walk, cnt)                                                          \
    do {                                                                                          \
        if (STATS && (threadIdx.x & 63u) == 0) {                                                  \
            cnt.c.quads += 1;                                                                     \
        }                                                                                         \
        if (STATS) {                                                                              \
            const uint32_t pn_ = (uint32_t)__popcll(__ballot(walk.mode == PBRS_WALK_NODE));      \
            const uint32_t pl_ = (uint32_t)__popcll(__ballot(walk.mode == PBRS_WALK_LEAF));      \
            const uint32_t px_ = (uint32_t)__popcll(__ballot(walk.mode == PBRS_WALK_XFER));      \
            if ((threadIdx.x & 63u) == 0) cnt.c.cuboids += pn_, cnt.c.disks += pl_, cnt.c.tri_shading += px_; \
        }                                                                                         \
    } while (0)
#else
#define PBRS_PROBE_UTIL_COUNT(walk, cnt)                                                          \
    do {                                                                                          \
        if (STATS) {                                                                              \
            PBRS_PROBE_ONE(walk.mode == PBRS_WALK_NODE, cuboids);                                 \
            PBRS_PROBE_ONE(true, quads);                                                          \
        }                                                                                         \
    } while (0)
#endif
#define PBRS_PROBE_XFER_COUNT(cnt)                               \
    do {                                                         \
        if (STATS) PBRS_PROBE_ONE(true, spheres); /* wave-level boundary-step executions */ \
    } while (0)
#ifdef PBRS_PROBE_UTIL2
#define PBRS_PROBE_LEAF_COUNT(cnt) \
    do {                           \
    } while (0)
#else
#define PBRS_PROBE_LEAF_COUNT(cnt)                               \
    do {                                                         \
        if (STATS) PBRS_PROBE_ONE(true, disks); /* wave-level leaf-step executions */ \
    } while (0)
#endif
#else
#define PBRS_PROBE_UTIL_COUNT(walk, cnt) \
    do {                                 \
    } while (0)
#define PBRS_PROBE_LEAF_COUNT(cnt) \
    do {                           \
    } while (0)
#define PBRS_PROBE_XFER_COUNT(cnt) \
    do {                           \
    } while (0)
#endif

// Developer probe (tools/shade_probe.py; -DPBRS_PROBE_SHADE builds only): wall cycles of k_shade's regions, summed per wave.
#ifdef PBRS_PROBE_SHADE
__device__ unsigned long long g_shade_probe[16];
#define PBRS_SHADE_MARK(k)                                          \
    do {                                                            \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        probe_acc[k] += now_ - probe_t;                             \
        probe_t = now_;                                             \
    } while (0)
#else
#define PBRS_SHADE_MARK(k) \
    do {                   \
    } while (0)
#endif
