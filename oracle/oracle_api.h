/* oracle/oracle_api.h — C entry points of the CPU oracle.  TEST INFRASTRUCTURE ONLY: bound by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by pbrs_amd/. */
#ifndef PBRS_ORACLE_API_H
#define PBRS_ORACLE_API_H

#include <stdint.h>

#include "../include/pbrs_scene_spec.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_scene oracle_scene;

typedef struct oracle_stats {
    uint64_t closest_rays, shadow_rays, tlas_nodes, blas_nodes, instances, instance_hits;
    uint64_t triangles, spheres, quads, cuboids, disks, tri_shading, shade_events, samples;
    uint64_t panics;        /* reference assert!/panic! sites reached */
    uint64_t tlas_ties;     /* tlas/src/bvh.rs:94 reached with equal t on both sides */
    uint64_t sphere_inside; /* D4: interior sphere hits (Interaction::new assert skipped) */
    uint64_t nonfinite_samples; /* camera samples whose radiance has a NaN or infinite component (pbrs_stats.invalid_samples) */
} oracle_stats;

typedef struct oracle_hit_record {
    float t;
    uint32_t inst;
    uint32_t prim;
    float b1, b2;
} oracle_hit_record;

#define ORACLE_TRACE_MAX_BOUNCES 16
typedef struct oracle_bounce_trace {
    uint32_t hit;
    float t;
    uint32_t inst, prim;
    float b1, b2;
    float pos[3], normal[3];
    float radiance_after_nee[3];
    float f[3], wi[3];
    float pr;
    uint32_t pr_is_mass;
    float beta_after[3];
} oracle_bounce_trace;
typedef struct oracle_path_trace {
    float ray_o[3], ray_d[3];
    uint32_t n_bounces;
    oracle_bounce_trace bounce[ORACLE_TRACE_MAX_BOUNCES];
    float radiance[3];
    uint32_t panics;
} oracle_path_trace;

oracle_scene* oracle_scene_build(const pbrs_scene_spec* spec);
void oracle_scene_free(oracle_scene*);
uint32_t oracle_tlas_height(const oracle_scene*);

int oracle_render_tile(const oracle_scene*, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint32_t strata_x, uint32_t strata_y,
                       uint32_t max_depth, uint64_t seed, uint32_t nthreads, float* rgb_out, oracle_stats* stats_out);
int oracle_render_tile_integrator(const oracle_scene*, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint32_t strata_x, uint32_t strata_y,
                                  uint32_t max_depth, uint64_t seed, uint32_t nthreads, uint32_t integrator, float* rgb_out,
                                  oracle_stats* stats_out);
int oracle_trace_sample(const oracle_scene*, uint32_t row, uint32_t col, uint32_t sample_index, uint32_t strata_x, uint32_t strata_y,
                        uint32_t max_depth, uint64_t seed, oracle_path_trace* trace);
int oracle_intersect_rays(const oracle_scene*, uint32_t n, const float* origins, const float* dirs, const float* tmax,
                          oracle_hit_record* hits_out, uint8_t* occluded_out, oracle_stats* stats_out, uint8_t* tie_out);
int oracle_camera_rays(const oracle_scene*, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint32_t sample_index, uint32_t strata_x,
                       uint32_t strata_y, uint64_t seed, float* origins, float* dirs);
/* texture/src/lib.rs `Texture::value(uv, p)` for n (uv, p) pairs; scene/src/lib.rs:105-117 for n ray directions. */
int oracle_texture_value(const pbrs_texture_spec* tex, uint32_t n, const float* uv, const float* pos, float* rgb_out);
int oracle_env_eval(const oracle_scene*, uint32_t n, const float* dirs, float* rgb_out);
int oracle_numeric_eval(uint32_t fn, uint32_t n, const float* x, const float* y, float* out);
/* radiometry/src/spectrum.rs: temperature_to_color(kelvin) (:38-55) and sampled_spectrum_to_color over n (lambda, value) samples
 * (:57-70).  Return the number of panic sites reached. */
int oracle_temperature_to_color(float kelvin, float* rgb_out);
int oracle_spd_to_color(uint32_t n, const float* lambdas_nm, const float* values, float* rgb_out);
int oracle_rng_stream(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, float* out);

/* oracle/selftest.cpp: the reference's own known-answer tests, transcribed.  Returns the number of
 * failed checks of test `name` (or of all tests when name is NULL); messages go to `log`. */
int oracle_selftest(const char* name, char* log, uint32_t log_cap);
uint32_t oracle_selftest_count(void);
const char* oracle_selftest_name(uint32_t i);

#ifdef __cplusplus
}
#endif
#endif
