/* pbrs_gpu.h — C ABI of the MI355X wavefront path tracer (libpbrs_gpu.so).
 *
 * What it replaces.  The reference has no FFI or plugin interface; its only seam on this path is
 * the integrator function pointer `fn(&Scene, ray::Ray, i32) -> Color` chosen at
 * src/main.rs:160-163 and called once per camera sample from the row loop at src/main.rs:192-213,
 * which rayon runs over rows at :219-224.  A per-ray seam cannot feed a GPU, so this ABI lifts it to
 * a per-tile batch seam that replaces src/main.rs:192-231 as a whole:
 *     (scene, camera, strata, depth) -> row-major RGB f32
 * with the same semantics (stratified jitter :197-201, `shoot_ray` :203, sequential f32 sum over
 * samples :205, `scale_down_by` :208).  Each entry point below cites the reference lines it stands
 * for.  INTEGRATION.md shows the Rust `extern "C"` block a maintainer would add.
 *
 * Conventions: POD structs, little-endian, plain pointers + counts, no ownership transfer
 * (`pbrs_upload_scene` copies; the caller keeps its buffers).  Never unwinds or aborts: every call
 * returns 0 or a negative PBRS_E_* and `pbrs_last_error` explains.  The reference's error model is
 * panic: where an `assert!` of the reference would fire, the device carries on with the
 * arithmetic result, as the oracle does.  What comes of it is visible in the output and counted:
 * `pbrs_stats.invalid_samples` is the number of camera samples whose radiance is not finite (a NaN or
 * an infinity in any channel) — always filled, equal to the oracle's count on the same scene and
 * seeds (tests/test_gpu_fuzz.py).  The assert sites themselves are counted by the oracle only
 * (`oracle_stats.panics`, test infrastructure): the parity tests assert that count to be zero on the
 * BASELINE scenes.
 * Threading: one `pbrs_ctx` per GPU (several on one GPU work too, e.g. one per scene); different contexts may be
 * driven concurrently from different host threads/processes (tests/test_gpu_contexts.py: two threads, two contexts,
 * disjoint bands of one frame); a single context is not re-entrant.  Nothing process-wide depends on the scene a
 * context holds (the kernels' dynamic-LDS limit is raised once per device, to the cap, in `pbrs_create`).
 * Stream ordering: a context runs on its own NON-BLOCKING stream.  Work the caller queues on the legacy default stream
 * (or on any other stream) is NOT ordered against a render: the output of `pbrs_render_tile_device` is valid only after
 * `pbrs_collect_stats`, after the caller has synchronised the context's stream, or — with `pbrs_set_stream` — in the
 * caller's own stream's order.  `pbrs_render_tile` (host output) synchronises before it returns.
 * Environment: the library reads no environment variable (developer builds with -DPBRS_DEV_OVERRIDES aside).
 *
 * The flattened scene (`pbrs_scene_desc`) is produced by the host side of the boundary — in the
 * reference's own language that is scene/ + tlas/ + shape/ (Rust); here pbrs_amd/csrc/host
 * (C++, include/pbrs_host.h) — by running `tlas::build_bvh` (tlas/src/bvh.rs:116-152) and
 * `recursive_build` (shape/src/blas.rs:333-420) and linearising the trees.
 */
#ifndef PBRS_GPU_H
#define PBRS_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PBRS_OK 0
#define PBRS_E_INVALID (-1)   /* bad argument / inconsistent scene description */
#define PBRS_E_DEVICE (-2)    /* HIP runtime error (message in pbrs_last_error) */
#define PBRS_E_NO_SCENE (-3)  /* render called before pbrs_upload_scene */
#define PBRS_E_LIMIT (-4)     /* traversal stack deeper than the LDS budget, tile too large, ... */

#define PBRS_LEAF_FLAG 0x80000000u
#define PBRS_TLAS_LEAF_KIND_SHIFT 8 /* TLAS leaves also carry the instance's shape kind in bits 8..10 of b */

/* A BVH node, 32 B, pre-order: the left child of node i is node i+1.
 *   inner: a = index of the right child, b = split axis (BLAS; 0 for TLAS)
 *   leaf : b = PBRS_LEAF_FLAG | count;  a = first triangle (BLAS, count <= n) or instance (TLAS, count = 1)
 * bbox as in geometry/src/bvh.rs:11-14.  TLAS: tlas/src/bvh.rs:11-18.  BLAS: shape/src/blas.rs:10-18. */
typedef struct pbrs_node {
    float min[3];
    uint32_t a;
    float max[3];
    uint32_t b;
} pbrs_node;

/* tlas/src/instance.rs:12-16.  Rows of the two affine Mat4 of geometry/src/transform.rs:16-19:
 * row r = (cols[0][r], cols[1][r], cols[2][r], cols[3][r]); the fourth row is (0,0,0,1). */
typedef struct pbrs_instance {
    float inv[3][4];
    float fwd[3][4];
    uint32_t shape_kind;  /* enum pbrs_shape_kind (pbrs_scene_spec.h) */
    uint32_t shape_index; /* analytic: index into shapes[]; mesh: index into meshes[] */
    uint32_t material;
    uint32_t flags; /* PBRS_INSTANCE_* */
    /* mesh instances: copies of meshes[shape_index].root / .flags so that entering the BLAS costs one record;
     * PBRS_SHAPE_TRIANGLE instances: blas_root = index of the triangle's own record in tri_verts[] */
    uint32_t blas_root;
    uint32_t mesh_flags;
    uint32_t pad[2];
} pbrs_instance;
/* inv and fwd are bit-exactly the identity: for a ray whose components are all finite and non-zero
 * `inverse * ray` (tlas/src/instance.rs:51) returns the ray's own bits, so the product can be skipped. */
#define PBRS_INSTANCE_IDENTITY 1u

/* Analytic shape parameters, 12 floats (shape/src/simple.rs:10-196), laid out as in pbrs_shape_spec.p;
 * cuboid min/max already ordered (Cuboid::from_points), disk normal already unit (Disk::new). */
typedef struct pbrs_shape {
    float p[12];
} pbrs_shape;

typedef struct pbrs_mesh {
    uint32_t root;       /* index of the BLAS root in blas_nodes[] */
    uint32_t n_nodes;
    uint32_t first_tri;  /* first triangle of this mesh in tri_verts[] / tri_shade[] */
    uint32_t n_tris;
    uint32_t height;     /* IsoBvhNode::height(), shape/src/blas.rs:21-26 */
    uint32_t flags;      /* PBRS_MESH_* */
    uint32_t pad[2];
} pbrs_mesh;
/* Every triangle of the mesh has three bit-identical vertex normals AND passes the tangent check of
 * shape/src/blas.rs:193-200 (which then depends on the triangle only, not on the hit or the ray), so the
 * traversal need not evaluate the shading frame of candidate hits. */
#define PBRS_MESH_FLAT_SHADING_OK 1u
/* Every triangle of the mesh passes the same tangent check for ANY hit on it, by a bound instead of an identity: the
 * rejected quantity |dpdu . n| is the rounding residue of a Gram-Schmidt step and stays below 12 eps / sin(theta),
 * theta = angle(dpdu_raw, n); the host proves sin(theta) >= 0.044 for every normal the interpolation can produce
 * (host/flatten.cpp, `smooth_shading_bound`), i.e. |dpdu . n| < 2e-5 against the 1e-3 threshold. */
#define PBRS_MESH_SMOOTH_SHADING_OK 2u
#define PBRS_MESH_SHADING_OK_MASK 3u

/* One triangle's geometry in BLAS leaf order, Q11 swap applied: `let (i,k,j) = index_triple`
 * (shape/src/blas.rs:162) => p0 = positions[i], p1 = positions[j], p2 = positions[k].
 * n = `(p0 - p1).cross(p2 - p1).try_hat()` (shape/src/simple.rs:436), which depends on the triangle only:
 * evaluated once by the host with the reference's operand order; all-NaN when try_hat returns None. */
typedef struct pbrs_tri_verts {
    float p0[3];
    float nx;
    float p1[3];
    float ny;
    float p2[3];
    float nz;
} pbrs_tri_verts;

/* Shading attributes of the same triangle, same vertex order (shape/src/blas.rs:170-185). */
typedef struct pbrs_tri_shade {
    float n0[3], n1[3], n2[3];
    float uv0[2], uv1[2], uv2[2];
    uint32_t orig; /* index of the triangle in the input index buffer */
} pbrs_tri_shade;

/* geometry/src/bxdf.rs:263-269 flattened: `bxdfs_at` is constant per material when its textures are Solid; a lobe whose
 * colour comes from another texture names it in `tex` and the colour is evaluated per hit (material/src/lib.rs). */
enum pbrs_bxdf_kind { PBRS_BXDF_SPECULAR = 0, PBRS_BXDF_DIFFUSE = 1, PBRS_BXDF_MICROFACET = 2, PBRS_BXDF_FOURIER = 3 };
enum pbrs_intrusion { PBRS_REFLECTION = 0, PBRS_TRANSMISSION = 1, PBRS_HYBRID = 2 };        /* bxdf.rs:35-40  */
enum pbrs_fresnel_kind { PBRS_FRESNEL_NOP = 0, PBRS_FRESNEL_DIELECTRIC = 1, PBRS_FRESNEL_CONDUCTOR = 2 }; /* :284-289 */
typedef struct pbrs_bxdf {
    uint32_t kind;
    uint32_t intrusion;   /* Specular; Fourier: index into pbrs_scene_desc::fourier_tables */
    uint32_t fresnel;     /* Specular, Microfacet */
    uint32_t oren_nayar;  /* Diffuse: 0 Lambertian, 1 Oren-Nayar */
    float albedo[3];
    float alpha_x;        /* Beckmann (microfacet.rs:11); already through roughness_to_alpha */
    float eta[3];         /* dielectric: eta_front, eta_back, -; conductor: eta_t rgb (eta_i = 1) */
    float alpha_y;
    float k[3];           /* conductor k rgb; Oren-Nayar: coeff_a, coeff_b, - */
    uint32_t tex;         /* 0: `albedo` is the colour; else (texture index + 1) | PBRS_BXDF_TEX_DROP_IF_BLACK */
} pbrs_bxdf;
/* Uber pushes a textured lobe only when the texture's value at the hit is not black (material/src/lib.rs:326-362) */
#define PBRS_BXDF_TEX_DROP_IF_BLACK 0x80000000u

#define PBRS_MAX_BXDFS 5 /* Uber, material/src/lib.rs:317-365 */
typedef struct pbrs_material {
    float emission[3]; /* Material::emission, material/src/lib.rs:24-26, :294-296 */
    uint32_t n_bxdfs;
    uint32_t first_bxdf;
    uint32_t flags; /* PBRS_MATERIAL_TEXTURED: some lobe has tex != 0 */
    uint32_t vis_class; /* palette entry of material_visualizer for `Material::summary()`, src/directlighting.rs:247-259 */
    /* 1 + index of a pbrs_bxdf record (outside the material's lobe list) whose albedo / tex hold the colour that
     * `Material::scatter` returns where it is a constant of the material — Lambertian albedo (possibly a texture), Mirror
     * albedo, Plastic diffuse, Dielectric transmit (material/src/lib.rs:163-177, :224-228, :427-432, :246-264) — for
     * normal_visualizer; 0 = not provided (PBRS_INTEGRATOR_NORMALS is then refused) */
    uint32_t vis_bxdf;
} pbrs_material;
#define PBRS_MATERIAL_TEXTURED 1u

/* texture/src/lib.rs:35-223 (Solid is folded into the lobes).  PERLIN: rand_vec = tex_floats[data .. data + 768),
 * perm_x|y|z = tex_words[perm .. perm + 768); IMAGE: texels = tex_floats[data .. data + 3 * width * height). */
typedef struct pbrs_texture {
    uint32_t kind; /* enum pbrs_texture_kind (pbrs_scene_spec.h) */
    float odd[3];
    float even[3];
    float freq;
    uint32_t width, height;
    uint32_t data; /* offset into tex_floats */
    uint32_t perm; /* offset into tex_words */
} pbrs_texture;

/* FourierTable (geometry/src/fourier.rs:99-151) after `build`: every array lives in the texture pools.  tex_floats: mu
 * [n_mu], cdf [n_mu^2], a0 [n_mu^2] (the order-0 cache), a [n_coeffs], recip [m_max] (recip[i] = 1 / i); tex_words:
 * a_offset [n_mu^2], m_lookup [n_mu^2] (i32 bit patterns, validated non-negative and in range by the host). */
typedef struct pbrs_fourier_table {
    uint32_t n_mu, n_channels, m_max;
    uint32_t mu, cdf, a0, a, recip; /* offsets into tex_floats */
    uint32_t a_offset, m_lookup;    /* offsets into tex_words  */
    uint32_t n_coeffs;
    uint32_t pad;
} pbrs_fourier_table;

/* light/src/lib.rs:107-111 + light/src/sample_shape.rs:38-43 (world-space shape, area precomputed). */
typedef struct pbrs_area_light {
    float emit[3];
    uint32_t shape_kind;
    float p[9];
    float area;
    float pad[2];
} pbrs_area_light;

/* light/src/lib.rs:29-39 */
typedef struct pbrs_delta_light {
    uint32_t kind;
    float v[3];
    float color[3];
    float world_radius;
} pbrs_delta_light;

typedef struct pbrs_scene_desc {
    uint32_t n_tlas_nodes;
    const pbrs_node* tlas_nodes;
    uint32_t tlas_height; /* BvhNode::height(), tlas/src/bvh.rs:56-61 */
    uint32_t n_instances;
    const pbrs_instance* instances;
    uint32_t n_shapes;
    const pbrs_shape* shapes;
    uint32_t n_meshes;
    const pbrs_mesh* meshes;
    uint32_t n_blas_nodes;
    const pbrs_node* blas_nodes;
    uint32_t n_triangles;
    const pbrs_tri_verts* tri_verts;
    const pbrs_tri_shade* tri_shade;
    uint32_t n_materials;
    const pbrs_material* materials;
    uint32_t n_bxdfs;
    const pbrs_bxdf* bxdfs;
    uint32_t n_area_lights;
    const pbrs_area_light* area_lights;
    uint32_t n_delta_lights;
    const pbrs_delta_light* delta_lights;
    float env_constant[3]; /* EnvLight::Constant, scene/src/lib.rs:12-16 */
    uint32_t env_kind;     /* enum pbrs_env_kind (pbrs_scene_spec.h) */
    uint32_t n_textures;
    const pbrs_texture* textures;
    uint32_t n_tex_floats;
    const float* tex_floats;
    uint32_t n_tex_words;
    const uint32_t* tex_words;
    uint32_t env_texture;  /* PBRS_ENV_IMAGE: index into textures[] */
    float env_scale[3];
    uint32_t n_fourier_tables;
    const pbrs_fourier_table* fourier_tables;
} pbrs_scene_desc;

/* geometry/src/camera.rs:9-17 with the three `orientation * {c,a,b}` products of shoot_ray (:68-70)
 * hoisted: they do not depend on the pixel. */
typedef struct pbrs_camera {
    float center[3];
    uint32_t width;
    float c[3]; /* orientation * c */
    uint32_t height;
    float a[3]; /* orientation * a */
    float pad0;
    float b[3]; /* orientation * b */
    float pad1;
} pbrs_camera;

/* Work counters in the units of SURVEY.md §8(d); filled only when `collect_counters` is set
 * (an instrumented kernel variant runs; the timed variant carries no counters). */
typedef struct pbrs_stats {
    uint64_t samples;       /* camera samples rendered                                  */
    uint64_t closest_rays;  /* BvhNode::intersect equivalents (tlas/src/bvh.rs:77)       */
    uint64_t shadow_rays;   /* BvhNode::occludes equivalents  (tlas/src/bvh.rs:105)      */
    uint64_t shade_events;  /* path vertices shaded                                      */
    uint64_t tlas_nodes, blas_nodes, instances, instance_hits, triangles, tri_shading;
    uint64_t spheres, quads, cuboids, disks;
    uint64_t shadow_tlas_nodes, shadow_blas_nodes, shadow_instances, shadow_triangles, shadow_prims;
    uint64_t invalid_samples; /* camera samples whose radiance has a NaN or infinite component (always filled) */
    /* HIP-event time per stage, summed over launches, in ms, on the context's stream */
    float ms_raygen, ms_extend, ms_shade, ms_shadow, ms_accumulate, ms_total;
    uint32_t launches_extend, launches_shadow, launches_shade, passes;
    /* Which instantiation of the traversal kernels the render's passes launched (always filled): bit 0 analytic shapes, 1 per-candidate
     * shading check, 2 scanned TLAS, 3 several node steps per round (deep BLAS), 4 walks over four-wide nodes, 5 full further node
     * steps (a scene with coordinates outside the guarded range of the division-free box test), 6 scene arrays staged in LDS, 7 an unscanned TLAS staged in LDS,
     * 8 (k_extend) the TLAS extent follows the reference's ray.t_max to the letter, rises included (a ParallelQuad next to a mesh);
     * 0x80000000: the instrumented variant (collect_counters). */
    uint32_t kernel_features_extend, kernel_features_shadow;
    /* Queue sizes per bounce, summed over the passes of the render (filled with the work counters): paths_at_bounce[b] = rays
     * `scene.tlas.intersect` sees at `for bounces in 0..depth` iteration b (src/pathintegrator.rs:14-16), i.e. k_extend's queue;
     * shadow_rays_at_bounce[b] = `scene.tlas.occludes` calls of that iteration's light estimate (k_shadow's queue).
     * Bounces beyond PBRS_STATS_MAX_BOUNCES - 1 are added to the last entry. */
    uint64_t paths_at_bounce[16];
    uint64_t shadow_rays_at_bounce[16];
} pbrs_stats;
#define PBRS_STATS_MAX_BOUNCES 16

typedef struct pbrs_ctx pbrs_ctx;

/* One context per device.  Stands for the per-thread state of the rayon row loop (src/main.rs:219-224). */
int pbrs_create(int device_ordinal, pbrs_ctx** out);
void pbrs_destroy(pbrs_ctx*);
const char* pbrs_last_error(const pbrs_ctx*);
/* Run the pipeline on an existing HIP stream (e.g. torch's current stream); NULL = the context's own, which is
 * created hipStreamNonBlocking: see "Stream ordering" above. */
int pbrs_set_stream(pbrs_ctx*, void* hip_stream);

/* A render of several passes hands every pass's late bounces (near-empty launches that end with the latency of their longest walks) to a
 * second, high-priority stream of the context, where they run beside the next pass's first bounces; the passes' samples still reach the
 * pixel sums in pass order (src/main.rs:205), so the image does not depend on it.  On by default; 0 keeps every pass on the context's
 * stream — what a host wants when it reads the per-stage milliseconds of pbrs_stats as exclusive times (with the overlap a stage's
 * event brackets include the time its kernels share the chip with the other stream's). */
int pbrs_set_pass_overlap(pbrs_ctx*, int enabled);

/* Copies the flattened scene into HBM.  Stands for building `Scene` (scene/src/lib.rs:36-63). */
int pbrs_upload_scene(pbrs_ctx*, const pbrs_scene_desc*);

typedef struct pbrs_render_params {
    uint32_t x0, y0, w, h;         /* tile, in pixels of the camera film                           */
    uint32_t strata_x, strata_y;   /* spp = strata_x * strata_y; the reference has both = msaa    */
    uint32_t max_depth;            /* `for bounces in 0..depth`, src/pathintegrator.rs:14          */
    uint32_t samples_per_pass;     /* sample indices traced concurrently per pixel (0 = auto)      */
    uint64_t seed;                 /* RNG contract, include/pbrs_numeric.h                          */
    uint32_t collect_counters;     /* run the instrumented kernels and fill the work counters      */
    uint32_t time_stages;          /* bracket every launch with HIP events and fill ms_*           */
    /* Interleaved row bands, the multi-GPU partition of the rayon row loop (src/main.rs:219-224):
     * with band_count > 1 the tile's h rows are the rows of bands band_index, band_index+band_count, ...
     * (band_rows rows each) counted from y0, packed:  film_row = y0 + ((r / band_rows) * band_count
     * + band_index) * band_rows + r % band_rows.  band_count <= 1 means a plain rectangular tile. */
    uint32_t band_rows, band_count, band_index;
    uint32_t integrator;           /* PBRS_INTEGRATOR_*: which `fn(&Scene, Ray, i32) -> Color` of src/main.rs:160-163 */
} pbrs_render_params;
/* src/pathintegrator.rs:9-74 */
#define PBRS_INTEGRATOR_PATH 0u
/* direct_lighting_integrator, src/directlighting.rs:14-47: emission, or the one-light estimate plus one level of perfect
 * specular reflection/refraction (src/bsdf.rs:104-113) followed by direct_lighting_debug_integrator (:49-56).  max_depth
 * only gates it (`depth <= 0` returns black); the chain is at most two rays long. */
#define PBRS_INTEGRATOR_DIRECT 1u
/* material_visualizer, src/directlighting.rs:234-271 (`--visualize-materials`, src/main.rs:166-187): one un-jittered ray per
 * pixel (`shoot_ray(row, col, (0.0, 0.0))`), a palette colour per kind of material at the first hit, a grey checker of
 * the ray direction where nothing is hit.  strata must be 1 x 1; max_depth is ignored (the reference passes 0). */
#define PBRS_INTEGRATOR_MATERIALS 2u
/* normal_visualizer, src/directlighting.rs:273-289 (`--visualize-normals`): one un-jittered ray per pixel,
 * (albedo of `mtl.scatter(-ray.dir, &hit)` + hit.normal) * 0.5, the environment where nothing is hit.  `scatter` is
 * `todo!()` for Glossy, Uber, Substrate and Fourier: their albedo counts as black.  Dielectric::scatter draws one random
 * number: the pixel's RNG stream supplies it (first draw after the two of the unused jitter).  strata must be 1 x 1. */
#define PBRS_INTEGRATOR_NORMALS 3u

/* Renders a tile; replaces src/main.rs:192-231 for the rows/cols of the tile.  `rgb_out` is
 * w*h*3 floats, row-major.  _host writes to caller-owned host memory (one D2H copy at the end);
 * _device leaves the result in caller-owned device memory and does not synchronise the stream. */
int pbrs_render_tile(pbrs_ctx*, const pbrs_camera*, const pbrs_render_params*, float* rgb_out_host, pbrs_stats* stats_out);
int pbrs_render_tile_device(pbrs_ctx*, const pbrs_camera*, const pbrs_render_params*, float* rgb_out_device, pbrs_stats* stats_out);
/* After a _device render with time_stages/collect_counters: waits for the stream and fills the stats. */
int pbrs_collect_stats(pbrs_ctx*, pbrs_stats* stats_out);

/* ---- parity-harness entry points (the reference's own functions, batched) -------------------------- */
typedef struct pbrs_hit_record {
    float t;
    uint32_t inst; /* 0xffffffff = miss */
    uint32_t prim;
    float b1, b2;
} pbrs_hit_record;
/* `scene.tlas.intersect(&mut ray)` (tlas/src/bvh.rs:77-103) / `scene.tlas.occludes(&ray)` (:105-113)
 * for n caller-supplied rays (host pointers; origins/dirs are n*3 floats). Either output may be NULL. */
int pbrs_intersect_rays(pbrs_ctx*, uint32_t n, const float* origins, const float* dirs, const float* tmax,
                        pbrs_hit_record* hits_out, uint8_t* occluded_out);
/* Which walks the context's last pbrs_intersect_rays call went through: every query takes the walk its stage runs in the pipeline
 * for the uploaded scene (occlusion: the four-wide any-hit walk of k_shadow where the TLAS is scanned and the BLASes are deep
 * enough, with its hand-off of rays outside the guarded range to the binary walk; closest hit: the binary walk).  slow_*: rays the
 * wide walk handed to the binary walk (0 when the stage runs the binary walk anyway). */
typedef struct pbrs_intersect_info {
    uint32_t wide_any, wide_closest;
    uint32_t slow_any, slow_closest;
} pbrs_intersect_info;
int pbrs_last_intersect_info(const pbrs_ctx*, pbrs_intersect_info* out);
/* Camera rays of one sample index for a tile (src/main.rs:197-203, geometry/src/camera.rs:65-77). */
int pbrs_camera_rays(pbrs_ctx*, const pbrs_camera*, const pbrs_render_params*, uint32_t sample_index, float* origins_out,
                     float* dirs_out);
/* include/pbrs_numeric.h evaluated on the device (fn ids as oracle_numeric_eval). */
int pbrs_numeric_eval(pbrs_ctx*, uint32_t fn, uint32_t n, const float* x, const float* y, float* out);
/* Per-sample radiance of one sample index for a tile (before the sum over samples), for bisecting. */
int pbrs_render_sample_radiance(pbrs_ctx*, const pbrs_camera*, const pbrs_render_params*, uint32_t sample_index, float* rgb_out_host);

#ifdef __cplusplus
}
#endif
#endif
