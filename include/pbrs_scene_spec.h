/* pbrs_scene_spec.h — the un-flattened scene vocabulary handed to a host-side scene builder.
 *
 * This is the plain-data equivalent of what the reference's scene/src/preset.rs or
 * scene/src/loader.rs hand to `tlas::build_bvh` + `Scene::new(..).with_lights(..)`:
 * a list of `Instance { shape, mtl, transform }` (tlas/src/instance.rs:12-16), the area / delta
 * light lists (scene/src/lib.rs:28-31), a constant environment colour (scene/src/lib.rs:12-16,
 * `EnvLight::Constant`) and the camera parameters of `Camera::new(..).look_at(..)`
 * (geometry/src/camera.rs:19-44).  It carries no acceleration structure: each consumer (the
 * product's host flattener in pbrs_amd/csrc/host, and independently the CPU oracle in oracle/)
 * runs its own restatement of tlas/src/bvh.rs:116-152 and shape/src/blas.rs:333-420 over it.
 *
 * All structs are POD, little-endian, pointers are borrowed for the duration of the call.
 */
#ifndef PBRS_SCENE_SPEC_H
#define PBRS_SCENE_SPEC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* shape/src/simple.rs + shape/src/blas.rs */
enum pbrs_shape_kind {
    PBRS_SHAPE_SPHERE = 0,   /* p = center.xyz, radius                       (simple.rs:10-13)   */
    PBRS_SHAPE_QUAD = 1,     /* p = origin.xyz, side_u.xyz, side_v.xyz       (simple.rs:69-73)   */
    PBRS_SHAPE_CUBOID = 2,   /* p = min.xyz, max.xyz                         (simple.rs:167-170) */
    PBRS_SHAPE_DISK = 3,     /* p = center.xyz, normal.xyz (unit), radial.xyz (simple.rs:34-39)  */
    PBRS_SHAPE_TRIANGLE = 4, /* p = p0.xyz, p1.xyz, p2.xyz                   (simple.rs:185-189) */
    PBRS_SHAPE_MESH = 5      /* mesh = index into meshes[]                   (blas.rs:87-95)     */
};

typedef struct pbrs_shape_spec {
    uint32_t kind;
    uint32_t mesh;
    float p[9];
} pbrs_shape_spec;

/* TriangleMesh::from_soa(positions, normals, uvs, indices) — shape/src/blas.rs:134-159 */
typedef struct pbrs_mesh_spec {
    uint32_t n_vertices;
    uint32_t n_triangles;
    const float* positions;  /* n_vertices * 3 */
    const float* normals;    /* n_vertices * 3 */
    const float* uvs;        /* n_vertices * 2 */
    const uint32_t* indices; /* n_triangles * 3, as given to from_soa (NOT yet (i,k,j)-swapped) */
} pbrs_mesh_spec;

/* texture/src/lib.rs.  `Solid` (:19-33) stays inline in the material parameters; the others are referenced by
 * pbrs_material_spec.tex.  Perlin's tables come from the unseeded `rand::random` in the reference (:66-95): here they are
 * part of the scene (any 256 unit vectors + three permutations of 0..255), so both sides evaluate the same noise. */
enum pbrs_texture_kind {
    PBRS_TEX_CHECKER = 1, /* odd / even by the sign of sin(10x) sin(10y) sin(10z)                 (:35-49)   */
    PBRS_TEX_PERLIN = 2,  /* marble: sin(freq * z + 10 * turbulence(p)) * 0.5 + 0.5, grey          (:51-171)  */
    PBRS_TEX_IMAGE = 3    /* nearest texel of a width x height RGB image, uv clamped to [0, 1]     (:173-223) */
};
typedef struct pbrs_texture_spec {
    uint32_t kind;
    float odd[3], even[3];    /* CHECKER */
    float freq;               /* PERLIN */
    uint32_t width, height;   /* IMAGE */
    const float* data;        /* PERLIN: rand_vec, 256 * 3;  IMAGE: width * height * 3 (row-major, already in [0, 1]) */
    const uint32_t* perm;     /* PERLIN: perm_x, perm_y, perm_z, 3 * 256 */
} pbrs_texture_spec;

/* material/src/lib.rs — colours are `Solid` unless pbrs_material_spec.tex names a texture. */
enum pbrs_material_kind {
    PBRS_MTL_LAMBERTIAN = 0,    /* p[0..3) albedo                                            (:31-44)  */
    PBRS_MTL_METAL = 1,         /* p[0..3) eta, p[3..6) k, p[6] fuzziness                    (:45-65)  */
    PBRS_MTL_GLOSSY = 2,        /* p[0..3) albedo, p[3] roughness                            (:67-79)  */
    PBRS_MTL_MIRROR = 3,        /* p[0..3) albedo                                            (:81-88)  */
    PBRS_MTL_PLASTIC = 4,       /* p[0..3) diffuse, p[3..6) specular, p[6] roughness; flag 1 (:90-95)  */
    PBRS_MTL_DIELECTRIC = 5,    /* p[0] refract_index, p[1..4) reflect, p[4..7) transmit     (:97-117) */
    PBRS_MTL_DIFFUSE_LIGHT = 6, /* p[0..3) emit                                              (:124-132)*/
    PBRS_MTL_UBER = 7,          /* p[0..3) kd, [3..6) ks, [6..9) kr, [9..12) kt, [12] rough_u,
                                   [13] rough_v, [14] eta, [15] opacity; flags 1,2,4         (:302-369)*/
    PBRS_MTL_SUBSTRATE = 8,     /* p[0..3) kd, p[3..6) ks                                    (:371-424)*/
    PBRS_MTL_FOURIER = 9        /* tex[0] = index into fourier_tables[] (NOT a texture)      (:451-475)*/
};
#define PBRS_MTL_FLAG_REMAP_ROUGHNESS 1u
#define PBRS_MTL_FLAG_HAS_KR 2u
#define PBRS_MTL_FLAG_HAS_KT 4u

/* geometry/src/fourier.rs:99-221 — the arrays of a `.bsdf` file (the SCATFUN format of layerlab, :13-51) as
 * FourierTable::from_file reads them after the 64-byte header; `FourierTable::build` (:115-151) derives m_max, the
 * order-0 cache and the reciprocal table from them on each side of the boundary. */
typedef struct pbrs_fourier_table_spec {
    uint32_t n_mu;       /* header.n_mu: elevational samples (>= 3)                               */
    uint32_t n_channels; /* 1 (monochromatic) or 3 (luminance, red, blue)                        */
    uint32_t n_coeffs;   /* header.n_coeffs                                                      */
    float eta;           /* header.eta (read, asserted finite and never used by the reference)   */
    const float* mu;                  /* n_mu zenith cosines, ascending                          */
    const float* cdf;                 /* n_mu * n_mu                                             */
    const int32_t* offset_and_length; /* n_mu * n_mu pairs: offset into a[], series length m     */
    const float* a;                   /* n_coeffs: per (mu_o, mu_i) pair n_channels * m values   */
} pbrs_fourier_table_spec;

typedef struct pbrs_material_spec {
    uint32_t kind;
    uint32_t flags;
    float p[16];
    /* 0 = the colour in p[]; t + 1 = textures[t].  LAMBERTIAN: tex[0] albedo (material/src/lib.rs:32-44);
     * UBER: tex[0..4) = kd, ks, kr, kt (:303-306).  Other kinds take no textures in the reference. */
    uint32_t tex[4];
} pbrs_material_spec;

/* tlas/src/instance.rs:12-16; transform = AffineTransform{forward, inverse} as two column-major
 * Mat4 (math/src/hcm.rs:477-479, geometry/src/transform.rs:16-19): m[4*col + row]. */
typedef struct pbrs_instance_spec {
    uint32_t shape;
    uint32_t material;
    float forward[16];
    float inverse[16];
} pbrs_instance_spec;

/* light/src/lib.rs:107-121 — shape is one of SPHERE / DISK / TRIANGLE / QUAD, in world space
 * (light/src/sample_shape.rs:38-43). */
typedef struct pbrs_area_light_spec {
    float emit[3];
    pbrs_shape_spec shape;
} pbrs_area_light_spec;

/* light/src/lib.rs:29-39 */
enum pbrs_delta_light_kind { PBRS_DELTA_POINT = 0, PBRS_DELTA_DISTANT = 1 };
typedef struct pbrs_delta_light_spec {
    uint32_t kind;
    float v[3];     /* POINT: position; DISTANT: casting_dir */
    float color[3]; /* POINT: intensity; DISTANT: radiance   */
    float world_radius;
} pbrs_delta_light_spec;

/* Camera::new((w,h), fov_y) + look_at(from, target, up) — geometry/src/camera.rs:19-44 */
typedef struct pbrs_camera_spec {
    uint32_t width, height;
    float fov_y_rad;
    float from[3], target[3], up[3];
} pbrs_camera_spec;

typedef struct pbrs_scene_spec {
    uint32_t n_meshes;
    const pbrs_mesh_spec* meshes;
    uint32_t n_shapes;
    const pbrs_shape_spec* shapes;
    uint32_t n_materials;
    const pbrs_material_spec* materials;
    uint32_t n_instances;
    const pbrs_instance_spec* instances;
    uint32_t n_area_lights;
    const pbrs_area_light_spec* area_lights;
    uint32_t n_delta_lights;
    const pbrs_delta_light_spec* delta_lights;
    float env_constant[3]; /* EnvLight::Constant (black = no env light, scene/src/lib.rs:96-102) */
    pbrs_camera_spec camera;
    uint32_t n_textures;
    const pbrs_texture_spec* textures;
    uint32_t env_kind;     /* enum pbrs_env_kind; CONSTANT reads env_constant */
    uint32_t env_texture;  /* IMAGE: index into textures[] (an IMAGE texture), looked up at lat-long (u, v) (:108-114) */
    float env_scale[3];    /* IMAGE: `scale_factor` */
    uint32_t n_fourier_tables;
    const pbrs_fourier_table_spec* fourier_tables; /* material::Fourier::from_file, one per `.bsdf` file */
} pbrs_scene_spec;
/* scene/src/lib.rs:20-24 EnvLight; the `Fn` arm carries one of the closures of scene/src/preset.rs:25-53 */
enum pbrs_env_kind { PBRS_ENV_CONSTANT = 0, PBRS_ENV_IMAGE = 1, PBRS_ENV_BLUE_SKY = 2, PBRS_ENV_DARK_ROOM = 3, PBRS_ENV_DUSK = 4 };

#ifdef __cplusplus
}
#endif
#endif
