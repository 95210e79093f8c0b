/* pbrs_host.h — host side of the pbrs_gpu boundary: scene vocabulary -> flattened SoA/AoS buffers.
 *
 * In the reference this work is Rust host code that "stays" (BASELINE.json north_star): building the
 * TLAS over instances (tlas/src/bvh.rs:116-152), one BLAS per TriangleMesh (shape/src/blas.rs:134-159,
 * :333-420), evaluating `Material::bxdfs_at` for constant textures (material/src/lib.rs:162-449),
 * `DiffuseAreaLight::new` (light/src/lib.rs:114-121) and `Camera::new(..).look_at(..)`
 * (geometry/src/camera.rs:19-44).  No Rust toolchain exists in this image, so the same host logic is
 * written in C++ (pbrs_amd/csrc/host/) behind this C API; its output is exactly the
 * `pbrs_scene_desc` + `pbrs_camera` that include/pbrs_gpu.h consumes, so a Rust host could replace
 * this library without touching the device side.
 */
#ifndef PBRS_HOST_H
#define PBRS_HOST_H

#include "pbrs_gpu.h"
#include "pbrs_scene_spec.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pbrs_host_scene pbrs_host_scene;

/* Scene::new(*tlas::build_bvh(instances), camera).with_lights(delta, area) — scene/src/lib.rs:36-63,:118-126 */
int pbrs_host_scene_build(const pbrs_scene_spec* spec, pbrs_host_scene** out);
void pbrs_host_scene_free(pbrs_host_scene*);
/* Borrowed views, valid until pbrs_host_scene_free. */
const pbrs_scene_desc* pbrs_host_scene_desc(const pbrs_host_scene*);
const pbrs_camera* pbrs_host_scene_camera(const pbrs_host_scene*);
/* Deepest traversal stack the device needs: tlas height + max BLAS height (tlas/src/bvh.rs:56-61,
 * shape/src/blas.rs:21-26). */
uint32_t pbrs_host_scene_stack_depth(const pbrs_host_scene*);
const char* pbrs_host_last_error(void);

/* pbrt-v3 front-end (scene_parser/src/, scene/src/loader.rs:41-879, scene/src/plyloader.rs, PNG image maps): reads the
 * subset of the format the reference supports and produces the plain scene spec that pbrs_host_scene_build consumes.
 * Constructs the reference leaves unimplemented (object instancing, spectral colours, Loop subdivision, ...) are errors here, reported through pbrs_host_load_error, never aborts. */
typedef struct pbrs_loaded_scene pbrs_loaded_scene;
int pbrs_host_load_pbrt(const char* path, pbrs_loaded_scene** out);
const pbrs_scene_spec* pbrs_loaded_scene_spec(const pbrs_loaded_scene*); /* borrowed, valid until _free */
void pbrs_loaded_scene_free(pbrs_loaded_scene*);
const char* pbrs_host_load_error(void);

/* Image output of the reference's front end (src/main.rs:28-53): `write_exr` — f32 RGB, the file a render ends in (:245) —
 * and `write_image` — 8-bit PNG of `gamma_encode().to_u8()` pixels (radiometry/src/color.rs:13-23, :54-66).  `rgb` is
 * row-major, 3 floats per pixel, as pbrs_render_tile returns it. */
int pbrs_host_write_exr(const char* path, const float* rgb, uint32_t width, uint32_t height);
int pbrs_host_write_png(const char* path, const float* rgb, uint32_t width, uint32_t height);
const char* pbrs_host_io_error(void);

#ifdef __cplusplus
}
#endif
#endif
