// oracle/ref_scene.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the reference's scene vocabulary on the path-integrator hot path:
//   geometry/src/interaction.rs, geometry/src/transform.rs (AffineTransform::apply),
//   shape/src/simple.rs, shape/src/blas.rs, tlas/src/{bvh,instance}.rs,
//   geometry/src/{bxdf,microfacet}.rs, material/src/lib.rs, light/src/{lib,sample_shape}.rs,
//   geometry/src/camera.rs, scene/src/lib.rs.
// Pointer-linked trees, recursion and virtual-free enum dispatch as in the Rust source; nothing
// here is shared with the HIP kernels except include/pbrs_numeric.h (the f32 libm contract).
#pragma once
#include <functional>
#include <memory>
#include <vector>

#include "../include/pbrs_scene_spec.h"
#include "ref_math.h"

namespace ref {

// Work counters in the units of SURVEY.md §8(d)'s algorithmic-bytes table.
struct Counters {
    uint64_t closest_rays = 0;   // BvhNode::intersect calls at the root (tlas/src/bvh.rs:77)
    uint64_t shadow_rays = 0;    // BvhNode::occludes calls at the root (tlas/src/bvh.rs:105)
    uint64_t tlas_nodes = 0;     // TLAS nodes whose bbox was tested
    uint64_t blas_nodes = 0;     // BLAS nodes whose bbox was tested
    uint64_t instances = 0;      // Instance::intersect / occludes entered
    uint64_t instance_hits = 0;  // Instance::intersect returned Some
    uint64_t triangles = 0;      // triangle tests (mesh or isolated)
    uint64_t spheres = 0, quads = 0, cuboids = 0, disks = 0;
    uint64_t tri_shading = 0;  // TriangleMesh::intersect_triangle past the geometric test
    uint64_t shade_events = 0; // path vertices shaded (bounce-loop iterations with a hit)
    uint64_t samples = 0;      // camera samples
    void add(const Counters& o) {
        closest_rays += o.closest_rays; shadow_rays += o.shadow_rays; tlas_nodes += o.tlas_nodes;
        blas_nodes += o.blas_nodes; instances += o.instances; instance_hits += o.instance_hits;
        triangles += o.triangles; spheres += o.spheres; quads += o.quads; cuboids += o.cuboids;
        disks += o.disks; tri_shading += o.tri_shading; shade_events += o.shade_events; samples += o.samples;
    }
};
extern thread_local Counters* g_cnt;
#define REF_COUNT(field)                \
    do {                                \
        if (::ref::g_cnt) ::ref::g_cnt->field++; \
    } while (0)

// ---- geometry/src/interaction.rs:12-70 --------------------------------------------------------
struct Interaction {
    Point3 pos;
    float ray_t;
    float u, v;
    Vec3 normal;
    Vec3 wo;
    Mat3 tbn;
    // Not in the reference: identifies the primitive for hit-record parity tests.
    uint32_t prim = 0;
    float b1 = 0.0f, b2 = 0.0f;
};
Interaction isect_new(Point3 pos, float ray_t, float u, float v, Vec3 normal, Vec3 wo);  // :23-34
Interaction isect_rayless(Point3 pos, float u, float v, Vec3 normal);                    // :37-39
Interaction with_dpdu(Interaction self, Vec3 dpdu);                                      // :45-61
inline Vec3 tangent(const Interaction& i) { return i.tbn.cols[0]; }                      // :41-43
Ray spawn_ray(const Interaction& i, Vec3 dir);                                           // :63-66
Ray spawn_limited_ray_to(const Interaction& i, Point3 pos);                              // :68-70
bool has_valid_frame(const Interaction& i);                                              // :72-87

// ---- shapes (shape/src/simple.rs, shape/src/blas.rs) --------------------------------------------
struct Sphere { Point3 center; float radius; };
struct Disk { Point3 center; Vec3 normal; Vec3 radial; };
struct ParallelQuad { Point3 origin; Vec3 side_u, side_v; };
struct Cuboid { Point3 min, max; };
struct IsolatedTriangle { Point3 p0, p1, p2; };

bool intersect_triangle(Point3 p0, Point3 p1, Point3 p2, const Ray& r, Interaction* out);  // simple.rs:435-475
bool intersect_triangle_pred(Point3 p0, Point3 p1, Point3 p2, const Ray& r);               // simple.rs:477-495

struct IsoBvhNode {  // blas.rs:10-18
    BBox bbox;
    bool is_leaf;
    std::unique_ptr<IsoBvhNode> child[2];
    int axis = 0;
    size_t start = 0, end = 0;
    size_t height() const;  // :21-26
    size_t count() const;   // :27-32
};
struct MeshTriangle {  // blas.rs:72-76
    uint32_t i, j, k;  // index_triple as given to from_soa
    BBox bbox;
    uint32_t orig;  // not in the reference: position in the input index buffer
};
struct TriangleMesh {  // blas.rs:87-95
    std::vector<Point3> positions;
    std::vector<Vec3> normals;
    std::vector<float> uvs;  // 2 per vertex
    std::vector<MeshTriangle> triangles;
    std::unique_ptr<IsoBvhNode> bvh_root;
    bool intersect_one(const MeshTriangle& tri, const Ray& r, Interaction* out) const;  // :161-207
    bool intersect_one_pred(const MeshTriangle& tri, const Ray& r) const;               // :208-211
};
std::unique_ptr<TriangleMesh> mesh_from_soa(const pbrs_mesh_spec& m);  // blas.rs:134-159

struct Shape {
    uint32_t kind;
    Sphere sphere;
    Disk disk;
    ParallelQuad quad;
    Cuboid cuboid;
    IsolatedTriangle tri;
    std::shared_ptr<TriangleMesh> mesh;
    BBox bbox() const;
    bool intersect(const Ray& r, Interaction* out) const;
    bool occludes(const Ray& r) const;
};
Shape shape_from_spec(const pbrs_shape_spec& s, const std::vector<std::shared_ptr<TriangleMesh>>& meshes);

// ---- BxDFs (geometry/src/bxdf.rs, geometry/src/microfacet.rs) ---------------------------------------
using Omega = Vec3;  // bxdf.rs:29 `struct Omega(pub Vec3)`
struct Fresnel {     // bxdf.rs:284-289
    enum Kind { Nop, Dielectric, Conductor } kind;
    float eta_front, eta_back;
    Color eta_i, eta_t, k;
    float refl_coeff(float cos_theta_i) const;  // :308-342
    Color eval(float cos_theta_i) const;        // :344-392
};
struct MicrofacetDistrib {  // microfacet.rs:10-13
    enum Kind { Beckmann, TrowbridgeReitz } kind;
    float alpha_x, alpha_y;
    float d(Omega wh) const;                            // :36-60
    float lambda(Omega w) const;                        // :65-88
    float g1(Omega w) const;                            // :99-101
    float g(Omega wo, Omega wi) const;                  // :106-108
    float pdf(Omega wo, Omega wh) const;                // :110-122
    Omega sample_wh(Omega wo, float u, float v) const;  // :124-159
};
float roughness_to_alpha(float roughness);  // microfacet.rs:16-23

// ---- geometry/src/fourier.rs:99-221 + math/src/spline.rs:187-335 --------------------------------
struct FourierTable {
    size_t m_max = 0, n_channels = 0;
    std::vector<float> mu, cdf, a0;
    std::vector<int32_t> a_offset, m_lookup;
    std::vector<float> a, recip;
    // FourierTable::build(n_channels, mu, cdf, a_offset, m_lookup, coefficients), :115-151
    static std::shared_ptr<FourierTable> build(const pbrs_fourier_table_spec& t);
    const float* get_ak(size_t offset_i, size_t offset_o, size_t* m) const;  // :160-165
};
size_t find_interval(size_t size, const std::function<bool(size_t)>& predicate);                  // spline.rs:161-185 (range.start)
bool catmull_rom_weights(const std::vector<float>& nodes, float x, long* offset, float w[4]);     // spline.rs:203-247
bool sample_catmull_rom_2d(const std::vector<float>& nodes_v, const std::vector<float>& nodes_h, const std::vector<float>& values,
                           const std::vector<float>& cdf, float alpha, float u, float* fval, float* x, float* pdf);  // :249-318
float fourier_sum(const float* a, size_t n, float cos_phi);                                       // fourier.rs:224-237
void sample_fourier(const float* ak, size_t n, const float* recip, float u, float* f, float* phi, float* pdf);  // :245-297

struct BXDF {  // bxdf.rs:263-269
    enum Kind { Specular, DiffuseReflect, MicrofacetReflect, FresnelBlend, Fourier } kind;
    const FourierTable* table = nullptr;  // Fourier (fourier.rs:219-221)
    // Specular (:395-399)
    Fresnel fresnel;
    Color albedo;
    enum Intrusion { Reflection, Transmission, Hybrid } intrusion;
    // DiffuseReflect (:514-518)
    bool oren_nayar = false;
    float coeff_a = 0.0f, coeff_b = 0.0f;
    // MicrofacetReflection (:577-581) / FresnelBlend (:648-652)
    MicrofacetDistrib distrib;
    Color diffuse, specular;

    Color eval(Omega wo, Omega wi) const;
    void sample(Omega wo, float r0, float r1, Color* f, Omega* wi, Prob* pr) const;
    Prob prob(Omega wo, Omega wi) const;
};
BXDF bxdf_mirror(Color albedo);                                        // :401-407
BXDF bxdf_dielectric(Color albedo, float eta_outer, float eta_inner);  // :411-417
BXDF bxdf_transmit(Color albedo, float eta_outer, float eta_inner);    // :419-425
BXDF bxdf_lambertian(Color albedo);                                    // :521-526
BXDF bxdf_oren_nayar(Color albedo, float sigma_rad);                   // :528-536
BXDF bxdf_microfacet(Color albedo, MicrofacetDistrib d, Fresnel f);    // :584-590
BXDF bxdf_fresnel_blend(Color diffuse, Color specular, MicrofacetDistrib d);  // :656-662
Fresnel fresnel_nop();
Fresnel fresnel_dielectric(float eta_front, float eta_back);  // :292-297
Fresnel fresnel_conductor(Color eta_real, Color eta_imag);    // :299-305
void concentric_sample_disk(float u, float v, float* x, float* y);  // :187-200
Omega cos_sample_hemisphere(float u, float v);                      // :202-206

// ---- src/bsdf.rs ----------------------------------------------------------------------------
struct BSDF {
    Mat3 frame;
    const std::vector<BXDF>* bxdfs;
    Color eval(Vec3 wo, Vec3 wi) const;                                              // :43-51
    float pdf(Vec3 wo, Vec3 wi) const;                                               // :53-57
    void sample(Vec3 wo_world, float u, float v, Color* f, Vec3* wi, Prob* pr) const;  // :59-103
    bool sample_specular(Vec3 wo_world, Color* f, Vec3* wi, Prob* pr) const;           // :104-113
    Omega world_to_local(Vec3 w) const;                                              // :114-118
    Vec3 local_to_world(Omega l) const;                                              // :120-124
};
BSDF bsdf_new_frame(const Interaction& isect);  // :18-31

// ---- texture/src/lib.rs ------------------------------------------------------------------------
struct Texture {
    uint32_t kind = 0;  // pbrs_texture_kind
    Color odd, even;                          // Checker :35-49
    float freq = 1.0f;                        // Perlin :51-171
    std::vector<Vec3> rand_vec;               // 256
    std::vector<uint32_t> perm_x, perm_y, perm_z;
    std::vector<Color> data;                  // Image :173-223
    uint32_t width = 0, height = 0;
    Color value(float u, float v, Point3 p) const;
    float noise(Point3 p) const;       // :97-137
    float turbulance(Point3 p) const;  // :139-147
};

// ---- material/src/lib.rs -----------------------------------------------------------------------
struct Material {
    pbrs_material_spec spec;
    std::shared_ptr<Texture> tex[4];  // nullptr = the Solid colour in spec.p
    std::shared_ptr<FourierTable> fourier;  // Fourier (:451-475)
    std::vector<BXDF> bxdfs_at(const Interaction& isect) const;  // per-kind `bxdfs_at`
    Color emission() const;                                      // :24-26, :294-296
};

// ---- light/src ----------------------------------------------------------------------------------------
struct SamplableShape {  // sample_shape.rs:38-43
    uint32_t kind;
    Sphere sphere;
    Disk disk;
    IsolatedTriangle tri;
    ParallelQuad quad;
    bool intersect(const Ray& r, Interaction* out) const;
    Interaction sample(float u, float v) const;
    Interaction sample_towards(const Interaction& target, float u, float v) const;
    bool pdf_at(const Interaction& reference, Vec3 wi, float* pdf) const;
    float area() const;
};
struct DiffuseAreaLight {  // lib.rs:107-111
    Color emit_radiance;
    SamplableShape shape;
    float area;
    Color radiance_from(const Interaction& from, Vec3 wo) const;                                 // :127-133
    bool radiance_to(const Interaction& target, Vec3 wi, Color* le, float* pdf, Ray* vis) const;  // :141-146
    void sample_incident_radiance(const Interaction& target, float u, float v, Color* li, Vec3* wi, Prob* pr,
                                  Ray* vis) const;  // :158-172
};
struct DeltaLight {  // lib.rs:29-39
    uint32_t kind;
    Vec3 v;
    Color color;
    float world_radius;
    void sample_incident_radiance(const Interaction& target, Color* li, Vec3* wi, Prob* pr, Ray* vis) const;  // :67-92
};

// ---- tlas/src ----------------------------------------------------------------------------------------------
struct Instance {  // instance.rs:12-16
    std::shared_ptr<Shape> shape;
    std::shared_ptr<Material> mtl;
    Mat4 forward, inverse;
    uint32_t index;  // not in the reference: position in the input instance list
    BBox bbox() const;                                           // :47-49
    bool intersect(const Ray& ray, Interaction* out) const;      // :50-67
    bool occludes(const Ray& ray) const;                         // :68-72
};
struct Hit {
    Interaction isect;
    const Instance* inst;
};
struct BvhNode {  // bvh.rs:11-18
    BBox bbox;
    std::unique_ptr<BvhNode> child[2];
    std::unique_ptr<Instance> leaf;
    bool intersect(Ray& ray, Hit* out) const;  // :77-103
    bool occludes(const Ray& ray) const;       // :105-113
    uint32_t height() const;                   // :56-61
};
std::unique_ptr<BvhNode> build_bvh(std::vector<std::unique_ptr<Instance>> instances);  // :116-152

// ---- geometry/src/camera.rs -------------------------------------------------------------------------
struct Camera {
    Point3 center;
    Vec3 a, b, c;
    uint32_t width, height;
    Mat3 orientation;
    bool shoot_ray(uint32_t row, uint32_t col, float dx, float dy, Ray* out) const;  // :65-77
};
Camera camera_from_spec(const pbrs_camera_spec& s);  // :19-44

// ---- scene/src/lib.rs:19-33 -------------------------------------------------------------------------
struct Scene {
    std::unique_ptr<BvhNode> tlas;
    std::vector<DeltaLight> delta_lights;
    std::vector<DiffuseAreaLight> area_lights;
    Color env_constant;
    uint32_t env_kind = 0;             // pbrs_env_kind: Constant, Image, or one of the Fn closures of preset.rs:25-53
    std::shared_ptr<Texture> env_map;  // Image
    Color env_scale;
    Camera camera;
    bool has_env_light() const { return env_kind != PBRS_ENV_CONSTANT || !is_black(env_constant); }  // :96-103
    Color eval_env_light(const Ray& ray) const;                                                      // :105-117
};
std::unique_ptr<Scene> scene_from_spec(const pbrs_scene_spec& spec);

// ---- radiometry/src/spectrum.rs:3-70, math/src/spline.rs:11-158 (ref_spectrum.cpp): load-time colour conversions ----------
std::vector<float> blackbody(float kelvin, const std::vector<float>& lambdas_nm);
std::vector<float> blackbody_normalized(float kelvin, const std::vector<float>& lambdas_nm);
Color temperature_to_color(float kelvin);
bool tridiagonal(const std::vector<float>& a, const std::vector<float>& b, const std::vector<float>& c, const std::vector<float>& rhs, std::vector<float>* x_out);
bool cubic_spline_zero_hess(const std::vector<std::pair<float, float>>& xs_and_ys, std::vector<float>* m_out);
struct CubicSpline {
    std::vector<float> m, xs, ys;
    static bool from_samples(const std::vector<std::pair<float, float>>& xs_and_ys, CubicSpline* out);
    float evaluate(float at) const;
};
Color sampled_spectrum_to_color(std::vector<std::pair<float, float>> lambdas_and_values);

}  // namespace ref
