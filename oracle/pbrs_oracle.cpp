// oracle/pbrs_oracle.cpp — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the reference's per-pixel Monte-Carlo path integrator:
//   src/main.rs:192-231 (pixel/sample loop, row parallelism), src/pathintegrator.rs:9-74,
//   src/directlighting.rs:58-232, scene/src/lib.rs.
// plus the C entry points tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg bind to.
// It is the checker for the HIP path, never the thing shipped: nothing under pbrs_amd/ links or
// loads this library.
//
// PARITY PINNING: the Rust reference cannot be built here (no cargo/rustc, un-vendored crates,
// SURVEY.md §8c).  The lower layers are pinned by transcribing every known-answer test the
// reference holds for them (oracle/selftest.cpp); at integrator level (radiance per pixel, BVH
// traversal, NEE) the reference has no tests and no golden images, so that level is
// "parity unpinned": it rests on this restatement following the cited lines.
#include <atomic>
#include <cstring>
#include <thread>

#include "oracle_api.h"
#include "ref_scene.h"

namespace ref {

std::unique_ptr<Scene> scene_from_spec(const pbrs_scene_spec& spec) {
    auto scene = std::make_unique<Scene>();
    std::vector<std::shared_ptr<TriangleMesh>> meshes;
    for (uint32_t m = 0; m < spec.n_meshes; ++m) meshes.push_back(std::shared_ptr<TriangleMesh>(mesh_from_soa(spec.meshes[m])));
    std::vector<std::shared_ptr<Shape>> shapes;
    for (uint32_t s = 0; s < spec.n_shapes; ++s) shapes.push_back(std::make_shared<Shape>(shape_from_spec(spec.shapes[s], meshes)));
    std::vector<std::shared_ptr<Texture>> textures;
    for (uint32_t t = 0; t < spec.n_textures; ++t) {
        const pbrs_texture_spec& ts = spec.textures[t];
        auto tx = std::make_shared<Texture>();
        tx->kind = ts.kind;
        tx->odd = Color{ts.odd[0], ts.odd[1], ts.odd[2]};
        tx->even = Color{ts.even[0], ts.even[1], ts.even[2]};
        tx->freq = ts.freq;
        tx->width = ts.width;
        tx->height = ts.height;
        if (ts.kind == PBRS_TEX_PERLIN) {
            for (int k = 0; k < 256; ++k) tx->rand_vec.push_back(Vec3{ts.data[3 * k], ts.data[3 * k + 1], ts.data[3 * k + 2]});
            tx->perm_x.assign(ts.perm, ts.perm + 256);
            tx->perm_y.assign(ts.perm + 256, ts.perm + 512);
            tx->perm_z.assign(ts.perm + 512, ts.perm + 768);
        } else if (ts.kind == PBRS_TEX_IMAGE) {
            for (size_t k = 0; k < (size_t)ts.width * ts.height; ++k) tx->data.push_back(Color{ts.data[3 * k], ts.data[3 * k + 1], ts.data[3 * k + 2]});
        }
        textures.push_back(tx);
    }
    std::vector<std::shared_ptr<FourierTable>> fourier_tables;  // material::Fourier::from_file, material/src/lib.rs:455-460
    for (uint32_t t = 0; t < spec.n_fourier_tables; ++t) fourier_tables.push_back(FourierTable::build(spec.fourier_tables[t]));
    std::vector<std::shared_ptr<Material>> mtls;
    for (uint32_t m = 0; m < spec.n_materials; ++m) {
        auto mt = std::make_shared<Material>();
        mt->spec = spec.materials[m];
        if (mt->spec.kind == PBRS_MTL_FOURIER) {
            mt->fourier = fourier_tables.at(mt->spec.tex[0]);  // an index into fourier_tables[], not a texture
        } else {
            for (int k = 0; k < 4; ++k)
                if (spec.materials[m].tex[k]) mt->tex[k] = textures.at(spec.materials[m].tex[k] - 1);
        }
        mtls.push_back(mt);
    }
    std::vector<std::unique_ptr<Instance>> instances;
    for (uint32_t i = 0; i < spec.n_instances; ++i) {
        auto inst = std::make_unique<Instance>();
        inst->shape = shapes.at(spec.instances[i].shape);
        inst->mtl = mtls.at(spec.instances[i].material);
        std::memcpy(&inst->forward, spec.instances[i].forward, sizeof(Mat4));
        std::memcpy(&inst->inverse, spec.instances[i].inverse, sizeof(Mat4));
        inst->index = i;
        instances.push_back(std::move(inst));
    }
    scene->tlas = build_bvh(std::move(instances));
    for (uint32_t l = 0; l < spec.n_area_lights; ++l) {
        const pbrs_area_light_spec& a = spec.area_lights[l];
        DiffuseAreaLight light{};
        light.emit_radiance = Color{a.emit[0], a.emit[1], a.emit[2]};
        Shape sh = shape_from_spec(a.shape, meshes);
        light.shape.kind = sh.kind;
        light.shape.sphere = sh.sphere;
        light.shape.disk = sh.disk;
        light.shape.tri = sh.tri;
        light.shape.quad = sh.quad;
        light.area = light.shape.area();  // light/src/lib.rs:114-121
        scene->area_lights.push_back(light);
    }
    for (uint32_t l = 0; l < spec.n_delta_lights; ++l) {
        const pbrs_delta_light_spec& d = spec.delta_lights[l];
        DeltaLight light{};
        light.kind = d.kind;
        light.v = Vec3{d.v[0], d.v[1], d.v[2]};
        light.color = Color{d.color[0], d.color[1], d.color[2]};
        light.world_radius = d.world_radius;
        scene->delta_lights.push_back(light);
    }
    scene->env_constant = Color{spec.env_constant[0], spec.env_constant[1], spec.env_constant[2]};
    scene->env_kind = spec.env_kind;
    if (spec.env_kind == PBRS_ENV_IMAGE) scene->env_map = textures.at(spec.env_texture);
    scene->env_scale = Color{spec.env_scale[0], spec.env_scale[1], spec.env_scale[2]};
    scene->camera = camera_from_spec(spec.camera);
    return scene;
}

// ---- scene/src/lib.rs:105-117, scene/src/preset.rs:25-53 ---------------------------------------------------------
Color Scene::eval_env_light(const Ray& ray) const {
    switch (env_kind) {
        case PBRS_ENV_IMAGE: {  // lib.rs:108-114
            float phi = pn_atan2(ray.dir.z, ray.dir.x);
            float u = pn_fract(phi * PN_FRAC_1_PI * 0.5f + 1.0f);
            float cos_theta = ray.dir.y / norm(ray.dir);
            float v = pn_acos(cos_theta) / PN_PI;
            return env_map->value(u, v, Point3{0.0f, 0.0f, 0.0f}) * env_scale;
        }
        case PBRS_ENV_BLUE_SKY: {  // preset.rs:25-30
            float y = (hat(ray.dir).y + 1.0f) * 0.5f;
            return Color{0.5f, 0.7f, 1.0f} * y + gray(1.0f) * (1.0f - y);
        }
        case PBRS_ENV_DARK_ROOM: {  // :32-37
            float y = (hat(ray.dir).y + 1.0f) * 0.5f;
            return gray(0.1f) * y + gray(0.1f) * (1.0f - y);
        }
        case PBRS_ENV_DUSK: {  // :39-52; Color::rgb(u8, u8, u8) = x as f32 / 255.0 (radiometry/src/color.rs)
            Color horizon{245.0f / 255.0f, 174.0f / 255.0f, 82.0f / 255.0f};
            Color dome{109.0f / 255.0f, 150.0f / 255.0f, 204.0f / 255.0f};
            float tilt = pn_acos(hat(ray.dir).y);
            if (tilt > PN_PI * 0.25f) return dome;
            if (tilt > 0.0f) {
                float t = tilt / (PN_PI * 0.25f);
                return dome * t + horizon * (1.0f - t);
            }
            return gray(0.2f);
        }
        default: return env_constant;
    }
}

// ---- src/directlighting.rs ----------------------------------------------------------------------------
static float power_heuristic2(float nf, float f_pdf, float ng, float g_pdf) {  // :224-232 with BETA = 2
    float f = nf * f_pdf;
    float g = ng * g_pdf;
    return pn_powi(f, 2) / (pn_powi(f, 2) + pn_powi(g, 2));
}

static bool scene_occludes(const Scene& scene, const Ray& r) {
    REF_COUNT(shadow_rays);
    return scene.tlas->occludes(r);
}

static Color estimate_direct_delta_light(const Interaction& hit, const Material& mtl, const DeltaLight& light, const Scene& scene) {  // :101-153
    std::vector<BXDF> bxdfs = mtl.bxdfs_at(hit);
    REF_ASSERT(!bxdfs.empty());
    BSDF bsdf = bsdf_new_frame(hit);
    bsdf.bxdfs = &bxdfs;
    Color light_radiance;
    Vec3 wi;
    Prob light_pr{};
    Ray vis;
    light.sample_incident_radiance(hit, &light_radiance, &wi, &light_pr, &vis);
    Color bsdf_value = bsdf.eval(hit.wo, wi) * pn_abs(dot(hit.normal, wi));
    if (!light_pr.is_positive() || is_black(light_radiance) || is_black(bsdf_value)) return black();
    float scatter_pdf = bsdf.pdf(hit.wo, wi);
    if (scene_occludes(scene, vis)) return black();
    float weight, pr;
    if (light_pr.is_mass) {
        weight = 1.0f;
        pr = light_pr.v;
    } else {
        weight = power_heuristic2(1.0f, light_pr.v, 1.0f, scatter_pdf);
        pr = light_pr.v;
    }
    return (bsdf_value * light_radiance) * weight * pn_weak_recip(pr);
}

static Color estimate_direct_area_light(const Interaction& hit, const Material& mtl, float su, float sv, const DiffuseAreaLight& light,
                                        float lu, float lv, const Scene& scene) {  // :155-222
    Color radiance_d = black();
    std::vector<BXDF> bxdfs = mtl.bxdfs_at(hit);
    BSDF bsdf = bsdf_new_frame(hit);
    bsdf.bxdfs = &bxdfs;

    Color light_radiance;
    Vec3 wi;
    Prob light_pr{};
    Ray vis;
    light.sample_incident_radiance(hit, lu, lv, &light_radiance, &wi, &light_pr, &vis);
    REF_ASSERT(light_pr.is_density());
    if (light_pr.is_positive() && !is_black(light_radiance)) {
        float light_pdf = light_pr.density();
        Color bsdf_value = bsdf.eval(hit.wo, wi) * pn_abs(dot(hit.normal, wi));
        float scatter_pdf = bsdf.pdf(hit.wo, wi);
        if (!is_black(bsdf_value) && scatter_pdf > 0.0f && !scene_occludes(scene, vis)) {
            float weight = power_heuristic2(1.0f, light_pdf, 1.0f, scatter_pdf);
            radiance_d = radiance_d + bsdf_value * light_radiance * weight * pn_weak_recip(light_pdf);
        }
    }
    // by_bsdf (:198-220)
    {
        Color bsdf_value;
        Vec3 wi2;
        Prob bsdf_pr{};
        bsdf.sample(hit.wo, su, sv, &bsdf_value, &wi2, &bsdf_pr);
        bsdf_value = bsdf_value * pn_abs(dot(hit.normal, wi2));
        if (!(is_black(bsdf_value) || !bsdf_pr.is_positive())) {
            Color incident_radiance;
            float light_pdf;
            Ray vis2;
            if (light.radiance_to(hit, wi2, &incident_radiance, &light_pdf, &vis2)) {
                if (!(is_black(incident_radiance) || light_pdf <= 0.0f || scene_occludes(scene, vis2))) {
                    float weight, pr;
                    if (bsdf_pr.is_mass) {
                        weight = 1.0f;
                        pr = bsdf_pr.v;
                    } else {
                        weight = power_heuristic2(1.0f, bsdf_pr.v, 1.0f, light_pdf);
                        pr = bsdf_pr.v;
                    }
                    Color f = bsdf_value * incident_radiance;
                    radiance_d = radiance_d + weight * f * pn_weak_recip(pr);
                }
            }
        }
    }
    return radiance_d;
}

static Color uniform_sample_one_light(const Interaction& hit, const Material& mtl, const Scene& scene, uint64_t* rng) {  // :58-99
    size_t num_lights = scene.delta_lights.size() + scene.area_lights.size() + (scene.has_env_light() ? 1 : 0);
    if (num_lights == 0) return black();
    float light_pdf = 1.0f / (float)num_lights;
    // RNG contract: gen_range(0..n) becomes min(floor(u*n), n-1).
    float uidx = pn_rng_f32(rng);
    size_t chosen_index = (size_t)(uidx * (float)num_lights);
    if (chosen_index > num_lights - 1) chosen_index = num_lights - 1;
    float lu = pn_rng_f32(rng), lv = pn_rng_f32(rng);
    float su = pn_rng_f32(rng), sv = pn_rng_f32(rng);
    Color one;
    if (chosen_index < scene.delta_lights.size()) {
        one = estimate_direct_delta_light(hit, mtl, scene.delta_lights[chosen_index], scene);
    } else if (chosen_index >= scene.delta_lights.size() && chosen_index < scene.area_lights.size()) {  // Q6
        one = estimate_direct_area_light(hit, mtl, su, sv, scene.area_lights[chosen_index - scene.delta_lights.size()], lu, lv, scene);
    } else {  // :80-96
        std::vector<BXDF> bxdfs = mtl.bxdfs_at(hit);
        REF_ASSERT(!bxdfs.empty());
        BSDF bsdf = bsdf_new_frame(hit);
        bsdf.bxdfs = &bxdfs;
        Color f;
        Vec3 wi;
        Prob pr{};
        bsdf.sample(hit.wo, su, sv, &f, &wi, &pr);
        Ray incident_ray = spawn_ray(hit, wi);
        Color incident_radiance = scene_occludes(scene, incident_ray) ? black() : scene.eval_env_light(incident_ray);
        one = incident_radiance * f * pn_abs(dot(wi, hit.normal)) * pn_weak_recip(pr.v);
    }
    return one * (1.0f / light_pdf);
}

// ---- src/pathintegrator.rs:9-74 --------------------------------------------------------------------------
static Color path_integrator(const Scene& scene, Ray ray, int depth, uint64_t* rng, oracle_path_trace* trace) {
    Color radiance = black();
    bool specular_bounce = false;
    Color beta = gray(1.0f);
    if (trace) trace->n_bounces = 0;
    for (int bounces = 0; bounces < depth; ++bounces) {
        Hit h;
        REF_COUNT(closest_rays);
        bool has_hit = scene.tlas->intersect(ray, &h);
        if (trace && bounces < ORACLE_TRACE_MAX_BOUNCES) {
            oracle_bounce_trace& b = trace->bounce[bounces];
            std::memset(&b, 0, sizeof(b));
            b.hit = has_hit ? 1 : 0;
            if (has_hit) {
                b.t = h.isect.ray_t;
                b.inst = h.inst->index;
                b.prim = h.isect.prim;
                b.b1 = h.isect.b1;
                b.b2 = h.isect.b2;
                b.pos[0] = h.isect.pos.x; b.pos[1] = h.isect.pos.y; b.pos[2] = h.isect.pos.z;
                b.normal[0] = h.isect.normal.x; b.normal[1] = h.isect.normal.y; b.normal[2] = h.isect.normal.z;
            }
            trace->n_bounces = bounces + 1;
        }
        if (bounces == 0 || specular_bounce) {
            Color e = has_hit ? h.inst->mtl->emission() : scene.eval_env_light(ray);
            radiance = radiance + beta * e;
        }
        if (!has_hit) break;
        REF_COUNT(shade_events);
        const Interaction& hit = h.isect;
        const Material& mtl = *h.inst->mtl;
        std::vector<BXDF> bxdfs = mtl.bxdfs_at(hit);

        radiance = radiance + beta * uniform_sample_one_light(hit, mtl, scene, rng);

        BSDF shading_point = bsdf_new_frame(hit);
        shading_point.bxdfs = &bxdfs;
        float r0 = pn_rng_f32(rng), r1 = pn_rng_f32(rng);
        Color f;
        Vec3 wi;
        Prob pr{};
        shading_point.sample(-ray.dir, r0, r1, &f, &wi, &pr);
        if (trace && bounces < ORACLE_TRACE_MAX_BOUNCES) {
            oracle_bounce_trace& b = trace->bounce[bounces];
            b.radiance_after_nee[0] = radiance.r; b.radiance_after_nee[1] = radiance.g; b.radiance_after_nee[2] = radiance.b;
            b.f[0] = f.r; b.f[1] = f.g; b.f[2] = f.b;
            b.wi[0] = wi.x; b.wi[1] = wi.y; b.wi[2] = wi.z;
            b.pr = pr.v;
            b.pr_is_mass = pr.is_mass;
        }
        if (is_black(f) || pr.is_zero()) break;
        specular_bounce = pr.is_mass;
        float p = pr.v;
        beta = beta * f * dot(wi, hit.normal) * pn_recip(p);  // Q9: no abs
        ray = spawn_ray(hit, wi);
        if (bounces > 3) {
            float q = pn_max(1.0f - luminance(beta), 0.05f);
            if (pn_rng_f32(rng) < q) break;
            beta = beta * pn_recip(1.0f - q);
        }
        if (trace && bounces < ORACLE_TRACE_MAX_BOUNCES) {
            oracle_bounce_trace& b = trace->bounce[bounces];
            b.beta_after[0] = beta.r; b.beta_after[1] = beta.g; b.beta_after[2] = beta.b;
        }
    }
    return radiance;
}

// ---- src/directlighting.rs:14-56 -------------------------------------------------------------------------
// direct_lighting_debug_integrator (:49-56): the light estimate at the first hit, or the environment; `_depth` unused.
static Color direct_lighting_debug_integrator(const Scene& scene, Ray ray, uint64_t* rng) {
    Hit h;
    REF_COUNT(closest_rays);
    if (scene.tlas->intersect(ray, &h)) {
        REF_COUNT(shade_events);
        return uniform_sample_one_light(h.isect, *h.inst->mtl, scene, rng);
    }
    return scene.eval_env_light(ray);
}
// material_visualizer (src/directlighting.rs:234-271): a palette colour for the kind of material at the first hit
// (`match mtl.summary()`, summaries in material/src/lib.rs), a grey checker of the ray direction where nothing is hit.
static Color material_visualizer(const Scene& scene, Ray ray) {
    static const uint8_t pal8[7][3] = {{232, 207, 59}, {124, 188, 126}, {30, 68, 176}, {15, 142, 205}, {44, 180, 172}, {216, 39, 252}, {143, 112, 252}};
    Hit h;
    REF_COUNT(closest_rays);
    if (scene.tlas->intersect(ray, &h)) {
        REF_COUNT(shade_events);
        int index;
        switch (h.inst->mtl->spec.kind) {  // the first arm whose string the summary contains
            case PBRS_MTL_LAMBERTIAN: index = 8; break;     // "Lambertian"
            case PBRS_MTL_METAL: index = 7; break;          // "Metal{ior = ...}"
            case PBRS_MTL_FOURIER: index = 6; break;        // "Fourier"
            case PBRS_MTL_MIRROR: index = 5; break;         // "Mirror{albedo = ...}"
            case PBRS_MTL_DIELECTRIC: index = 4; break;     // "Dielectric{ior = ...}"
            case PBRS_MTL_DIFFUSE_LIGHT: index = 3; break;  // "DiffuseLight{emit = ...}"
            case PBRS_MTL_UBER: index = 2; break;           // "uber"
            case PBRS_MTL_SUBSTRATE: index = 1; break;      // "substrate"
            case PBRS_MTL_PLASTIC: index = 0; break;        // "plastic"
            default: index = 9; break;                      // "Glossy" matches no arm
        }
        if (index < 7) return Color{(float)pal8[index][0] / 255.0f, (float)pal8[index][1] / 255.0f, (float)pal8[index][2] / 255.0f};  // Color::rgb
        return index == 7 ? gray(0.3f) : index == 8 ? gray(0.9f) : black();
    }
    // `(x * 50.0).floor() as i32 + (y * 50.0).floor() as i32` (wrapping, as a release build adds)
    int parity = (int)((uint32_t)pn_f32_to_i32(pn_floor(ray.dir.x * 50.0f)) + (uint32_t)pn_f32_to_i32(pn_floor(ray.dir.y * 50.0f)));
    return parity % 2 == 0 ? gray(0.9f) : gray(0.7f);
}
// normal_visualizer (src/directlighting.rs:273-289): (albedo of `mtl.scatter(-ray.dir, &hit)` + hit.normal) * 0.5, the
// environment where nothing is hit.  `scatter` per material: material/src/lib.rs:163-177 (Lambertian), :192-199 (Metal),
// :224-228 (Mirror), :246-264 (Dielectric; its one `rand::random::<f32>()` is drawn from `rng`), :282-286 (DiffuseLight),
// :427-432 (Plastic); Glossy :213-215, Uber :314-316, Substrate :390-392 and Fourier :463-465 are `todo!()`: counted as a
// panic, the pixel is black.  The directions `scatter` also draws do not reach the returned colour.
static Color normal_visualizer(const Scene& scene, Ray ray, uint64_t* rng) {
    auto c3 = [](const float* q) { return Color{q[0], q[1], q[2]}; };
    Hit h;
    REF_COUNT(closest_rays);
    if (!scene.tlas->intersect(ray, &h)) return scene.eval_env_light(ray);
    REF_COUNT(shade_events);
    const Interaction& isect = h.isect;
    const Material& m = *h.inst->mtl;
    const float* p = m.spec.p;
    const Vec3 wi_in = -ray.dir;
    Color albedo = black();
    switch (m.spec.kind) {
        case PBRS_MTL_LAMBERTIAN: albedo = m.tex[0] ? m.tex[0]->value(isect.u, isect.v, isect.pos) : c3(p); break;
        case PBRS_MTL_METAL: albedo = fresnel_conductor(c3(p), c3(p + 3)).eval(pn_abs(dot(isect.normal, wi_in))); break;
        case PBRS_MTL_MIRROR: albedo = c3(p); break;
        case PBRS_MTL_PLASTIC: albedo = c3(p); break;  // self.diffuse
        case PBRS_MTL_DIFFUSE_LIGHT: break;
        case PBRS_MTL_DIELECTRIC: {
            const float ior = p[0];
            const Color reflect_c = c3(p + 1), transmit_c = c3(p + 4);
            Vec3 wi = hat(wi_in);
            Vec3 outward;
            float ratio, cosine;
            if (dot(isect.normal, wi) < 0.0f) {
                outward = -isect.normal;
                ratio = ior;
                cosine = -dot(isect.normal, wi);
            } else {
                outward = isect.normal;
                ratio = 1.0f / ior;
                cosine = dot(isect.normal, wi);
            }
            Vec3 wt;
            float reflect_pr = 1.0f;
            albedo = reflect_c;
            if (refract(outward, wi, ratio, &wt)) {
                float r0 = (1.0f - ior) / (1.0f + ior);  // schlick, :477-481
                r0 = r0 * r0;
                reflect_pr = r0 + (1.0f - r0) * pn_powi(1.0f - cosine, 5);
                albedo = transmit_c;
            }
            if (pn_rng_f32(rng) < reflect_pr) albedo = reflect_c;
            break;
        }
        default: REF_ASSERT(false && "Material::scatter: todo!()"); break;
    }
    return (albedo + Color{isect.normal.x, isect.normal.y, isect.normal.z}) * 0.5f;
}
// direct_lighting_integrator (:14-47): emitters return their emission; everything else the one-light estimate plus one
// level of perfect-specular reflection/refraction followed by the debug integrator.
static Color direct_lighting_integrator(const Scene& scene, Ray ray, int depth, uint64_t* rng) {
    if (depth <= 0) return black();
    Hit h;
    REF_COUNT(closest_rays);
    if (!scene.tlas->intersect(ray, &h)) return scene.eval_env_light(ray);
    REF_COUNT(shade_events);
    const Interaction& hit = h.isect;
    const Material& mtl = *h.inst->mtl;
    if (!is_black(mtl.emission())) return mtl.emission();
    Color direct = uniform_sample_one_light(hit, mtl, scene, rng);
    std::vector<BXDF> bxdfs = mtl.bxdfs_at(hit);
    BSDF bsdf = bsdf_new_frame(hit);
    bsdf.bxdfs = &bxdfs;
    Color spec_refl = black();
    Color f;
    Vec3 wi;
    Prob pr{};
    if (bsdf.sample_specular(hit.wo, &f, &wi, &pr)) {
        REF_ASSERT(pr.is_mass);
        Ray refl_ray = spawn_ray(hit, wi);
        Color s = direct_lighting_debug_integrator(scene, refl_ray, rng);
        spec_refl = s * f * pn_weak_recip(pr.v);
    }
    return direct + spec_refl;
}

}  // namespace ref

// ===== C entry points ========================================================================================
using namespace ref;

struct oracle_scene {
    std::unique_ptr<Scene> scene;
    Diag build_diag;
};

extern "C" {

oracle_scene* oracle_scene_build(const pbrs_scene_spec* spec) {
    auto* s = new oracle_scene();
    g_diag = &s->build_diag;
    s->scene = scene_from_spec(*spec);
    g_diag = nullptr;
    return s;
}
void oracle_scene_free(oracle_scene* s) { delete s; }

uint32_t oracle_tlas_height(const oracle_scene* s) { return s->scene->tlas->height(); }

static void copy_counters(const Counters& c, const Diag& d, oracle_stats* out) {
    if (!out) return;
    out->closest_rays = c.closest_rays; out->shadow_rays = c.shadow_rays; out->tlas_nodes = c.tlas_nodes;
    out->blas_nodes = c.blas_nodes; out->instances = c.instances; out->instance_hits = c.instance_hits;
    out->triangles = c.triangles; out->spheres = c.spheres; out->quads = c.quads; out->cuboids = c.cuboids;
    out->disks = c.disks; out->tri_shading = c.tri_shading; out->shade_events = c.shade_events; out->samples = c.samples;
    out->panics = d.panics; out->tlas_ties = d.tlas_ties; out->sphere_inside = d.sphere_inside;
    out->nonfinite_samples = d.nonfinite_samples;
}

// src/main.rs:192-231 restricted to the tile [x0,x0+w) x [y0,y0+h); rows are dealt to `nthreads`
// std::threads (rayon's into_par_iter over rows, :219-224).  strata_x = strata_y = msaa reproduces
// the reference's `i / msaa`, `i % msaa` stratification (:197-201).
int oracle_render_tile(const oracle_scene* os, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint32_t strata_x, uint32_t strata_y,
                       uint32_t max_depth, uint64_t seed, uint32_t nthreads, float* rgb_out, oracle_stats* stats_out) {
    return oracle_render_tile_integrator(os, x0, y0, w, h, strata_x, strata_y, max_depth, seed, nthreads, 0, rgb_out, stats_out);
}

// `integrator`: 0 = path_integrator (src/pathintegrator.rs), 1 = direct_lighting_integrator (src/directlighting.rs:14-47),
// 2 = material_visualizer (:234-271), 3 = normal_visualizer (:273-289) — both with strata 1 x 1, the ray through the pixel's
// corner: `shoot_ray(row, col, (0.0, 0.0))`, src/main.rs:170; all fit the reference's seam `fn(&Scene, Ray, i32) -> Color` (src/main.rs:160-187).
int oracle_render_tile_integrator(const oracle_scene* os, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint32_t strata_x,
                                  uint32_t strata_y, uint32_t max_depth, uint64_t seed, uint32_t nthreads, uint32_t integrator,
                                  float* rgb_out, oracle_stats* stats_out) {
    if (integrator > 3 || (integrator >= 2 && (strata_x != 1 || strata_y != 1))) return -1;
    const Scene& scene = *os->scene;
    if (nthreads == 0) nthreads = 1;
    const uint32_t width = scene.camera.width;
    const uint32_t spp = strata_x * strata_y;
    std::atomic<uint32_t> next_row{0};
    std::vector<Counters> cnts(nthreads);
    std::vector<Diag> diags(nthreads);
    auto worker = [&](uint32_t tid) {
        g_cnt = &cnts[tid];
        g_diag = &diags[tid];
        for (;;) {
            uint32_t ry = next_row.fetch_add(1);
            if (ry >= h) break;
            uint32_t row = y0 + ry;
            for (uint32_t cx = 0; cx < w; ++cx) {
                uint32_t col = x0 + cx;
                Color color_sum = black();
                for (uint32_t i = 0; i < spp; ++i) {
                    uint64_t rng = pn_rng_init(seed, row * width + col, i);
                    float r0 = pn_rng_f32(&rng), r1 = pn_rng_f32(&rng);
                    float jx = ((float)(i / strata_y) + r0) / (float)strata_x;
                    float jy = ((float)(i % strata_y) + r1) / (float)strata_y;
                    if (integrator >= 2) jx = jy = 0.0f;
                    Ray ray;
                    scene.camera.shoot_ray(row, col, jx, jy, &ray);
                    REF_COUNT(samples);
                    const Color li = integrator == 0   ? path_integrator(scene, ray, (int)max_depth, &rng, nullptr)
                                     : integrator == 1 ? direct_lighting_integrator(scene, ray, (int)max_depth, &rng)
                                     : integrator == 2 ? material_visualizer(scene, ray)
                                                       : normal_visualizer(scene, ray, &rng);
                    if (!(pn_isfinite(li.r) && pn_isfinite(li.g) && pn_isfinite(li.b)) && g_diag) g_diag->nonfinite_samples++;
                    color_sum = color_sum + li;
                }
                Color color = color_sum * (1.0f / (float)spp);  // scale_down_by, color.rs:90-95
                float* px = rgb_out + 3 * ((size_t)ry * w + cx);
                px[0] = color.r;
                px[1] = color.g;
                px[2] = color.b;
            }
        }
        g_cnt = nullptr;
        g_diag = nullptr;
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < nthreads; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (auto& t : pool) t.join();
    Counters total;
    Diag dtotal;
    for (uint32_t t = 0; t < nthreads; ++t) {
        total.add(cnts[t]);
        dtotal.panics += diags[t].panics;
        dtotal.tlas_ties += diags[t].tlas_ties;
        dtotal.sphere_inside += diags[t].sphere_inside;
        dtotal.nonfinite_samples += diags[t].nonfinite_samples;
    }
    copy_counters(total, dtotal, stats_out);
    return 0;
}

// One camera sample with a per-bounce trace (hit record, radiance after NEE, sampled f / wi / pr, beta).
int oracle_trace_sample(const oracle_scene* os, uint32_t row, uint32_t col, uint32_t sample_index, uint32_t strata_x, uint32_t strata_y,
                        uint32_t max_depth, uint64_t seed, oracle_path_trace* trace) {
    const Scene& scene = *os->scene;
    Diag d;
    g_diag = &d;
    uint64_t rng = pn_rng_init(seed, row * scene.camera.width + col, sample_index);
    float r0 = pn_rng_f32(&rng), r1 = pn_rng_f32(&rng);
    float jx = ((float)(sample_index / strata_y) + r0) / (float)strata_x;
    float jy = ((float)(sample_index % strata_y) + r1) / (float)strata_y;
    Ray ray;
    scene.camera.shoot_ray(row, col, jx, jy, &ray);
    trace->ray_o[0] = ray.origin.x; trace->ray_o[1] = ray.origin.y; trace->ray_o[2] = ray.origin.z;
    trace->ray_d[0] = ray.dir.x; trace->ray_d[1] = ray.dir.y; trace->ray_d[2] = ray.dir.z;
    Color L = path_integrator(scene, ray, (int)max_depth, &rng, trace);
    trace->radiance[0] = L.r; trace->radiance[1] = L.g; trace->radiance[2] = L.b;
    trace->panics = (uint32_t)d.panics;
    g_diag = nullptr;
    return 0;
}

// Closest hit / any hit for caller-supplied rays (tlas/src/bvh.rs:77-113): the `extend` / `shadow`
// kernel parity harness.  hits_out: n records of {t, inst, prim, b1, b2} (inst = 0xffffffff on a miss).
int oracle_intersect_rays(const oracle_scene* os, uint32_t n, const float* origins, const float* dirs, const float* tmax,
                          oracle_hit_record* hits_out, uint8_t* occluded_out, oracle_stats* stats_out, uint8_t* tie_out) {
    const Scene& scene = *os->scene;
    Counters c;
    Diag d;
    g_cnt = &c;
    g_diag = &d;
    for (uint32_t i = 0; i < n; ++i) {
        Ray r{Vec3{origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]}, Vec3{dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]}, tmax[i]};
        if (hits_out) {
            Ray rm = r;
            Hit h;
            c.closest_rays++;
            uint64_t ties_before = d.tlas_ties;
            bool found = scene.tlas->intersect(rm, &h);
            if (tie_out) tie_out[i] = d.tlas_ties != ties_before ? 1 : 0;
            if (found) {
                hits_out[i].t = h.isect.ray_t;
                hits_out[i].inst = h.inst->index;
                hits_out[i].prim = h.isect.prim;
                hits_out[i].b1 = h.isect.b1;
                hits_out[i].b2 = h.isect.b2;
            } else {
                hits_out[i].t = pn_inf();
                hits_out[i].inst = 0xffffffffu;
                hits_out[i].prim = 0;
                hits_out[i].b1 = hits_out[i].b2 = 0.0f;
            }
        }
        if (occluded_out) {
            c.shadow_rays++;
            occluded_out[i] = scene.tlas->occludes(r) ? 1 : 0;
        }
    }
    copy_counters(c, d, stats_out);
    g_cnt = nullptr;
    g_diag = nullptr;
    return 0;
}

// Camera rays exactly as src/main.rs:197-203 generates them (for raygen parity).
int oracle_camera_rays(const oracle_scene* os, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint32_t sample_index,
                       uint32_t strata_x, uint32_t strata_y, uint64_t seed, float* origins, float* dirs) {
    const Scene& scene = *os->scene;
    for (uint32_t ry = 0; ry < h; ++ry)
        for (uint32_t cx = 0; cx < w; ++cx) {
            uint32_t row = y0 + ry, col = x0 + cx;
            uint64_t rng = pn_rng_init(seed, row * scene.camera.width + col, sample_index);
            float r0 = pn_rng_f32(&rng), r1 = pn_rng_f32(&rng);
            float jx = ((float)(sample_index / strata_y) + r0) / (float)strata_x;
            float jy = ((float)(sample_index % strata_y) + r1) / (float)strata_y;
            Ray ray;
            scene.camera.shoot_ray(row, col, jx, jy, &ray);
            size_t k = (size_t)ry * w + cx;
            origins[3 * k] = ray.origin.x; origins[3 * k + 1] = ray.origin.y; origins[3 * k + 2] = ray.origin.z;
            dirs[3 * k] = ray.dir.x; dirs[3 * k + 1] = ray.dir.y; dirs[3 * k + 2] = ray.dir.z;
        }
    return 0;
}

// The f32 numeric contract, for ulp checks against libm and bitwise checks against gfx950.
int oracle_texture_value(const pbrs_texture_spec* ts, uint32_t n, const float* uv, const float* pos, float* rgb_out) {
    pbrs_scene_spec spec{};
    spec.n_textures = 1;
    spec.textures = ts;
    pbrs_camera_spec cam{};
    cam.width = cam.height = 1;
    cam.fov_y_rad = 1.0f;
    cam.target[2] = 1.0f;
    cam.up[1] = 1.0f;
    spec.camera = cam;
    // only the texture table is needed: build it the way scene_from_spec does
    Texture tx;
    tx.kind = ts->kind;
    tx.odd = Color{ts->odd[0], ts->odd[1], ts->odd[2]};
    tx.even = Color{ts->even[0], ts->even[1], ts->even[2]};
    tx.freq = ts->freq;
    tx.width = ts->width;
    tx.height = ts->height;
    if (ts->kind == PBRS_TEX_PERLIN) {
        for (int k = 0; k < 256; ++k) tx.rand_vec.push_back(Vec3{ts->data[3 * k], ts->data[3 * k + 1], ts->data[3 * k + 2]});
        tx.perm_x.assign(ts->perm, ts->perm + 256);
        tx.perm_y.assign(ts->perm + 256, ts->perm + 512);
        tx.perm_z.assign(ts->perm + 512, ts->perm + 768);
    } else if (ts->kind == PBRS_TEX_IMAGE) {
        for (size_t k = 0; k < (size_t)ts->width * ts->height; ++k) tx.data.push_back(Color{ts->data[3 * k], ts->data[3 * k + 1], ts->data[3 * k + 2]});
    }
    Diag diag;
    g_diag = &diag;
    for (uint32_t i = 0; i < n; ++i) {
        Color c = tx.value(uv[2 * i], uv[2 * i + 1], Point3{pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]});
        rgb_out[3 * i] = c.r;
        rgb_out[3 * i + 1] = c.g;
        rgb_out[3 * i + 2] = c.b;
    }
    g_diag = nullptr;
    return (int)diag.panics;
}
int oracle_env_eval(const oracle_scene* os, uint32_t n, const float* dirs, float* rgb_out) {
    for (uint32_t i = 0; i < n; ++i) {
        Ray r{};
        r.origin = Point3{0.0f, 0.0f, 0.0f};
        r.dir = Vec3{dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]};
        r.t_max = pn_inf();
        Color c = os->scene->eval_env_light(r);
        rgb_out[3 * i] = c.r;
        rgb_out[3 * i + 1] = c.g;
        rgb_out[3 * i + 2] = c.b;
    }
    return 0;
}

int oracle_numeric_eval(uint32_t fn, uint32_t n, const float* x, const float* y, float* out) {
    for (uint32_t i = 0; i < n; ++i) {
        float a = x[i], b = y ? y[i] : 0.0f, r;
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
            default: return -1;
        }
        out[i] = r;
    }
    return 0;
}
int oracle_temperature_to_color(float kelvin, float* rgb_out) {
    Diag d;
    g_diag = &d;
    Color c = temperature_to_color(kelvin);
    g_diag = nullptr;
    rgb_out[0] = c.r, rgb_out[1] = c.g, rgb_out[2] = c.b;
    return (int)d.panics;
}
int oracle_spd_to_color(uint32_t n, const float* lambdas_nm, const float* values, float* rgb_out) {
    Diag d;
    g_diag = &d;
    std::vector<std::pair<float, float>> samples;
    for (uint32_t i = 0; i < n; ++i) samples.push_back({lambdas_nm[i], values[i]});
    Color c = sampled_spectrum_to_color(samples);
    g_diag = nullptr;
    rgb_out[0] = c.r, rgb_out[1] = c.g, rgb_out[2] = c.b;
    return (int)d.panics;
}
int oracle_rng_stream(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, float* out) {
    uint64_t s = pn_rng_init(seed, pixel, sample);
    for (uint32_t i = 0; i < n; ++i) out[i] = pn_rng_f32(&s);
    return 0;
}

}  // extern "C"
