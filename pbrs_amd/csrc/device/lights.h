// device/lights.h — light sampling for next-event estimation in the `shade` stage.
//
// Restates light/src/lib.rs (DeltaLight :67-92, DiffuseAreaLight :127-172) and
// light/src/sample_shape.rs (default pdf_at :28-33, Sphere :185-254, Disk :258-273,
// IsolatedTriangle :277-293, ParallelQuad :297-308).  Light shapes live in world space
// (pbrs_area_light.p); the ray-vs-own-shape tests of pdf_at / radiance_to are not scene rays.
#pragma once
#include "bsdf.h"

// Interaction::spawn_ray / spawn_limited_ray_to (geometry/src/interaction.rs:63-70)
PD void spawn_ray(const Isect& is, f3 dir, f3& o, f3& d) {
    f3 out_normal = pn_signum(dot(dir, is.normal)) * is.normal;
    o = is.pos + out_normal * 0.001f;
    d = dir;
}

struct LightPoint {
    f3 pos, normal;
};

// `kind`: the light's shape kind — L.shape_kind, or a compile-time constant in the k_shade variants for scenes whose area
// lights all have one shape (PBRS_SHADE_LIGHT_*, chosen at upload): the record's kind need not be read and the other
// shapes' sampling and intersection code is gone.
PD bool light_shape_intersect(const pbrs_area_light& L, uint32_t kind, f3 o, f3 d, LightPoint& out) {
    const float* p = L.p;
    switch (kind) {
        case PBRS_SHAPE_SPHERE: {
            float t;
            if (!sphere_hit_t(ld3(p), p[3], o, d, pn_inf(), t)) return false;
            Isect i = sphere_isect(ld3(p), p[3], o, d, t);
            out.pos = i.pos;
            out.normal = i.normal;
            return true;
        }
        case PBRS_SHAPE_DISK: {
            float t;
            if (!disk_hit_t(ld3(p), ld3(p + 3), ld3(p + 6), o, d, pn_inf(), t)) return false;
            Isect i = disk_isect(ld3(p), ld3(p + 3), ld3(p + 6), o, d, t);
            out.pos = i.pos;
            out.normal = i.normal;
            return true;
        }
        case PBRS_SHAPE_TRIANGLE: {
            f3 p0 = ld3(p), p1 = ld3(p + 3), p2 = ld3(p + 6);
            TriHit h;
            if (!tri_hit(p0, p1, p2, o, d, pn_inf(), h)) return false;
            out.pos = bary_lerp(p0, p1, p2, h.b0, h.b1);
            out.normal = h.normal;
            return true;
        }
        default: {
            float t, u, v;
            f3 n;
            if (!quad_hit(ld3(p), ld3(p + 3), ld3(p + 6), o, d, pn_inf(), t, u, v, n)) return false;
            out.pos = ld3(p) + u * ld3(p + 3) + ld3(p + 6) * v;
            out.normal = hat(n);
            return true;
        }
    }
}

PD LightPoint sphere_sample(f3 center, float radius, float u, float v) {  // sample_shape.rs:185-195
    float theta = 2.0f * PN_PI * u;
    float phi = pn_acos(2.0f * v - 1.0f);
    f3 dir = mk3(pn_sin(phi) * pn_cos(theta), pn_sin(phi) * pn_sin(theta), 2.0f * v - 1.0f);
    return LightPoint{center + radius * dir, dir};
}
PD LightPoint sphere_sample_towards(f3 center, float radius, f3 target_pos, float u, float v) {  // :197-236
    f3 wc = center - target_pos;
    if (norm2(wc) < pn_sq(radius)) return sphere_sample(center, radius, u, v);
    float sin_theta_max_2 = pn_sq(radius) / norm2(wc);
    float cos_theta_max = pn_sqrt(pn_max(1.0f - sin_theta_max_2, 0.0f));
    float cos_t = (1.0f - u) + u * cos_theta_max;
    float sin_theta_2 = pn_max(1.0f - pn_sq(cos_t), 0.0f);
    float phi = v * 2.0f * PN_PI;
    float dc = norm(wc);
    float ds = dc * cos_t - pn_sqrt(pn_max(pn_sq(radius) - norm2(wc) * sin_theta_2, 0.0f));
    float cos_alpha = (norm2(wc) + pn_sq(radius) - pn_sq(ds)) / (2.0f * dc * radius);
    float sin_alpha = pn_sqrt(pn_max(1.0f - pn_sq(cos_alpha), 0.0f));
    f3 normal_object_space = spherical_direction(sin_alpha, cos_alpha, phi);
    f3 wcx, wcy;
    make_coord_system(-hat(wc), wcx, wcy);
    f3 normal_world_space = mat3_mul(wcx, wcy, -hat(wc), normal_object_space);
    f3 point_on_sphere = normal_world_space * radius + center;
    return LightPoint{point_on_sphere, normal_world_space};
}
PD bool sphere_pdf_at(f3 center, float radius, f3 ref_pos, f3 wi, float& pdf) {  // :238-250
    f3 ref_to_center = center - ref_pos;
    if (norm2(ref_to_center) < pn_sq(radius)) {
        pdf = 1.0f / (pn_sq(radius) * 4.0f * PN_PI);
        return true;
    }
    float sin_theta_max_2 = pn_sq(radius) / norm2(ref_to_center);
    float cos_theta_max = pn_sqrt(pn_max(1.0f - sin_theta_max_2, 0.0f));
    float cos_t = dot(ref_to_center, wi) / (norm(ref_to_center) * norm(wi));
    if (cos_t > cos_theta_max) {
        pdf = 1.0f / (2.0f * PN_PI * (1.0f - cos_theta_max));
        return true;
    }
    return false;
}

PD LightPoint light_sample_towards(const pbrs_area_light& L, uint32_t kind, const Isect& target, float u, float v) {
    const float* p = L.p;
    switch (kind) {
        case PBRS_SHAPE_SPHERE: return sphere_sample_towards(ld3(p), p[3], target.pos, u, v);
        case PBRS_SHAPE_DISK: {  // :258-269
            float cos_t, sin_t;
            concentric_sample_disk(u, v, cos_t, sin_t);
            f3 dn = ld3(p + 3), radial = ld3(p + 6);
            f3 radial2 = cross(dn, radial);
            f3 cp = radial * cos_t + radial2 * sin_t;
            return LightPoint{ld3(p) + cp, facing(dn, target.normal)};
        }
        case PBRS_SHAPE_TRIANGLE: {  // :277-290
            if (u + v > 1.0f) {
                float nu = 1.0f - v, nv = 1.0f - u;
                u = nu;
                v = nv;
            }
            f3 p0 = ld3(p), p1 = ld3(p + 3), p2 = ld3(p + 6);
            f3 position = p0 + (p1 - p0) * u + (p2 - p0) * v;
            f3 normal = hat(cross(p0 - p1, p2 - p1));
            return LightPoint{position, normal};
        }
        default: {  // :297-305
            f3 position = ld3(p) + u * ld3(p + 3) + v * ld3(p + 6);
            return LightPoint{position, cross(ld3(p + 3), ld3(p + 6))};
        }
    }
}
PD bool light_pdf_at(const pbrs_area_light& L, uint32_t kind, const Isect& reference, f3 wi, float& pdf) {
    if (kind == PBRS_SHAPE_SPHERE) return sphere_pdf_at(ld3(L.p), L.p[3], reference.pos, wi, pdf);
    f3 o, d;  // default impl :28-33 (Q4: distance, not distance squared)
    spawn_ray(reference, wi, o, d);
    LightPoint hit;
    if (!light_shape_intersect(L, kind, o, d, hit)) return false;
    pdf = norm(reference.pos - hit.pos) / (pn_abs(dot(hit.normal, -wi)) * L.area);
    return true;
}

struct ShadowRay {
    f3 o, d;
    float t_max;
};
PD ShadowRay limited_ray_to(const Isect& is, f3 pos) {  // spawn_limited_ray_to
    ShadowRay r;
    spawn_ray(is, pos - is.pos, r.o, r.d);
    r.t_max = 1.0f - 0.001f;
    return r;
}

// DiffuseAreaLight::sample_incident_radiance (light/src/lib.rs:158-172)
PD void area_sample_incident(const pbrs_area_light& L, uint32_t kind, const Isect& target, float u, float v, f3& li, f3& wi, float& pdf, ShadowRay& vis) {
    LightPoint pt = light_sample_towards(L, kind, target, u, v);
    wi = hat(pt.pos - target.pos);
    li = !pn_sign_negative(dot(pt.normal, -wi)) ? ld3(L.emit) : gray(0.0f);  // radiance_from :127-133
    if (!light_pdf_at(L, kind, target, wi, pdf)) pdf = 0.0f;
    vis = limited_ray_to(target, pt.pos);
}
// DiffuseAreaLight::radiance_to (:141-146)
PD bool area_radiance_to(const pbrs_area_light& L, uint32_t kind, const Isect& target, f3 wi, f3& le, float& pdf, ShadowRay& vis) {
    f3 o, d;
    spawn_ray(target, wi, o, d);
    LightPoint hit;
    if (!light_shape_intersect(L, kind, o, d, hit)) return false;
    if (kind == PBRS_SHAPE_SPHERE) {
        if (!sphere_pdf_at(ld3(L.p), L.p[3], target.pos, wi, pdf)) return false;
    } else {
        // the default pdf_at (sample_shape.rs:28-33) spawns the same ray from the same point and intersects the same
        // shape again: a pure function of the same operands, so its hit is `hit`
        pdf = norm(target.pos - hit.pos) / (pn_abs(dot(hit.normal, -wi)) * L.area);
    }
    vis = limited_ray_to(target, hit.pos);
    le = ld3(L.emit);
    return true;
}
// DeltaLight::sample_incident_radiance (:67-92); always Prob::Mass(1.0)
PD void delta_sample_incident(const pbrs_delta_light& L, const Isect& target, f3& li, f3& wi, ShadowRay& vis) {
    f3 v = ld3(L.v), color = ld3(L.color);
    if (L.kind == PBRS_DELTA_POINT) {
        li = color * pn_weak_recip(norm2(v - target.pos));
        wi = hat(v - target.pos);
        vis = limited_ray_to(target, v);
        return;
    }
    f3 outside_world = target.pos - L.world_radius * 2.0f * v;
    vis = limited_ray_to(target, outside_world);
    li = color;
    wi = -v;
}
