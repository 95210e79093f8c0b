// device/textures.h — texture/src/lib.rs:35-223 (Checker, Perlin marble, nearest-neighbour Image) and the environment
// light (scene/src/lib.rs:105-117 with the closures of scene/src/preset.rs:25-53), for one lane.
#pragma once
#include "shapes.h"

// Perlin::noise (:97-137): trilinear blend of lattice-vector dot products with a smoothstep on each axis
PD float perlin_noise(const DevScene& S, const pbrs_texture& t, f3 p) {
    const float fx = p.x * t.freq, fy = p.y * t.freq, fz = p.z * t.freq;
    const float flx = pn_floor(fx), fly = pn_floor(fy), flz = pn_floor(fz);
    const int i = (int)flx, j = (int)fly, k = (int)flz;  // `f.floor() as i32` (saturating; in range for any scene scale in use)
    float u = fx - flx, v = fy - fly, w = fz - flz;
    u = u * u * (3.0f - 2.0f * u);
    v = v * v * (3.0f - 2.0f * v);
    w = w * w * (3.0f - 2.0f * w);
    const float* rand_vec = S.tex_floats + t.data;
    const uint32_t* perm_x = S.tex_words + t.perm;
    const uint32_t *perm_y = perm_x + 256, *perm_z = perm_x + 512;
    float accum = 0.0f;
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj)
#pragma unroll
            for (int dk = 0; dk < 2; ++dk) {
                const uint32_t index = perm_x[(i + di) & 255] ^ perm_y[(j + dj) & 255] ^ perm_z[(k + dk) & 255];
                const f3 c = ld3(rand_vec + 3u * index);
                const f3 weight_v = mk3(u - (float)di, v - (float)dj, w - (float)dk);
                const float dot_product = dot(c, weight_v);
                accum += ((float)di * u + (float)(1 - di) * (1.0f - u)) * ((float)dj * v + (float)(1 - dj) * (1.0f - v)) *
                         ((float)dk * w + (float)(1 - dk) * (1.0f - w)) * dot_product;
            }
    return accum;
}
PD float perlin_turbulance(const DevScene& S, const pbrs_texture& t, f3 p) {  // :139-147
    float acc = 0.0f;
    for (int i = 0; i < 7; ++i) {
        const float scale = pn_powi(2.0f, i);
        acc = acc + pn_powi(0.5f, i) * perlin_noise(S, t, mk3(p.x * scale, p.y * scale, p.z * scale));
    }
    return pn_abs(acc);
}
// `(x) as usize` of Rust: saturating, NaN -> 0
PD uint64_t to_usize(float x) { return x != x ? 0ull : (x <= 0.0f ? 0ull : (x >= 1.8446744e19f ? ~0ull : (uint64_t)x)); }

PD f3 tex_image_value(const DevScene& S, const pbrs_texture& t, float u, float v) {  // Image :211-223
    const float uc = pn_clamp(u, 0.0f, 1.0f), vc = pn_clamp(v, 0.0f, 1.0f);
    const uint64_t col = to_usize(uc * (float)t.width) % t.width;
    const uint64_t row = to_usize(vc * (float)t.height) % t.height;
    return ld3(S.tex_floats + t.data + 3ull * (row * t.width + col));
}
PD f3 tex_value(const DevScene& S, uint32_t id, float u, float v, f3 p) {
    const pbrs_texture& t = S.textures[id];
    if (t.kind == PBRS_TEX_CHECKER) {  // :40-49
        const float sines = pn_sin(10.0f * p.x) * pn_sin(10.0f * p.y) * pn_sin(10.0f * p.z);
        return sines < 0.0f ? ld3(t.odd) : ld3(t.even);
    }
    if (t.kind == PBRS_TEX_PERLIN)  // :150-160, a marble-like texture
        return pn_mul_add(pn_sin(t.freq * p.z + 10.0f * perlin_turbulance(S, t, p)), 0.5f, 0.5f) * gray(1.0f);
    return tex_image_value(S, t, u, v);
}

// Scene::eval_env_light (scene/src/lib.rs:105-117)
PD f3 env_eval(const DevScene& S, f3 dir) {
    switch (S.env_kind) {
        case PBRS_ENV_IMAGE: {
            const float phi = pn_atan2(dir.z, dir.x);
            const float u = pn_fract(phi * PN_FRAC_1_PI * 0.5f + 1.0f);
            const float cos_theta = dir.y / norm(dir);
            const float v = pn_acos(cos_theta) / PN_PI;
            // pbrs_upload_scene only accepts an Image texture here: the kernels of untextured scenes stay free of the other kinds
            return cmul(tex_image_value(S, S.textures[S.env_texture], u, v), ld3(S.env_scale));
        }
        case PBRS_ENV_BLUE_SKY: {  // preset.rs:25-30
            const float y = (hat(dir).y + 1.0f) * 0.5f;
            return mk3(0.5f, 0.7f, 1.0f) * y + gray(1.0f) * (1.0f - y);
        }
        case PBRS_ENV_DARK_ROOM: {  // :32-37
            const float y = (hat(dir).y + 1.0f) * 0.5f;
            return gray(0.1f) * y + gray(0.1f) * (1.0f - y);
        }
        case PBRS_ENV_DUSK: {  // :39-52
            const f3 horizon = mk3(245.0f / 255.0f, 174.0f / 255.0f, 82.0f / 255.0f);
            const f3 dome = mk3(109.0f / 255.0f, 150.0f / 255.0f, 204.0f / 255.0f);
            const float tilt = pn_acos(hat(dir).y);
            if (tilt > PN_PI * 0.25f) return dome;
            if (tilt > 0.0f) {
                const float t = tilt / (PN_PI * 0.25f);
                return dome * t + horizon * (1.0f - t);
            }
            return gray(0.2f);
        }
        default: return ld3(S.env);
    }
}
