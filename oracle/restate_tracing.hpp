// oracle/restate_tracing.hpp — TEST INFRASTRUCTURE ONLY (the checker, never the product).
//
// CPU restatement of the reference's `tracing` crate (camera, sampling, lights, materials, the
// trace_ray bounce loop, image buffer, thread driver) for the render hot path.  Each function cites
// the reference file:line it follows.
//
// Random numbers: the reference draws from `rand 0.10.1` SmallRng (Xoshiro256++), OS-seeded
// (tracing/src/worker.rs:35) — not reproducible even by the reference, and `rand` is not vendored.
// PARITY UNPINNED at stream level.  Two sampler modes are provided:
//   SEQ     — one sequential Xoshiro256++ stream per thread, libm sin/cos: the reference's structure;
//   COUNTER — include/ptx_sampling.h (pure function of seed/pixel/sample/dimension, deterministic
//             sincos): traces exactly the paths the GPU kernels trace.
// Pinned by the reference's material tests (tracing/src/material.rs:201-370) via tests/.
#pragma once
#include <thread>

#include "../include/ptx_sampling.h"
#include "restate_kdtree.hpp"

namespace orc {

// ------------------------------------------------------------------ samplers
struct Xoshiro256pp {  // rand 0.10 SmallRng on 64-bit targets (recalled; parity unpinned)
    uint64_t s[4];
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    explicit Xoshiro256pp(uint64_t seed) {
        for (int i = 0; i < 4; ++i) {
            seed += 0x9E3779B97F4A7C15ull;
            uint64_t z = seed;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            s[i] = z ^ (z >> 31);
        }
    }
    uint64_t next_u64() {
        uint64_t result = rotl(s[0] + s[3], 23) + s[0];
        uint64_t t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return result;
    }
    uint32_t next_u32() { return (uint32_t)(next_u64() >> 32); }
};

struct Sampler {
    enum Mode { SEQ = 0, COUNTER = 1 } mode = SEQ;
    Xoshiro256pp* seq = nullptr;
    uint64_t key = 0;

    float u01(uint32_t dim) {  // rand: random::<f32>() = (u32 >> 8) * 2^-24
        if (mode == COUNTER) return ptx_u01(key, dim);
        return (float)(seq->next_u32() >> 8) * 5.9604644775390625e-8f;
    }
    float range_open(float low, float high) {  // rand UniformFloat::sample_single
        float scale = high - low;
        for (;;) {
            uint32_t bits = (seq->next_u32() >> 9) | 0x3F800000u;
            float v12;
            std::memcpy(&v12, &bits, 4);
            float res = (v12 - 1.0f) * scale + low;
            if (res < high) return res;
        }
    }
    float sym_open(uint32_t dim) {  // sampling.rs:10
        return mode == COUNTER ? ptx_sym_open(key, dim) : range_open(-1.0f, 1.0f);
    }
    float angle(uint32_t dim) {  // sampling.rs:11
        return mode == COUNTER ? ptx_angle(key, dim) : range_open(0.0f, 6.28318530717958647692f);
    }
    float sym_closed(uint32_t dim) {  // sampling.rs:28-29, rand sample_single_inclusive
        if (mode == COUNTER) return ptx_sym_closed(key, dim);
        uint32_t bits = (seq->next_u32() >> 9) | 0x3F800000u;
        float v12;
        std::memcpy(&v12, &bits, 4);
        float scale = 2.0f / (1.0f - 1.1920929e-7f);
        float res = (v12 - 1.0f) * scale + -1.0f;
        return res > 1.0f ? 1.0f : res;
    }
    void sincos(float x, float& s, float& c) {
        if (mode == COUNTER) {
            ptx_sincos(x, &s, &c);
        } else {
            s = std::sin(x);
            c = std::cos(x);
        }
    }
};

// ------------------------------------------------------------------ sampling.rs
// sampling.rs:9-17
inline Vec3 uniform_sample_unit_sphere(Sampler& rng, uint32_t dim) {
    float z = rng.sym_open(dim);
    float a = rng.angle(dim + 1);
    float r = std::sqrt(1.0f - z * z);
    float a_sin, a_cos;
    rng.sincos(a, a_sin, a_cos);
    return {r * a_cos, r * a_sin, z};
}
// sampling.rs:27-42
inline Vec2 concentric_sample_unit_disk(Sampler& rng, uint32_t dim) {
    float x = rng.sym_closed(dim);
    float y = rng.sym_closed(dim + 1);
    if (x == 0.0f && y == 0.0f) return {0.0f, 0.0f};
    float r, theta;
    if (x >= -y && x > y) {
        r = x;
        theta = y / x;
    } else if (x >= -y) {
        r = y;
        theta = 2.0f - x / y;
    } else if (x <= y) {
        r = -x;
        theta = 4.0f + y / x;
    } else {
        r = -y;
        theta = 6.0f - x / y;
    }
    float s, c;
    rng.sincos(theta * 0.785398163397448309616f, s, c);
    return {r * s, r * c};  // Vec2::from((sin, cos)), sampling.rs:41
}
// sampling.rs:44-48
inline Vec3 cosine_sample_hemisphere(Sampler& rng, uint32_t dim) {
    Vec2 ret = concentric_sample_unit_disk(rng, dim);
    float z = std::sqrt(std::fmax(0.0f, 1.0f - ret.x * ret.x - ret.y * ret.y));
    return {ret.x, ret.y, z};
}

// ------------------------------------------------------------------ camera.rs
struct Pinhole {
    Vec3 position, plane, dx, dy;
    uint32_t width, height;
    // camera.rs:15-24 + 71-85
    static Pinhole make(Vec3 position, Vec3 target, Vec3 up_in, float fov_degrees, uint32_t w,
                        uint32_t h) {
        Vec3 direction = normalize(target - position);
        Vec3 up = normalize(up_in);
        Vec3 right = normalize(cross(direction, up_in));
        float aspect_ratio = (float)w / (float)h;
        float half_fov_radians = fov_degrees * 3.14159265358979323846f / 360.0f;
        Vec3 x = right * (std::sin(half_fov_radians) * aspect_ratio);
        Vec3 y = up * std::sin(half_fov_radians);
        Vec3 z = direction * std::cos(half_fov_radians);
        return {position, z + y - x, 2.0f * x, -2.0f * y, w, h};
    }
    // camera.rs:88-93
    Ray ray(Vec2 v) const { return {position, plane + v.x * dx + v.y * dy}; }
};

// ------------------------------------------------------------------ light.rs
struct Light {  // every OBJ-scene light is a SphericalLight (light.rs:79-89)
    Vec3 center, intensity;
    float radius;
    // light.rs:17-19 via :64
    Vec3 emitted(Vec3 point) const { return intensity / length_squared(center - point); }
    // light.rs:34-37, 69-76: returns the shadow ray; t_range is always 0..=1
    Ray sample_shadow_ray(Vec3 point, Sampler& rng, uint32_t dim) const {
        Vec3 target = center + uniform_sample_unit_sphere(rng, dim) * radius;
        return Ray::between(point, target);
    }
};

// ------------------------------------------------------------------ material.rs
struct Texture {
    uint32_t width = 0, height = 0;
    std::vector<float> rgb;  // linear-light f32, row 0 = top of the PNG
};
struct Material {
    int texture = -1;  // AlbedoSource::Texture index, or -1 = AlbedoSource::Color
    Vec3 color{0, 0, 0};
    Vec3 schlick_f0{0, 0, 0};
    float transmission = 0.0f;
    float ior = 1.0f;
};
struct Surface {
    Vec3 wi, n;
    Vec2 uv;
};
struct BsdfSample {
    bool is_delta;
    float pdf;
    Vec3 bsdf, wo;
};

inline float luminance(Vec3 c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; }  // :12-15
inline Vec3 reflect(Vec3 v, Vec3 n) { return v - 2.0f * dot(v, n) * n; }                    // :17-19
inline Vec3 perpendicular(Vec3 v) {                                                         // :21-32
    float vx = std::fabs(v.x), vy = std::fabs(v.y), vz = std::fabs(v.z);
    if (vx < vy && vx < vz) return {0.0f, -v.z, v.y};
    if (vy < vz) return {-v.z, 0.0f, v.x};
    return {-v.y, v.x, 0.0f};
}
inline Vec3 schlicks_approximation(Vec3 r0, Vec3 wi, Vec3 n) {  // :34-38
    float cos_theta = std::fmax(dot(wi, n), 0.0f);
    float b = 1.0f - cos_theta;
    float b2 = b * b;
    float t = b * (b2 * b2);  // powi(5): b * (b^2)^2 (llvm.powi / __powisf2 square-and-multiply)
    return r0 + (1.0f - r0) * t;
}
inline BsdfSample bsdf_zero(Vec3 n) { return {true, 0.0f, {0, 0, 0}, n}; }  // :55-64
inline BsdfSample sample_specular(const Surface& s, Vec3 color, float probability) {  // :74-82
    Vec3 wo = normalize(reflect(-s.wi, s.n));
    return {true, probability, color, wo};
}
inline BsdfSample sample_refraction(const Surface& s, float ior, Vec3 diffuse,
                                    Vec3 transmitted_diffuse, float probability) {  // :84-112
    bool is_entering = dot(s.wi, s.n) < 0.0f;
    float n1 = is_entering ? 1.0f : ior;
    float n2 = is_entering ? ior : 1.0f;
    float eta = n1 / n2;
    Vec3 normal = is_entering ? s.n : -s.n;
    Vec3 incoming = -s.wi;
    float cos_theta_i = dot(incoming, normal);
    float sin2_theta_t = eta * eta * (1.0f - cos_theta_i * cos_theta_i);
    if (sin2_theta_t >= 1.0f) return sample_specular(s, diffuse, probability);
    float cos_theta_t = std::sqrt(1.0f - sin2_theta_t);
    Vec3 wo = normalize(eta * incoming + (eta * cos_theta_i - cos_theta_t) * normal);
    float eta_scale = (n2 * n2) / (n1 * n1);
    return {true, probability, transmitted_diffuse * eta_scale, wo};
}
constexpr float FRAC_1_PI = 0.318309886183790671538f;
inline BsdfSample sample_diffuse(const Surface& s, Sampler& rng, uint32_t dim,
                                 Vec3 transmitted_diffuse, float probability) {  // :114-136
    Vec3 tangent = normalize(perpendicular(s.n));
    Vec3 bitangent = cross(s.n, tangent);
    Vec3 h = cosine_sample_hemisphere(rng, dim);
    Vec3 wo = normalize(h.x * tangent + h.y * bitangent + h.z * s.n);
    float cos_theta = std::fmax(dot(wo, s.n), 0.0f);
    return {false, probability * cos_theta * FRAC_1_PI, transmitted_diffuse * FRAC_1_PI, wo};
}
// material/albedo.rs:10-28
inline float wrap01(float x) {
    float y = x - std::floor(x);
    return y == 1.0f ? 0.0f : y;
}
inline Vec3 albedo_get(const Material& m, const std::vector<Texture>& textures, Vec2 uv) {
    if (m.texture < 0) return m.color;
    const Texture& t = textures[m.texture];
    float px = std::floor((float)t.width * wrap01(uv.x));
    float py = std::floor((float)t.height * wrap01(uv.y));
    size_t o = ((size_t)(uint32_t)py * t.width + (uint32_t)px) * 3;
    return {t.rgb[o], t.rgb[o + 1], t.rgb[o + 2]};
}
// material.rs:167-191.  `dim` = dimension of the lobe pick; the diffuse lobe uses dim+1, dim+2.
inline BsdfSample material_sample(const Material& m, const std::vector<Texture>& textures,
                                  const Surface& s, Sampler& rng, uint32_t dim) {
    Vec3 f = schlicks_approximation(m.schlick_f0, s.wi, s.n);
    Vec3 diffuse = albedo_get(m, textures, s.uv);
    Vec3 transmitted_diffuse = (1.0f - f) * diffuse;
    float specular_strength = luminance(f);
    float diffuse_strength = luminance(transmitted_diffuse) * (1.0f - m.transmission);
    float refraction_strength = luminance(transmitted_diffuse) * m.transmission;
    float total_strength = specular_strength + diffuse_strength + refraction_strength;
    if (total_strength <= 0.0f) return bsdf_zero(s.n);
    float p_specular = specular_strength / total_strength;
    float p_diffuse = diffuse_strength / total_strength;
    float p_refraction = refraction_strength / total_strength;
    float r = rng.u01(dim);
    if (p_specular > 0.0f && r < p_specular) return sample_specular(s, f, p_specular);
    if (r < p_specular + p_refraction)
        return sample_refraction(s, m.ior, f, transmitted_diffuse, p_refraction);
    if (p_diffuse > 0.0f) return sample_diffuse(s, rng, dim + 1, transmitted_diffuse, p_diffuse);
    return bsdf_zero(s.n);
}
// material.rs:153-164 (lerp: glam `self * (1.0 - s) + rhs * s`)
inline Material material_from_mtl(Vec3 kd, Vec3 ks, float ni, float pm, float transparency,
                                  int texture) {
    Material m;
    m.texture = texture;
    m.color = kd;
    float q = (ni - 1.0f) / (ni + 1.0f);
    float f0d = q * q;
    Vec3 a{f0d, f0d, f0d};
    m.schlick_f0 = a * (1.0f - pm) + ks * pm;
    m.transmission = transparency;
    m.ior = ni;
    return m;
}

// ------------------------------------------------------------------ properties.rs
struct TriangleProperties {  // properties.rs:10-14
    uint32_t material;
    Vec3 n0, n1, n2;
    Vec2 uv0, uv1, uv2;
    // properties.rs:24-26 → geometry/src/triangle.rs:144-146
    Vec3 compute_normal(float u, float v) const {
        return normalize((1.0f - (u + v)) * n0 + u * n1 + v * n2);
    }
    // properties.rs:29-31 → geometry/src/triangle.rs:158-160
    Vec2 compute_texcoord(float u, float v) const { return (1.0f - (u + v)) * uv0 + u * uv1 + v * uv2; }
};

// ------------------------------------------------------------------ scene + pathtracer.rs
struct Scene {
    std::vector<AnyTriangle> triangles;
    std::vector<TriangleProperties> properties;
    std::vector<Material> materials;
    std::vector<Texture> textures;
    std::vector<Light> lights;
    std::unique_ptr<KdNode> kdtree;
    Vec3 environment{0.8f, 0.8f, 0.8f};  // pathtracer-cli/src/main.rs:155
    uint32_t max_bounces = 10;
};

struct RayStats {
    uint64_t extend_rays = 0, shadow_rays = 0;
    TraversalCounters extend, shadow;
};

// Optional recorder for every ray the integrator shoots (kdtree-tester-cli methodology).
struct RayRecord {
    Ray ray;
    float t_lo, t_hi;
    std::optional<IndexedIntersection> hit;
};

// pathtracer.rs:25-111.  In COUNTER mode dimensions follow include/ptx_sampling.h.
inline Vec3 trace_ray(const Scene& sc, Sampler& rng, Ray ray, RayStats* stats = nullptr,
                      std::vector<RayRecord>* rec = nullptr) {
    Vec3 accumulated_radiance{0, 0, 0};
    Vec3 accumulated_transport{1, 1, 1};
    const uint32_t n_lights = (uint32_t)sc.lights.size();
    for (uint32_t bounce = 1; bounce <= sc.max_bounces; ++bounce) {
        const uint32_t base = ptx_dim_base(bounce, n_lights);
        auto intersection =
            kd_intersect(sc.kdtree.get(), sc.triangles, ray, 0.0f, std::numeric_limits<float>::max(),
                         stats ? &stats->extend : nullptr);
        if (stats) ++stats->extend_rays;
        if (rec) rec->push_back({ray, 0.0f, std::numeric_limits<float>::max(), intersection});
        if (!intersection) return accumulated_radiance + accumulated_transport * sc.environment;

        Vec3 wi = -ray.direction;
        const TriangleProperties& prop = sc.properties[intersection->index];
        Vec3 n = prop.compute_normal(intersection->inner.u, intersection->inner.v);
        Vec2 uv = prop.compute_texcoord(intersection->inner.u, intersection->inner.v);
        const Material& material = sc.materials[prop.material];

        Vec3 offset = 0.00001f * n;
        Vec3 point = ray.param(intersection->inner.t);
        Vec3 point_above = point + offset;
        Vec3 point_below = point - offset;
        Surface surface{wi, n, uv};

        BsdfSample sample = material_sample(material, sc.textures, surface, rng, base);

        Vec3 incoming_radiance{0, 0, 0};
        if (!sample.is_delta) {
            for (uint32_t l = 0; l < n_lights; ++l) {
                const Light& light = sc.lights[l];
                Ray shadow_ray = light.sample_shadow_ray(point_above, rng, base + 3 + 2 * l);
                auto shadow_hit = kd_intersect(sc.kdtree.get(), sc.triangles, shadow_ray, 0.0f, 1.0f,
                                               stats ? &stats->shadow : nullptr);
                if (stats) ++stats->shadow_rays;
                if (rec) rec->push_back({shadow_ray, 0.0f, 1.0f, shadow_hit});
                Vec3 contribution{0, 0, 0};
                if (!shadow_hit) {
                    Vec3 radiance = light.emitted(point);
                    Vec3 wo = normalize(shadow_ray.direction);
                    contribution = sample.bsdf * radiance * std::fabs(dot(wo, surface.n));
                }
                incoming_radiance = incoming_radiance + contribution;  // Iterator::sum fold from ZERO
            }
        }
        accumulated_radiance = accumulated_radiance + accumulated_transport * incoming_radiance;

        if (sample.pdf == 0.0f) return accumulated_radiance;

        float cosine_term = dot(sample.wo, surface.n);
        if (sample.is_delta) {
            accumulated_transport = accumulated_transport * sample.bsdf;
        } else {
            accumulated_transport =
                accumulated_transport * (sample.bsdf * (std::fabs(cosine_term) / sample.pdf));
        }
        if (length_squared(accumulated_transport) <= 1.0e-4f) return accumulated_radiance;

        Vec3 next_start_point = cosine_term >= 0.0f ? point_above : point_below;
        ray = Ray{next_start_point, sample.wo};
    }
    return accumulated_radiance;
}

// pathtracer.rs:113-117
inline Ray sample_ray_for_pixel(const Pinhole& pinhole, Sampler& rng, uint32_t px, uint32_t py) {
    Vec2 pixel_center = Vec2{(float)px, (float)py} + Vec2{rng.u01(0), rng.u01(1)};
    return pinhole.ray(pixel_center / Vec2{(float)pinhole.width, (float)pinhole.height});
}

// pathtracer.rs:131-141 — one full-frame iteration accumulated INTO buffer (row-major, += per pixel).
inline void render_mut(const Scene& sc, const Pinhole& pinhole, Sampler& rng, uint64_t seed,
                       uint32_t sample_index, float* buffer, RayStats* stats = nullptr) {
    for (uint32_t y = 0; y < pinhole.height; ++y) {
        for (uint32_t x = 0; x < pinhole.width; ++x) {
            uint32_t pixel = y * pinhole.width + x;
            if (rng.mode == Sampler::COUNTER) rng.key = ptx_path_key(seed, pixel, sample_index);
            Ray ray = sample_ray_for_pixel(pinhole, rng, x, y);
            Vec3 v = trace_ray(sc, rng, ray, stats);
            buffer[3 * pixel + 0] += v.x;
            buffer[3 * pixel + 1] += v.y;
            buffer[3 * pixel + 2] += v.z;
        }
    }
}

// worker.rs:27-50, 84-119: T threads x N iterations each, private sampler + private buffer, buffers
// summed in thread order.  Thread i renders global samples first + i*N .. first + (i+1)*N - 1.
inline void render_parallel_iterations(const Scene& sc, const Pinhole& pinhole, uint32_t threads,
                                       uint32_t iterations_per_thread, int mode, uint64_t seed,
                                       uint32_t first_sample, float* accum /* += */,
                                       RayStats* stats_out = nullptr) {
    size_t n = (size_t)pinhole.width * pinhole.height * 3;
    std::vector<std::vector<float>> buffers(threads, std::vector<float>(n, 0.0f));
    std::vector<RayStats> stats(threads);
    std::vector<std::thread> pool;
    for (uint32_t i = 0; i < threads; ++i) {
        pool.emplace_back([&, i] {
            Xoshiro256pp seq(ptx_mix64(seed + i));
            Sampler rng;
            rng.mode = (Sampler::Mode)mode;
            rng.seq = &seq;
            for (uint32_t it = 0; it < iterations_per_thread; ++it)
                render_mut(sc, pinhole, rng, seed, first_sample + i * iterations_per_thread + it,
                           buffers[i].data(), stats_out ? &stats[i] : nullptr);
        });
    }
    for (auto& t : pool) t.join();
    std::vector<float>& total = buffers[0];  // buffers.reduce(Add::add), worker.rs:106-107
    for (uint32_t i = 1; i < threads; ++i)
        for (size_t k = 0; k < n; ++k) total[k] = total[k] + buffers[i][k];
    for (size_t k = 0; k < n; ++k) accum[k] += total[k];
    if (stats_out) {
        for (auto& s : stats) {
            stats_out->extend_rays += s.extend_rays;
            stats_out->shadow_rays += s.shadow_rays;
            stats_out->extend.inner_nodes += s.extend.inner_nodes;
            stats_out->extend.leaves += s.extend.leaves;
            stats_out->extend.tri_tests += s.extend.tri_tests;
            stats_out->shadow.inner_nodes += s.shadow.inner_nodes;
            stats_out->shadow.leaves += s.shadow.leaves;
            stats_out->shadow.tri_tests += s.shadow.tri_tests;
        }
    }
}

// image_buffer.rs:11-13, 29-38
inline void to_rgb8(const float* pixels, size_t n_pixels, uint16_t iterations, uint8_t* out) {
    float iterations_inv = 1.0f / (float)iterations;
    auto conv = [&](float p) -> uint8_t {
        float g = std::pow(p * iterations_inv, 1.0f / 2.2f);
        g = g < 1.0f ? g : 1.0f;
        float c = std::round(g * 255.0f);
        if (!(c == c)) return 0;  // `as u8` saturating cast: NaN -> 0
        if (c <= 0.0f) return 0;
        if (c >= 255.0f) return 255;
        return (uint8_t)c;
    };
    for (size_t i = 0; i < n_pixels * 3; ++i) out[i] = conv(pixels[i]);
}

}  // namespace orc
