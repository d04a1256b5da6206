// rt_oracle.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of andros21/rustracer's per-pixel render loop (reference crate v1.0.4).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
// the product (librustracer_amd.so) never links, loads or calls anything from oracle/.
//
// The Rust reference cannot be built in this environment (no cargo/rustc, no network), so this
// file restates its algorithm literally — recursive solve, linear shape scan, a full HitRecord
// for every candidate shape, the serial all_rays prepass — one function per reference function,
// each citing the reference file:line it follows (paths relative to /root/reference).
//
// PARITY PIN: tests/test_oracle_kat.py checks this restatement against every known-answer test
// the reference's own #[cfg(test)] modules hold for the path: exact PCG u32/f32
// vectors (random.rs:56-81), sphere/plane/world hits (shape.rs:223-568, world.rs:62-92),
// flat/onoff/furnace/background renderers (render.rs:221-313), pigments and BRDF scatter
// (material.rs:324-405), camera corner rays (camera.rs:169-202), image tracer orientation and
// coverage (imagetracer.rs:137-183), ONB orthonormality (normal.rs:235-253).
// UNPINNED: the last ulp of sin/cos/atan2/acos — the reference calls the platform libm, which
// Cargo.lock does not pin (SURVEY.md §8c iii).  math_mode 0 below calls this box's glibc (what a
// gnu-target reference binary would do here); math_mode 1 uses include/rt_strict_math.h, the
// bit-reproducible definition the GPU kernels also use.  tests compare the two modes and report
// the outlier-pixel fraction.
//
// Build: see oracle/Makefile  (g++ -O2 -ffp-contract=off -fno-fast-math, no -march flags).

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../include/rt_strict_math.h"
#include "../include/rustracer_amd.h"

namespace {

// ---------------------------------------------------------------------------------------------
// transcendental back-ends
// ---------------------------------------------------------------------------------------------
struct MathFns {
    float (*sin)(float);
    float (*cos)(float);
    float (*atan2)(float, float);
    float (*acos)(float);
};
float libm_sin(float x) { return ::sinf(x); }
float libm_cos(float x) { return ::cosf(x); }
float libm_atan2(float y, float x) { return ::atan2f(y, x); }
float libm_acos(float x) { return ::acosf(x); }
float strict_sin(float x) { return rt_sinf(x); }
float strict_cos(float x) { return rt_cosf(x); }
float strict_atan2(float y, float x) { return rt_atan2f(y, x); }
float strict_acos(float x) { return rt_acosf(x); }
const MathFns kLibm = {libm_sin, libm_cos, libm_atan2, libm_acos};
const MathFns kStrict = {strict_sin, strict_cos, strict_atan2, strict_acos};
const MathFns& math_for(int mode) { return mode == 0 ? kLibm : kStrict; }

const float PI = 3.14159265358979323846f;  // std::f32::consts::PI

// ---------------------------------------------------------------------------------------------
// value types: vector.rs, point.rs, normal.rs, color.rs, misc.rs
// ---------------------------------------------------------------------------------------------
struct Vector { float x, y, z; };
struct Point { float x, y, z; };
struct Normal { float x, y, z; };
struct Color { float r, g, b; };
struct Vector2D { float u, v; };

// vector.rs:53-55
float dot(Vector a, Vector b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// vector.rs:60-67
float squared_norm(Vector a) { return dot(a, a); }
float norm(Vector a) { return std::sqrt(squared_norm(a)); }
// vector.rs:157-166
Vector mul(Vector a, float s) { return {a.x * s, a.y * s, a.z * s}; }
// vector.rs:72-75
Vector normalize(Vector a) { return mul(a, 1.0f / norm(a)); }
// vector.rs:121-143
Vector add(Vector a, Vector b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
Vector sub(Vector a, Vector b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
// point.rs:44-53
Point add(Point a, Vector b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
// color.rs:49-83
Color add(Color a, Color b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
Color mul(Color a, Color b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }
Color mul(Color a, float s) { return {a.r * s, a.g * s, a.b * s}; }

// ---------------------------------------------------------------------------------------------
// random.rs
// ---------------------------------------------------------------------------------------------
struct Pcg {
    uint64_t state, inc;
    // random.rs:26-34
    Pcg(uint64_t init_state, uint64_t init_seq) {
        state = 0;
        inc = (init_seq << 1) | 1u;
        random();
        state += init_state;  // wrapping (release build semantics)
        random();
    }
    // random.rs:37-43
    uint32_t random() {
        const uint64_t old_state = state;
        state = old_state * 6364136223846793005ULL + inc;
        const uint32_t xor_shifted = (uint32_t)(((old_state >> 18) ^ old_state) >> 27);
        const uint32_t rot = (uint32_t)(old_state >> 59);
        return (xor_shifted >> rot) | (xor_shifted << ((0u - rot) & 31u));
    }
    // random.rs:46-48  (u32::MAX as f32 == 2^32)
    float random_float() { return (float)random() / 4294967296.0f; }
};

// ---------------------------------------------------------------------------------------------
// ray.rs
// ---------------------------------------------------------------------------------------------
struct Ray {
    Point origin;
    Vector dir;
    float tmin, tmax;
    uint32_t depth;
    // ray.rs:26-28
    Point at(float t) const { return add(origin, mul(dir, t)); }
};
// ray.rs:46-54
Ray default_ray() {
    return {{0.f, 0.f, 0.f}, {1.f, 0.f, 0.f}, 1e-5f, std::numeric_limits<float>::infinity(), 0};
}

// ---------------------------------------------------------------------------------------------
// transformation.rs
// ---------------------------------------------------------------------------------------------
struct Matrix { float e[4][4]; };
struct Transformation {
    Matrix m, invm;
    // transformation.rs:106-111
    Transformation inverse() const { return {invm, m}; }
};

// transformation.rs:133-143
Vector apply(const Transformation& t, Vector v) {
    const auto& m = t.m.e;
    return {m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z,
            m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z,
            m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z};
}
// transformation.rs:145-161  (inverse transposed)
Normal apply(const Transformation& t, Normal n) {
    const auto& i = t.invm.e;
    return {i[0][0] * n.x + i[1][0] * n.y + i[2][0] * n.z,
            i[0][1] * n.x + i[1][1] * n.y + i[2][1] * n.z,
            i[0][2] * n.x + i[1][2] * n.y + i[2][2] * n.z};
}
// transformation.rs:163-195
Point apply(const Transformation& t, Point p) {
    const auto& m = t.m.e;
    Point np = {m[0][0] * p.x + m[0][1] * p.y + m[0][2] * p.z + m[0][3],
                m[1][0] * p.x + m[1][1] * p.y + m[1][2] * p.z + m[1][3],
                m[2][0] * p.x + m[2][1] * p.y + m[2][2] * p.z + m[2][3]};
    const float w = p.x * m[3][0] + p.y * m[3][1] + p.z * m[3][2] + m[3][3];
    if (w == 1.0f) return np;
    return {np.x / w, np.y / w, np.z / w};
}
// transformation.rs:197-209
Ray apply(const Transformation& t, const Ray& r) {
    return {apply(t, r.origin), apply(t, r.dir), r.tmin, r.tmax, r.depth};
}

Transformation from_flat(const float* m, const float* invm) {
    Transformation t;
    std::memcpy(t.m.e, m, sizeof(float) * 16);
    std::memcpy(t.invm.e, invm, sizeof(float) * 16);
    return t;
}

// ---------------------------------------------------------------------------------------------
// normal.rs:143-158
// ---------------------------------------------------------------------------------------------
void create_onb_from_z(Normal normal, Vector* e1, Vector* e2, Vector* e3) {
    const float sign = std::copysign(1.0f, normal.z);
    const float a = -1.0f / (sign + normal.z);
    const float b = normal.x * normal.y * a;
    *e1 = {1.0f + sign * normal.x * normal.x * a, sign * b, -sign * normal.x};
    *e2 = {b, sign + normal.y * normal.y * a, -normal.y};
    *e3 = {normal.x, normal.y, normal.z};
}

// ---------------------------------------------------------------------------------------------
// material.rs
// ---------------------------------------------------------------------------------------------
struct Texture { uint32_t w, h; const float* rgb; };

struct Pigment {
    uint32_t kind;
    Color c1, c2;
    uint32_t steps;
    const Texture* tex;
    // material.rs:106-115
    Color get_color(Vector2D uv) const {
        switch (kind) {
            case RT_PIGMENT_UNIFORM:  // material.rs:32-36
                return c1;
            case RT_PIGMENT_IMAGE: {  // material.rs:54-70
                uint32_t col = rt_f32_to_u32_sat(uv.u * (float)tex->w);
                uint32_t row = rt_f32_to_u32_sat(uv.v * (float)tex->h);
                if (col >= tex->w) col = tex->w - 1;
                if (row >= tex->h) row = tex->h - 1;
                const float* p = tex->rgb + 3 * ((size_t)row * tex->w + col);  // hdrimage.rs:46-48
                return {p[0], p[1], p[2]};
            }
            default: {  // material.rs:86-96
                const uint32_t int_u = rt_f32_to_u32_sat(std::floor(uv.u * (float)steps));
                const uint32_t int_v = rt_f32_to_u32_sat(std::floor(uv.v * (float)steps));
                return (int_u % 2 == int_v % 2) ? c1 : c2;
            }
        }
    }
};

struct Material {
    uint32_t brdf;  // RT_BRDF_*
    Pigment brdf_pigment, emitted_radiance;
};

// material.rs:161-186
Ray diffuse_scatter_ray(const MathFns& mf, float pcg0, float pcg1, Point interaction_point,
                        Normal normal, uint32_t depth) {
    Vector e1, e2, e3;
    create_onb_from_z(normal, &e1, &e2, &e3);
    const float cos_theta_sq = pcg0;
    const float cos_theta = std::sqrt(cos_theta_sq);
    const float sin_theta = std::sqrt(1.0f - cos_theta_sq);
    const float phi = 2.0f * PI * pcg1;
    const Vector dir = add(add(mul(mul(e1, mf.cos(phi)), cos_theta), mul(mul(e2, mf.sin(phi)), cos_theta)),
                           mul(e3, sin_theta));
    return {interaction_point, dir, 1.0e-3f, std::numeric_limits<float>::infinity(), depth};
}

// material.rs:225-246
Ray specular_scatter_ray(Vector incoming_dir, Point interaction_point, Normal normal, uint32_t depth) {
    const Vector ray_dir = normalize(incoming_dir);
    const Vector n = normalize(Vector{normal.x, normal.y, normal.z});
    const float dot_prod = dot(n, ray_dir);
    const Vector dir = sub(ray_dir, mul(mul(n, 2.0f), dot_prod));
    return {interaction_point, dir, 1e-5f, std::numeric_limits<float>::infinity(), depth};
}

// ---------------------------------------------------------------------------------------------
// shape.rs
// ---------------------------------------------------------------------------------------------
struct HitRecord {
    Point world_point;
    Normal normal;
    Vector2D surface_point;
    float t;
    Ray ray;
    const Material* material;
    uint32_t shape_index;  // bookkeeping only (for rt_hit.shape)
};

struct Counters {
    uint64_t rays = 0, samples = 0, sphere_tests = 0, plane_tests = 0, diffuse_scatters = 0,
             specular_scatters = 0;
    void operator+=(const Counters& o) {
        rays += o.rays; samples += o.samples; sphere_tests += o.sphere_tests;
        plane_tests += o.plane_tests; diffuse_scatters += o.diffuse_scatters;
        specular_scatters += o.specular_scatters;
    }
};

struct Shape {
    uint32_t kind;
    Transformation transformation;
    const Material* material;
    uint32_t index;
};

// shape.rs:88-95
Normal sphere_normal(Point point, Vector ray_dir) {
    const Normal result = {point.x, point.y, point.z};
    if (dot(Vector{point.x, point.y, point.z}, ray_dir) < 0.0f) return result;
    return {-result.x, -result.y, -result.z};
}
// shape.rs:101-108
Vector2D sphere_point_to_uv(const MathFns& mf, Point point) {
    float u = mf.atan2(point.y, point.x) / (2.0f * PI);
    const float v = mf.acos(point.z) / PI;
    if (u < 0.0f) u += 1.0f;
    return {u, v};
}
// shape.rs:112-143
bool sphere_ray_intersection(const MathFns& mf, const Shape& s, const Ray& ray, HitRecord* out) {
    const Ray inv_ray = apply(s.transformation.inverse(), ray);
    const Vector origin_vec = {inv_ray.origin.x, inv_ray.origin.y, inv_ray.origin.z};
    const float a = squared_norm(inv_ray.dir);
    const float b = 2.0f * dot(origin_vec, inv_ray.dir);
    const float c = squared_norm(origin_vec) - 1.0f;
    const float delta = b * b - 4.0f * a * c;
    if (delta <= 0.0f) return false;
    const float sqrt_delta = std::sqrt(delta);
    const float t1 = (-b - sqrt_delta) / (2.0f * a);
    const float t2 = (-b + sqrt_delta) / (2.0f * a);
    float first_hit_t;
    if ((t1 > inv_ray.tmin) && (t1 < inv_ray.tmax)) first_hit_t = t1;
    else if ((t2 > inv_ray.tmin) && (t2 < inv_ray.tmax)) first_hit_t = t2;
    else return false;
    const Point hit_point = inv_ray.at(first_hit_t);
    out->world_point = apply(s.transformation, hit_point);
    out->normal = apply(s.transformation, sphere_normal(hit_point, inv_ray.dir));
    out->surface_point = sphere_point_to_uv(mf, hit_point);
    out->t = first_hit_t;
    out->ray = ray;
    out->material = s.material;
    out->shape_index = s.index;
    return true;
}
// shape.rs:175-182
Normal plane_normal(Vector ray_dir) {
    const Normal normal = {0.f, 0.f, 1.f};
    if (dot(Vector{normal.x, normal.y, normal.z}, ray_dir) < 0.0f) return normal;
    return {-normal.x, -normal.y, -normal.z};
}
// shape.rs:187-192
Vector2D plane_point_to_uv(Point point) {
    return {point.x - std::floor(point.x), point.y - std::floor(point.y)};
}
// shape.rs:196-214
bool plane_ray_intersection(const Shape& s, const Ray& ray, HitRecord* out) {
    const Ray inv_ray = apply(s.transformation.inverse(), ray);
    if (std::fabs(inv_ray.dir.z) < 1e-5f) return false;
    const float t = -inv_ray.origin.z / inv_ray.dir.z;
    if ((t <= inv_ray.tmin) || (t >= inv_ray.tmax)) return false;
    const Point hit_point = inv_ray.at(t);
    out->world_point = apply(s.transformation, hit_point);
    out->normal = apply(s.transformation, plane_normal(inv_ray.dir));
    out->surface_point = plane_point_to_uv(hit_point);
    out->t = t;
    out->ray = ray;
    out->material = s.material;
    out->shape_index = s.index;
    return true;
}

// ---------------------------------------------------------------------------------------------
// world.rs:32-50
// ---------------------------------------------------------------------------------------------
struct World {
    std::vector<Shape> shapes;
    std::vector<Material> materials;
    std::vector<Texture> textures;
    const MathFns* mf;

    bool ray_intersection(const Ray& ray, HitRecord* closest, Counters* cnt) const {
        cnt->rays++;
        bool have = false;
        for (const Shape& shape : shapes) {
            HitRecord shape_hit;
            bool hit;
            if (shape.kind == RT_SHAPE_SPHERE) {
                cnt->sphere_tests++;
                hit = sphere_ray_intersection(*mf, shape, ray, &shape_hit);
            } else {
                cnt->plane_tests++;
                hit = plane_ray_intersection(shape, ray, &shape_hit);
            }
            if (!hit) continue;
            if (!have) { *closest = shape_hit; have = true; }
            else if (shape_hit.t < closest->t) { *closest = shape_hit; }
        }
        return have;
    }
};

bool build_world(World* w, const rt_shape* shapes, uint32_t n_shapes, const rt_material* materials,
                 uint32_t n_materials, const rt_texture* textures, uint32_t n_textures, int math_mode) {
    w->mf = &math_for(math_mode);
    w->textures.resize(n_textures);
    for (uint32_t i = 0; i < n_textures; ++i) w->textures[i] = {textures[i].w, textures[i].h, textures[i].rgb};
    auto conv = [&](const rt_pigment& p, Pigment* out) -> bool {
        out->kind = p.kind;
        out->c1 = {p.c1[0], p.c1[1], p.c1[2]};
        out->c2 = {p.c2[0], p.c2[1], p.c2[2]};
        out->steps = p.steps;
        out->tex = nullptr;
        if (p.kind == RT_PIGMENT_IMAGE) {
            if (p.texture >= n_textures) return false;
            out->tex = &w->textures[p.texture];
        }
        return p.kind <= RT_PIGMENT_IMAGE;
    };
    w->materials.resize(n_materials);
    for (uint32_t i = 0; i < n_materials; ++i) {
        w->materials[i].brdf = materials[i].brdf;
        if (!conv(materials[i].brdf_pigment, &w->materials[i].brdf_pigment)) return false;
        if (!conv(materials[i].emitted, &w->materials[i].emitted_radiance)) return false;
    }
    w->shapes.resize(n_shapes);
    for (uint32_t i = 0; i < n_shapes; ++i) {
        if (shapes[i].material >= n_materials || shapes[i].kind > RT_SHAPE_PLANE) return false;
        w->shapes[i] = {shapes[i].kind, from_flat(shapes[i].m, shapes[i].invm),
                        &w->materials[shapes[i].material], i};
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// render.rs
// ---------------------------------------------------------------------------------------------
struct Renderer {
    const World* world;
    rt_renderer p;
    Color bg() const { return {p.bg[0], p.bg[1], p.bg[2]}; }
    Color fg() const { return {p.fg[0], p.fg[1], p.fg[2]}; }

    // render.rs:195-205
    Color solve(const Ray& ray, Pcg& pcg, Counters* cnt) const {
        switch (p.algorithm) {
            case RT_ALGO_ONOFF: return solve_onoff(ray, cnt);
            case RT_ALGO_FLAT: return solve_flat(ray, cnt);
            case RT_ALGO_PATHTRACER: return solve_pathtracer(ray, pcg, cnt);
            default: return {1.0f, 2.0f, 3.0f};  // render.rs:180-185
        }
    }
    // render.rs:57-67
    Color solve_onoff(const Ray& ray, Counters* cnt) const {
        HitRecord hit;
        return world->ray_intersection(ray, &hit, cnt) ? fg() : bg();
    }
    // render.rs:76-89
    Color solve_flat(const Ray& ray, Counters* cnt) const {
        HitRecord hit;
        if (!world->ray_intersection(ray, &hit, cnt)) return bg();
        return add(hit.material->emitted_radiance.get_color(hit.surface_point),
                   hit.material->brdf_pigment.get_color(hit.surface_point));
    }
    // render.rs:136-172
    Color solve_pathtracer(const Ray& ray, Pcg& pcg, Counters* cnt) const {
        if (ray.depth > p.max_depth) return {0.f, 0.f, 0.f};
        HitRecord hit;
        if (!world->ray_intersection(ray, &hit, cnt)) return bg();
        const Material* hit_material = hit.material;
        Color hit_color = hit_material->brdf_pigment.get_color(hit.surface_point);
        const Color emitted_radiance = hit_material->emitted_radiance.get_color(hit.surface_point);
        const float hit_color_lum = rt_maxf(hit_color.r, rt_maxf(hit_color.g, hit_color.b));
        if (ray.depth >= p.rr_limit) {
            const float q = rt_maxf(1.0f - hit_color_lum, 0.05f);
            if (pcg.random_float() > q) hit_color = mul(hit_color, 1.0f / (1.0f - q));
            else return emitted_radiance;
        }
        Color cum_radiance = {0.f, 0.f, 0.f};
        if (hit_color_lum > 0.0f) {
            for (uint32_t i = 0; i < p.num_of_rays; ++i) {
                const float r0 = pcg.random_float();  // tuple fields evaluate left to right
                const float r1 = pcg.random_float();
                Ray new_ray;
                if (hit_material->brdf == RT_BRDF_DIFFUSE) {  // material.rs:265-284
                    new_ray = diffuse_scatter_ray(*world->mf, r0, r1, hit.world_point, hit.normal, ray.depth + 1);
                    if (ray.depth + 1 <= p.max_depth) cnt->diffuse_scatters++;
                } else {
                    new_ray = specular_scatter_ray(hit.ray.dir, hit.world_point, hit.normal, ray.depth + 1);
                    if (ray.depth + 1 <= p.max_depth) cnt->specular_scatters++;
                }
                const Color new_radiance = solve_pathtracer(new_ray, pcg, cnt);
                cum_radiance = add(cum_radiance, mul(hit_color, new_radiance));
            }
        }
        return add(emitted_radiance, mul(cum_radiance, 1.0f / (float)p.num_of_rays));
    }
};

// ---------------------------------------------------------------------------------------------
// camera.rs
// ---------------------------------------------------------------------------------------------
struct Camera {
    rt_camera c;
    Transformation t;  // only .m is read by Transformation * Ray
    // camera.rs:68-75, :132-143, :155-160
    Ray fire_ray(float u, float v) const {
        Ray r = default_ray();
        if (c.kind == RT_CAMERA_ORTHOGONAL) {
            r.origin = {-1.0f, (1.0f - 2.0f * u) * c.aspect, 2.0f * v - 1.0f};
            r.dir = {1.0f, 0.0f, 0.0f};
        } else {
            r.origin = {-c.distance, 0.0f, 0.0f};
            r.dir = {c.distance, (1.0f - 2.0f * u) * c.aspect, 2.0f * v - 1.0f};
        }
        return apply(t, r);
    }
};
Camera make_camera(const rt_camera* c) {
    Camera cam;
    cam.c = *c;
    float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    cam.t = from_flat(c->m, ident);
    return cam;
}

// ---------------------------------------------------------------------------------------------
// imagetracer.rs
// ---------------------------------------------------------------------------------------------
struct ImageTracer {
    uint32_t width, height;
    Camera camera;
    // imagetracer.rs:52-56
    Ray fire_ray(uint32_t col, uint32_t row, float u_pixel, float v_pixel) const {
        const float u = ((float)col + u_pixel) / (float)width;
        const float v = 1.0f - ((float)row + v_pixel) / (float)height;
        return camera.fire_ray(u, v);
    }
};

struct Rays { uint64_t seq; size_t first_ray; };

// optional diagnostics sink: World::ray_intersection calls per pixel of the next fire_all_rays
uint32_t* g_pixel_rays = nullptr;

}  // namespace

// =============================================================================================
// exported checker entry points (ctypes)
// =============================================================================================
extern "C" {

// ImageTracer::fire_all_rays (imagetracer.rs:105-125) incl. the serial all_rays prepass
// (imagetracer.rs:61-84).  Rows [row_begin,row_end) only are shaded (the prepass still walks the
// master PCG over every earlier pixel, as the reference's single stream requires); out_rgb holds
// (row_end-row_begin)*width*3 floats.  Pixels are handed to n_threads workers from an atomic
// counter (rayon par_iter stand-in; order-preserving like collect()).
// Columns [col_begin, col_end) of those rows only (a window of a frame too large to shade whole on the CPU: 10 002 shapes at
// 7680x4320); out_rgb holds (row_end-row_begin)*(col_end-col_begin)*3 floats.  The master PCG still walks every pixel.
int oracle_fire_all_rays_window(const rt_shape* shapes, uint32_t n_shapes, const rt_material* materials,
                                uint32_t n_materials, const rt_texture* textures, uint32_t n_textures,
                                const rt_camera* camera, const rt_renderer* renderer, uint32_t width,
                                uint32_t height, uint64_t init_state, uint64_t init_seq,
                                uint32_t antialiasing_level, uint32_t row_begin, uint32_t row_end,
                                uint32_t col_begin, uint32_t col_end,
                                int math_mode, int n_threads, float* out_rgb, rt_stats* stats) {
    World world;
    if (!build_world(&world, shapes, n_shapes, materials, n_materials, textures, n_textures, math_mode)) return 1;
    if (row_end > height) row_end = height;
    if (row_begin > row_end) row_begin = row_end;
    if (col_end > width) col_end = width;
    if (col_begin > col_end) col_begin = col_end;
    const auto t0 = std::chrono::steady_clock::now();
    Renderer r{&world, *renderer};
    ImageTracer tracer{width, height, make_camera(camera)};
    const uint32_t aa = antialiasing_level;

    // all_rays (imagetracer.rs:61-84)
    std::vector<Rays> all_rays;
    std::vector<Ray> ray_store;
    all_rays.reserve((size_t)(row_end - row_begin) * (col_end - col_begin));
    ray_store.reserve((size_t)(row_end - row_begin) * (col_end - col_begin) * aa * aa);
    Pcg pcg(init_state, init_seq);
    for (uint32_t row = 0; row < row_end; ++row) {
        for (uint32_t col = 0; col < width; ++col) {
            const size_t first = ray_store.size();
            for (uint32_t sub_row = 0; sub_row < aa; ++sub_row) {
                for (uint32_t sub_col = 0; sub_col < aa; ++sub_col) {
                    const float u_pixel = ((float)sub_row + pcg.random_float()) / (float)aa;
                    const float v_pixel = ((float)sub_col + pcg.random_float()) / (float)aa;
                    if (row >= row_begin && col >= col_begin && col < col_end) ray_store.push_back(tracer.fire_ray(col, row, u_pixel, v_pixel));
                }
            }
            const uint64_t seq = (uint64_t)pcg.random();
            if (row >= row_begin && col >= col_begin && col < col_end) all_rays.push_back({seq, first});
        }
    }

    // fire_all_rays (imagetracer.rs:112-124)
    const size_t n_pixels = all_rays.size();
    const size_t rays_per_pixel = (size_t)aa * aa;
    const float inv_samples = 1.0f / (float)(uint32_t)(aa * aa);
    if (n_threads < 1) n_threads = 1;
    std::atomic<size_t> next{0};
    std::vector<Counters> counters(n_threads);
    auto worker = [&](int tid) {
        Counters cnt;
        // pixels per hand-out: 64, fewer when the call shades only a small window (every thread should get several)
        size_t grain = n_pixels / (8 * (size_t)n_threads);
        if (grain > 64) grain = 64;
        if (grain < 1) grain = 1;
        for (;;) {
            const size_t b = next.fetch_add(grain);
            if (b >= n_pixels) break;
            const size_t e = b + grain < n_pixels ? b + grain : n_pixels;
            for (size_t i = b; i < e; ++i) {
                Color color = {0.f, 0.f, 0.f};
                const uint64_t rays_before = cnt.rays;
                Pcg ppcg(init_state, all_rays[i].seq);
                for (size_t k = 0; k < rays_per_pixel; ++k) {
                    cnt.samples++;
                    color = add(color, r.solve(ray_store[all_rays[i].first_ray + k], ppcg, &cnt));
                }
                color = mul(color, inv_samples);
                out_rgb[3 * i + 0] = color.r;
                out_rgb[3 * i + 1] = color.g;
                out_rgb[3 * i + 2] = color.b;
                if (g_pixel_rays) g_pixel_rays[i] = (uint32_t)(cnt.rays - rays_before);
            }
        }
        counters[tid] = cnt;
    };
    const auto t1 = std::chrono::steady_clock::now();
    if (n_threads == 1) {
        worker(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t) th.emplace_back(worker, t);
        for (auto& t : th) t.join();
    }
    const auto t2 = std::chrono::steady_clock::now();
    if (stats) {
        Counters tot;
        for (auto& c : counters) tot += c;
        stats->rays = tot.rays; stats->samples = tot.samples;
        stats->sphere_tests = tot.sphere_tests; stats->plane_tests = tot.plane_tests;
        stats->diffuse_scatters = tot.diffuse_scatters; stats->specular_scatters = tot.specular_scatters;
        stats->rays_shared = 0;                          // the reference traces every child
        stats->kernel_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();  // par_iter part
        stats->total_ms = std::chrono::duration<double, std::milli>(t2 - t0).count();   // incl. all_rays
    }
    return 0;
}

int oracle_fire_all_rays(const rt_shape* shapes, uint32_t n_shapes, const rt_material* materials,
                         uint32_t n_materials, const rt_texture* textures, uint32_t n_textures,
                         const rt_camera* camera, const rt_renderer* renderer, uint32_t width,
                         uint32_t height, uint64_t init_state, uint64_t init_seq,
                         uint32_t antialiasing_level, uint32_t row_begin, uint32_t row_end,
                         int math_mode, int n_threads, float* out_rgb, rt_stats* stats) {
    return oracle_fire_all_rays_window(shapes, n_shapes, materials, n_materials, textures, n_textures, camera, renderer, width,
                                       height, init_state, init_seq, antialiasing_level, row_begin, row_end, 0u, width,
                                       math_mode, n_threads, out_rgb, stats);
}

// Diagnostics: per-pixel ray counts of subsequent oracle_fire_all_rays calls (NULL = off).
int oracle_set_pixel_ray_counter(uint32_t* buf) {
    g_pixel_rays = buf;
    return 0;
}

// all_rays seeds only (imagetracer.rs:61-84): per pixel the aa^2 (u_pixel, v_pixel) pairs and seq.
int oracle_all_rays_seeds(uint32_t width, uint32_t height, uint64_t init_state, uint64_t init_seq,
                          uint32_t aa, float* uv_pixel /* w*h*aa*aa*2 */, uint64_t* seq /* w*h */) {
    Pcg pcg(init_state, init_seq);
    size_t k = 0;
    for (uint32_t row = 0; row < height; ++row)
        for (uint32_t col = 0; col < width; ++col) {
            for (uint32_t sub_row = 0; sub_row < aa; ++sub_row)
                for (uint32_t sub_col = 0; sub_col < aa; ++sub_col) {
                    uv_pixel[k++] = ((float)sub_row + pcg.random_float()) / (float)aa;
                    uv_pixel[k++] = ((float)sub_col + pcg.random_float()) / (float)aa;
                }
            seq[(size_t)row * width + col] = (uint64_t)pcg.random();
        }
    return 0;
}

// ImageTracer::fire_ray (imagetracer.rs:52-56) for explicit (col,row,u_pixel,v_pixel) tuples.
int oracle_fire_ray(const rt_camera* camera, uint32_t width, uint32_t height, const uint32_t* col,
                    const uint32_t* row, const float* u_pixel, const float* v_pixel, uint64_t n,
                    rt_ray* out) {
    ImageTracer tracer{width, height, make_camera(camera)};
    for (uint64_t i = 0; i < n; ++i) {
        const Ray r = tracer.fire_ray(col[i], row[i], u_pixel[i], v_pixel[i]);
        out[i] = {{r.origin.x, r.origin.y, r.origin.z}, {r.dir.x, r.dir.y, r.dir.z}, r.tmin, r.tmax, r.depth};
    }
    return 0;
}

// Camera::fire_ray (camera.rs:155-160) for explicit (u,v).
int oracle_camera_fire_ray(const rt_camera* camera, const float* u, const float* v, uint64_t n, rt_ray* out) {
    const Camera cam = make_camera(camera);
    for (uint64_t i = 0; i < n; ++i) {
        const Ray r = cam.fire_ray(u[i], v[i]);
        out[i] = {{r.origin.x, r.origin.y, r.origin.z}, {r.dir.x, r.dir.y, r.dir.z}, r.tmin, r.tmax, r.depth};
    }
    return 0;
}

static Ray to_ray(const rt_ray& r) {
    return {{r.origin[0], r.origin[1], r.origin[2]}, {r.dir[0], r.dir[1], r.dir[2]}, r.tmin, r.tmax, r.depth};
}

// World::ray_intersection (world.rs:32-50)
int oracle_world_ray_intersection(const rt_shape* shapes, uint32_t n_shapes, const rt_material* materials,
                                  uint32_t n_materials, const rt_texture* textures, uint32_t n_textures,
                                  const rt_ray* rays, uint64_t n_rays, int math_mode, rt_hit* hits) {
    World world;
    if (!build_world(&world, shapes, n_shapes, materials, n_materials, textures, n_textures, math_mode)) return 1;
    Counters cnt;
    for (uint64_t i = 0; i < n_rays; ++i) {
        HitRecord h;
        std::memset(&hits[i], 0, sizeof(rt_hit));
        if (world.ray_intersection(to_ray(rays[i]), &h, &cnt)) {
            hits[i].hit = 1;
            hits[i].shape = h.shape_index;
            hits[i].world_point[0] = h.world_point.x; hits[i].world_point[1] = h.world_point.y; hits[i].world_point[2] = h.world_point.z;
            hits[i].normal[0] = h.normal.x; hits[i].normal[1] = h.normal.y; hits[i].normal[2] = h.normal.z;
            hits[i].uv[0] = h.surface_point.u; hits[i].uv[1] = h.surface_point.v;
            hits[i].t = h.t;
        }
    }
    return 0;
}

// Renderer::solve (render.rs:195-205)
int oracle_renderer_solve(const rt_shape* shapes, uint32_t n_shapes, const rt_material* materials,
                          uint32_t n_materials, const rt_texture* textures, uint32_t n_textures,
                          const rt_renderer* renderer, const rt_ray* rays, rt_pcg* pcgs, uint64_t n_rays,
                          int math_mode, float* out_rgb) {
    World world;
    if (!build_world(&world, shapes, n_shapes, materials, n_materials, textures, n_textures, math_mode)) return 1;
    Renderer r{&world, *renderer};
    Counters cnt;
    for (uint64_t i = 0; i < n_rays; ++i) {
        Pcg pcg(0, 0);
        pcg.state = pcgs[i].state;
        pcg.inc = pcgs[i].inc;
        const Color c = r.solve(to_ray(rays[i]), pcg, &cnt);
        pcgs[i].state = pcg.state;
        pcgs[i].inc = pcg.inc;
        out_rgb[3 * i] = c.r; out_rgb[3 * i + 1] = c.g; out_rgb[3 * i + 2] = c.b;
    }
    return 0;
}

// Pcg::new + random (random.rs:26-43)
int oracle_pcg_new(uint64_t init_state, uint64_t init_seq, rt_pcg* out) {
    Pcg p(init_state, init_seq);
    out->state = p.state;
    out->inc = p.inc;
    return 0;
}
int oracle_pcg_random(const uint64_t* init_state, const uint64_t* init_seq, uint64_t n_seeds,
                      uint32_t n_draws, uint32_t* out) {
    for (uint64_t i = 0; i < n_seeds; ++i) {
        Pcg p(init_state[i], init_seq[i]);
        for (uint32_t k = 0; k < n_draws; ++k) out[i * n_draws + k] = p.random();
    }
    return 0;
}
// Pcg::random_float (random.rs:46-48) continuing from an explicit state
int oracle_pcg_random_float(rt_pcg* pcg, uint32_t n_draws, float* out) {
    Pcg p(0, 0);
    p.state = pcg->state;
    p.inc = pcg->inc;
    for (uint32_t k = 0; k < n_draws; ++k) out[k] = p.random_float();
    pcg->state = p.state;
    pcg->inc = p.inc;
    return 0;
}

// Pigment::get_color (material.rs:106-115)
int oracle_pigment_get_color(const rt_pigment* pigment, const rt_texture* textures, uint32_t n_textures,
                             const float* uv, uint64_t n, float* out_rgb) {
    Texture tex{0, 0, nullptr};
    Pigment p;
    p.kind = pigment->kind;
    p.c1 = {pigment->c1[0], pigment->c1[1], pigment->c1[2]};
    p.c2 = {pigment->c2[0], pigment->c2[1], pigment->c2[2]};
    p.steps = pigment->steps;
    p.tex = nullptr;
    if (pigment->kind == RT_PIGMENT_IMAGE) {
        if (pigment->texture >= n_textures) return 1;
        tex = {textures[pigment->texture].w, textures[pigment->texture].h, textures[pigment->texture].rgb};
        p.tex = &tex;
    }
    for (uint64_t i = 0; i < n; ++i) {
        const Color c = p.get_color({uv[2 * i], uv[2 * i + 1]});
        out_rgb[3 * i] = c.r; out_rgb[3 * i + 1] = c.g; out_rgb[3 * i + 2] = c.b;
    }
    return 0;
}

// BRDF::scatter_ray (material.rs:265-284)
int oracle_scatter_ray(uint32_t brdf, float pcg0, float pcg1, const float* incoming_dir,
                       const float* interaction_point, const float* normal, uint32_t depth,
                       int math_mode, rt_ray* out) {
    const Point ip = {interaction_point[0], interaction_point[1], interaction_point[2]};
    const Normal n = {normal[0], normal[1], normal[2]};
    Ray r;
    if (brdf == RT_BRDF_DIFFUSE) r = diffuse_scatter_ray(math_for(math_mode), pcg0, pcg1, ip, n, depth);
    else r = specular_scatter_ray({incoming_dir[0], incoming_dir[1], incoming_dir[2]}, ip, n, depth);
    *out = {{r.origin.x, r.origin.y, r.origin.z}, {r.dir.x, r.dir.y, r.dir.z}, r.tmin, r.tmax, r.depth};
    return 0;
}

// normal.rs:143-158
int oracle_create_onb_from_z(const float* normal, float* e123 /* 9 */) {
    Vector e1, e2, e3;
    create_onb_from_z({normal[0], normal[1], normal[2]}, &e1, &e2, &e3);
    e123[0] = e1.x; e123[1] = e1.y; e123[2] = e1.z;
    e123[3] = e2.x; e123[4] = e2.y; e123[5] = e2.z;
    e123[6] = e3.x; e123[7] = e3.y; e123[8] = e3.z;
    return 0;
}

// Transformation * {Point, Vector, Normal} (transformation.rs:133-195); what: 0 point 1 vector 2 normal
int oracle_transform(const float* m, const float* invm, int what, const float* in3, float* out3) {
    const Transformation t = from_flat(m, invm);
    if (what == 0) { const Point p = apply(t, Point{in3[0], in3[1], in3[2]}); out3[0] = p.x; out3[1] = p.y; out3[2] = p.z; }
    else if (what == 1) { const Vector p = apply(t, Vector{in3[0], in3[1], in3[2]}); out3[0] = p.x; out3[1] = p.y; out3[2] = p.z; }
    else { const Normal p = apply(t, Normal{in3[0], in3[1], in3[2]}); out3[0] = p.x; out3[1] = p.y; out3[2] = p.z; }
    return 0;
}

// include/rt_strict_math.h (mode 1) or this box's glibc (mode 0) on the host.
// fn: 0 sin, 1 cos, 2 atan2(y=a, x=b), 3 acos, 4 a/b, 5 sqrt(a)
int oracle_math_eval(int fn, int math_mode, const float* a, const float* b, uint64_t n, float* out) {
    const MathFns& mf = math_for(math_mode);
    for (uint64_t i = 0; i < n; ++i) {
        switch (fn) {
            case 0: out[i] = mf.sin(a[i]); break;
            case 1: out[i] = mf.cos(a[i]); break;
            case 2: out[i] = mf.atan2(a[i], b[i]); break;
            case 3: out[i] = mf.acos(a[i]); break;
            case 4: out[i] = a[i] / b[i]; break;
            case 7: out[i] = rt_maxf(a[i], b[i]); break;                     // the host body of the shared helper
            case 8: { const uint32_t u = rt_f32_to_u32_sat(a[i]); std::memcpy(&out[i], &u, 4); break; }
            default: out[i] = std::sqrt(a[i]); break;
        }
    }
    return 0;
}

}  // extern "C"
