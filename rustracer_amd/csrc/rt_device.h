// rt_device.h — device-resident scene layout and the per-ray device functions shared by every
// kernel of librustracer_amd (gfx950 only).
//
// Every function states the reference function whose arithmetic it reproduces (paths relative to
// /root/reference).  All f32 expressions keep the reference's operand order and associativity;
// the translation unit is compiled with -ffp-contract=off so no mul/add pair is fused.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_strict_math.h"
#include "../../include/rustracer_amd.h"

#pragma clang fp contract(off)

namespace rtd {

// ---------------------------------------------------------------------------------------------
// HBM layout of the flattened World
// ---------------------------------------------------------------------------------------------

// Scan table entry (64 B): what the closest-hit loop needs per shape, read wave-uniformly
// (LDS broadcast when the table is staged, scalar loads otherwise).
struct __attribute__((aligned(16))) DScan {
    float inv[12];      // Transformation.invm rows 0..2, row-major
    uint32_t kind;      // RT_SHAPE_*
    uint32_t material;  // index into the material table (the shape index when staged in LDS)
    uint32_t needs_uv;  // 1 if either pigment of the material reads (u,v)
    uint32_t shape;     // the hierarchy's leaf records (a copy of this table in leaf order): the shape's index in the World
};
static_assert(sizeof(DScan) == 64, "DScan layout");
// Transformation.m rows 0..2 (48 B), read per lane once per ray for the winner.
struct __attribute__((aligned(16))) DMRows { float m[12]; };
static_assert(sizeof(DMRows) == 48, "DMRows layout");
// Last rows of invm and m (32 B); only read when some matrix of the scene is not affine.
struct __attribute__((aligned(16))) DWRows { float inv_w[4]; float m_w[4]; };
static_assert(sizeof(DWRows) == 32, "DWRows layout");

// Shading record of a shape (32 B): the colors of its material when both pigments are uniform
// (material.rs:32-36), so the common case needs two 16-B loads and no pigment evaluation.
struct __attribute__((aligned(16))) DShade {
    float hc[3];        // brdf pigment color (uniform)
    uint32_t flags;     // bit 0: RT_BRDF_*; bit 1 / bit 2: the brdf / emitted pigment reads (u,v) -> evaluate the DMaterial
    float em[3];        // emitted radiance (uniform)
    uint32_t pad;
};
static_assert(sizeof(DShade) == 32, "DShade layout");

struct DPigment {
    uint32_t kind;      // RT_PIGMENT_*
    float c1[3];
    float c2[3];
    uint32_t steps;
    uint32_t tex_w, tex_h;
    uint32_t tex_off;   // float offset of the texture in the texel pool
    uint32_t pad;
};
static_assert(sizeof(DPigment) == 48, "DPigment layout");

struct __attribute__((aligned(16))) DMaterial {
    uint32_t brdf;      // RT_BRDF_*
    uint32_t pad[3];
    DPigment brdf_pigment;
    DPigment emitted;
};
static_assert(sizeof(DMaterial) == 112, "DMaterial layout");

struct Vec3 { float x, y, z; };

// ---------------------------------------------------------------------------------------------
// PCG (src/random.rs)
// ---------------------------------------------------------------------------------------------
#define RT_PCG_MULT 6364136223846793005ULL

// Pcg::random output permutation of the OLD state (random.rs:40-42).
__device__ __forceinline__ uint32_t pcg_output(uint64_t old_state) {
    const uint32_t xor_shifted = (uint32_t)(((old_state >> 18) ^ old_state) >> 27);
    const uint32_t rot = (uint32_t)(old_state >> 59);
    return (xor_shifted >> rot) | (xor_shifted << ((0u - rot) & 31u));
}
// Pcg::random (random.rs:37-43)
__device__ __forceinline__ uint32_t pcg_random(uint64_t& state, uint64_t inc) {
    const uint64_t old_state = state;
    state = old_state * RT_PCG_MULT + inc;
    return pcg_output(old_state);
}
// Pcg::random_float (random.rs:46-48): r as f32 / 2^32 == rn(r) * 2^-32 exactly.
__device__ __forceinline__ float pcg_random_float(uint64_t& state, uint64_t inc) {
    return (float)pcg_random(state, inc) * 2.3283064365386962890625e-10f;
}
// Pcg::new (random.rs:26-34): state after the two warm-up draws.
__device__ __forceinline__ void pcg_new(uint64_t init_state, uint64_t init_seq, uint64_t& state, uint64_t& inc) {
    inc = (init_seq << 1) | 1ULL;
    state = 0;
    (void)pcg_random(state, inc);
    state += init_state;
    (void)pcg_random(state, inc);
}

// ---------------------------------------------------------------------------------------------
// Transformation * {Point, Vector, Normal} (src/transformation.rs:133-195)
// ---------------------------------------------------------------------------------------------

// T*Point with matrix rows M[0..15]; AFFINE promises M[12..15] == {0,0,0,1} bit-exactly, in which
// case w == 1 for every finite point and the reference takes the no-divide branch.
template <bool AFFINE>
__device__ __forceinline__ Vec3 xform_point(const float* __restrict__ M, float x, float y, float z) {
    Vec3 r;
    r.x = M[0] * x + M[1] * y + M[2] * z + M[3];
    r.y = M[4] * x + M[5] * y + M[6] * z + M[7];
    r.z = M[8] * x + M[9] * y + M[10] * z + M[11];
    if (!AFFINE) {
        const float w = x * M[12] + y * M[13] + z * M[14] + M[15];
        if (!(w == 1.0f)) {
            r.x = r.x / w;
            r.y = r.y / w;
            r.z = r.z / w;
        }
    }
    return r;
}
__device__ __forceinline__ Vec3 xform_vector(const float* __restrict__ M, float x, float y, float z) {
    Vec3 r;
    r.x = M[0] * x + M[1] * y + M[2] * z;
    r.y = M[4] * x + M[5] * y + M[6] * z;
    r.z = M[8] * x + M[9] * y + M[10] * z;
    return r;
}
// ---------------------------------------------------------------------------------------------
// Correctly rounded f32 sqrt without the compiler's denormal-range prologue/epilogue: the same
// v_sqrt_f32 + one-ulp-neighbour residual test the compiler expands __builtin_sqrtf to, minus the
// 2^32 pre-scaling it needs only for x < 2^-96 and the zero/inf class select (0, +inf and NaN fall
// through the residual tests unchanged).  The caller guarantees x == 0 or x >= 2^-96.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sqrt_rn_normal(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u);
    const float s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x);
    const float r_up = __builtin_fmaf(-s_up, s, x);
    float r = (r_dn <= 0.0f) ? s_dn : s;
    r = (r_up > 0.0f) ? s_up : r;
    return r;
}
// sqrt for any x: the short form unless some lane of the wave holds a tiny positive x (wave-uniform branch)
__device__ __forceinline__ float sqrt_rn(float x) {
    if (__ballot(x > 0.0f && x < 0x1.0p-96f) != 0ull) return __builtin_sqrtf(x);
    return sqrt_rn_normal(x);
}

// ---------------------------------------------------------------------------------------------
// Shape::ray_intersection, candidate part: only `t` is produced here; the HitRecord of the final
// winner is built once by hit_record() below (result-identical: World keeps one record).
// ---------------------------------------------------------------------------------------------

// Sphere::ray_intersection up to first_hit_t (src/shape.rs:112-133) and Plane::ray_intersection up to t
// (src/shape.rs:196-204), the reference's values and decisions (sqrt of a negative delta is NaN and fails every
// comparison; delta <= 0 is rejected explicitly; a NaN plane t passes both range tests exactly as in the reference).
//
// One IEEE divide per sphere instead of the reference's two, same result: the reference takes t1 = n1 / (2a) if it lies
// in (tmin, tmax), else t2 = n2 / (2a).  2a is +0, positive, +inf or NaN (a sum of squares), so with tmin >= 0 a
// numerator n1 that is not > 0 gives a quotient that is <= 0, -inf or NaN and fails `t1 > tmin` whatever its rounding:
// such lanes (every ray that starts inside the sphere, e.g. all rays against a sky sphere; every ray whose sphere lies
// behind it) divide n2 straight away.  Lanes with n1 > 0 divide n1; the few whose t1 then falls outside (tmin, tmax)
// divide again (a wave-level rarity).  A negative tmin (batch calls with the caller's rays) keeps the first root.
__device__ __forceinline__ bool sphere_t_nb(Vec3 o, Vec3 d, float tmin, float tmax, float& t_out) {
    const float a = d.x * d.x + d.y * d.y + d.z * d.z;
    const float b = 2.0f * (o.x * d.x + o.y * d.y + o.z * d.z);
    const float c = (o.x * o.x + o.y * o.y + o.z * o.z) - 1.0f;
    const float delta = b * b - 4.0f * a * c;
    const float sqrt_delta = sqrt_rn(delta);
    const float a2 = 2.0f * a;
    const float n1 = -b - sqrt_delta, n2 = -b + sqrt_delta;
    const bool first = (n1 > 0.0f) || !(tmin >= 0.0f);
    float t = (first ? n1 : n2) / a2;
    bool h = (t > tmin) && (t < tmax);
    if (first && !h) {
        t = n2 / a2;
        h = (t > tmin) && (t < tmax);
    }
    t_out = t;
    return (delta > 0.0f) && h;
}
__device__ __forceinline__ bool plane_t_nb(Vec3 o, Vec3 d, float tmin, float tmax, float& t_out) {
    const float t = -o.z / d.z;
    t_out = t;
    return !(__builtin_fabsf(d.z) < 1e-5f) && !((t <= tmin) || (t >= tmax));
}

struct HitRec {
    Vec3 world_point;
    Vec3 normal;
    float u, v;
};

// ---------------------------------------------------------------------------------------------
// Transformation * {Point, Vector, Normal} on matrix rows held in registers (DScan / DMRows / DWRows)
// ---------------------------------------------------------------------------------------------
struct Rows3 { float4 r0, r1, r2; };
__device__ __forceinline__ Rows3 load_rows(const float* __restrict__ p) {
    const float4* q = reinterpret_cast<const float4*>(p);
    Rows3 r; r.r0 = q[0]; r.r1 = q[1]; r.r2 = q[2];
    return r;
}
// T*Point (src/transformation.rs:163-195): rows 0..2 in R; W = last row (read only if !AFFINE).
template <bool AFFINE>
__device__ __forceinline__ Vec3 rows_point(const Rows3& R, const float* __restrict__ W, float x, float y, float z) {
    Vec3 r;
    r.x = R.r0.x * x + R.r0.y * y + R.r0.z * z + R.r0.w;
    r.y = R.r1.x * x + R.r1.y * y + R.r1.z * z + R.r1.w;
    r.z = R.r2.x * x + R.r2.y * y + R.r2.z * z + R.r2.w;
    if (!AFFINE) {
        const float w = x * W[0] + y * W[1] + z * W[2] + W[3];
        if (!(w == 1.0f)) {
            r.x = r.x / w;
            r.y = r.y / w;
            r.z = r.z / w;
        }
    }
    return r;
}
// T*Vector (src/transformation.rs:133-143)
__device__ __forceinline__ Vec3 rows_vector(const Rows3& R, float x, float y, float z) {
    Vec3 r;
    r.x = R.r0.x * x + R.r0.y * y + R.r0.z * z;
    r.y = R.r1.x * x + R.r1.y * y + R.r1.z * z;
    r.z = R.r2.x * x + R.r2.y * y + R.r2.z * z;
    return r;
}
// T*Normal with I = invm rows (src/transformation.rs:145-161)
__device__ __forceinline__ Vec3 rows_normal(const Rows3& I, float x, float y, float z) {
    Vec3 r;
    r.x = I.r0.x * x + I.r1.x * y + I.r2.x * z;
    r.y = I.r0.y * x + I.r1.y * y + I.r2.y * z;
    r.z = I.r0.z * x + I.r1.z * y + I.r2.z * z;
    return r;
}

// ---------------------------------------------------------------------------------------------
// Diagonal inverse matrices (every shape a composition of scalings and translations: rows 0..2 of invm are
// [a 0 0 tx], [0 b 0 ty], [0 0 c tz] with exact zeros).  The reference still evaluates the full products
// ((m0*x + m1*y) + m2*z) + m3 (src/transformation.rs:133-195); with finite operands the two zero terms are signed
// zeros, and adding a signed zero changes nothing unless the other operand is itself a zero.  So a*x + t equals the
// full expression bit for bit unless a*x and t are both zeros, where only the SIGN of the resulting zero can differ;
// likewise a*d equals the full direction component unless it is a zero.  diag_ray_ok() admits only rays whose
// direction components are all non-zero (|d_k| in [2^-60, 2^60]; the host checks |a|, |b|, |c| in the same range, so
// no product underflows to zero) and whose origin is finite and bounded.  Then d'_k != 0 exactly as in the reference,
// and a zero o'_k of either sign is harmless: a, c, delta and the roots do not see it (squares; b only if every
// term is a zero, and then -b -/+ sqrt(delta) absorbs it), the plane's t = -/+0 is rejected by t <= tmin for any
// tmin >= 0 either way, and the hit point o'_k + d'_k * t is the non-zero d'_k * t (t > tmin > 0, no underflow:
// >= 2^-120 * tmin).  Every decision and every value that leaves the scan is therefore identical.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool diag_ray_ok(Vec3 o, Vec3 d) {
    const float lim = 0x1.0p60f, tiny = 0x1.0p-60f;
    const float so = __builtin_fabsf(o.x) + __builtin_fabsf(o.y) + __builtin_fabsf(o.z);   // NaN / inf propagate
    const float sd = __builtin_fabsf(d.x) + __builtin_fabsf(d.y) + __builtin_fabsf(d.z);
    const float mn = __builtin_fminf(__builtin_fminf(__builtin_fabsf(d.x), __builtin_fabsf(d.y)), __builtin_fabsf(d.z));
    return (so <= lim) && (sd <= lim) && (mn >= tiny);
}
__device__ __forceinline__ Vec3 rows_point_diag(const Rows3& R, float x, float y, float z) {
    Vec3 r;
    r.x = R.r0.x * x + R.r0.w;
    r.y = R.r1.y * y + R.r1.w;
    r.z = R.r2.z * z + R.r2.w;
    return r;
}
__device__ __forceinline__ Vec3 rows_vector_diag(const Rows3& R, float x, float y, float z) {
    Vec3 r;
    r.x = R.r0.x * x;
    r.y = R.r1.y * y;
    r.z = R.r2.z * z;
    return r;
}

// World::ray_intersection (src/world.rs:32-50) over a DScan table that every lane reads at the same
// index (LDS broadcast or scalar loads): linear scan in insertion order, strict `<`, so the earliest
// shape wins ties and the first hit is taken unconditionally (even t = NaN).  G shapes per iteration
// (the loads of a group are issued together), no control flow inside a shape test; also returns the
// winner's object-space ray and kind.
template <bool AFFINE, int G, bool DIAG = false>
__device__ __forceinline__ int scan_closest_hit_g(const DScan* __restrict__ scan, const DWRows* __restrict__ wrows,
                                                  uint32_t n_shapes, Vec3 ro, Vec3 rd, float tmin, float tmax,
                                                  float& best_t, Vec3& best_o, Vec3& best_d, uint32_t& best_kind) {
    int best = -1;
    best_t = 0.0f;
    best_kind = 0u;
    best_o = {0.f, 0.f, 0.f};
    best_d = {0.f, 0.f, 0.f};
    for (uint32_t i0 = 0; i0 < n_shapes; i0 += G) {
        Rows3 R[G];
        uint32_t kind[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint32_t i = (i0 + g < n_shapes) ? i0 + g : n_shapes - 1u;
            R[g] = load_rows(scan[i].inv);
            kind[g] = (uint32_t)__builtin_amdgcn_readfirstlane((int)scan[i].kind);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint32_t i = (i0 + g < n_shapes) ? i0 + g : n_shapes - 1u;
            const Vec3 o = DIAG ? rows_point_diag(R[g], ro.x, ro.y, ro.z)
                                : rows_point<AFFINE>(R[g], AFFINE ? nullptr : wrows[i].inv_w, ro.x, ro.y, ro.z);
            const Vec3 d = DIAG ? rows_vector_diag(R[g], rd.x, rd.y, rd.z) : rows_vector(R[g], rd.x, rd.y, rd.z);
            float t = 0.0f;
            bool hit;
            if (kind[g] == RT_SHAPE_SPHERE) hit = sphere_t_nb(o, d, tmin, tmax, t);
            else hit = plane_t_nb(o, d, tmin, tmax, t);
            if ((i0 + g < n_shapes) && hit && (best < 0 || t < best_t)) {
                best = (int)(i0 + g);
                best_t = t;
                best_o = o;
                best_d = d;
                best_kind = kind[g];
            }
        }
    }
    return best;
}

// ---------------------------------------------------------------------------------------------
// Exact-winner bounding volume hierarchy for scenes too large for the linear scan to
// stay in LDS.  World::ray_intersection's result is min over the hits by (t, index) -- the linear scan's
// strict `<` keeps the earliest shape on ties -- so any traversal order gives the reference's winner as
// long as (1) every shape runs the SAME object-space test (sphere_t_nb / plane_t_nb on its own rows) and
// (2) no shape whose f32 test can report a hit is culled.  (2) is the delicate part: far from a small
// sphere the f32 discriminant is dominated by rounding, and the reference "hits" rays that pass well
// outside it.  With eps = 2^-24, rho2 / rhoF = spectral / Frobenius norm of invm's 3x3, kF = rhoF * (spectral
// norm of m's 3x3), D >= |origin - any point of the shape's box| and W >= |origin| + |centre| (Euclidean),
// forward error analysis of o' = T^-1 o (|do'| <= 4 eps rhoF W), d' = T^-1 d (|dd'| <= 3 eps kF |d'|), a, b, c
// and delta = b*b - 4ac gives |d delta| <= eps a [(48 kF + 68)|o'|^2 + 64 |o'| rhoF W + 24 kF + 28], i.e. delta
// computed > 0 only if the ray passes the unit sphere in object space within radius sqrt(1 + S),
//     S <= eps [(12 kF + 17) rho2^2 D^2 + 16 rho2 rhoF D W + 6 kF + 7]
// and the reported hit point lies in a ball of that order too.  A node is culled only if the ray misses its
// box grown by hmax * (sqrt(1 + 2 S) - 1) (2x the bound, S from the node's maxima) plus 1e-6 W for the
// slab arithmetic itself, or enters it beyond the best t so far by a margin.  Planes and shapes with
// projective matrices are unbounded: they are tested for every ray.  Rays with a non-finite component, and rays
// for which an unbounded shape reports a hit at t = NaN, take the linear scan (a NaN t wins or loses by order there).
// ---------------------------------------------------------------------------------------------
// What a node test needs is folded into the node on the host (rt_host.inc: BvhBuild::build), in double, rounded up.  With
// D = |origin - farthest corner of the box|, c = that corner's distance from the world origin, W^2 = 2 |o|^2 + 2 c^2 and the
// maxima below the node hmax, qa = 2 (12 kF + 17) rho2^2, qb = 2 * 16 rho2 rhoF, g0 = 2 eps (6 kF + 7):
//     S = eps (qa D^2 + qb D W) + g0,      box growth  hmax (sqrt(1 + S) - 1) <= hmax min(S / 2, sqrt(S))
// The node carries sa = eps qa / 2, sb = eps qb / 2, sg = g0 / 2 (each padded by 1e-5 for the roundings of their
// evaluation and the v_sqrt_f32 estimate of D W) and 2 c^2; the traversal evaluates S / 2 = sa D^2 + sb sqrt(D^2 W^2) + sg.
// Two shortcuts that were measured and do NOT work on the sphere-flake: bounding D W by (D^2 + W^2) / 2 inflates the boxes of
// small shapes near the ray's origin (D << W), and dropping the sqrt(S) branch inflates them for rays that start on the sky
// sphere (D ~ 100, S ~ 140 for the smallest spheres: growth 69 hmax instead of 12 hmax) -- 16x the frame time either way.
// The slab arithmetic's own rounding is covered by 1e-6 (|o|_1 + |c|_1): the |c|_1 part is added to the half-extents here,
// the |o|_1 part per ray.
struct __attribute__((aligned(16))) DBvhNode {
    float cx, cy, cz, hmax;   // box centre; largest half-extent of a shape below this node
    float hx, hy, hz, sa;     // box half-extents + 1e-6 |farthest corner|_1; sa (above)
    uint32_t link;            // inner node: the next node once this subtree is done or culled, as a byte offset into the node
                              // array (DFS order: first child = this + 1); leaf (the node after it is always the next one):
                              // 0x80000000 | first * 32 + count: entries [first, first + count) of the leaf records (DBvh.leaf_scan)
    float sb, sg, c2x2;       // (above)
};
static_assert(sizeof(DBvhNode) == 48, "DBvhNode layout");

struct DBvh {
    const DBvhNode* nodes;       // n_nodes, or 8 * n_nodes: the tree laid out once per sign octant of the ray direction (near half first)
    const DScan* leaf_scan;      // the scan records of the bounded shapes in leaf order (DScan.shape: the shape's index): a leaf's
                                 // tests read consecutive records, no index list in between
    const uint32_t* unbounded;   // shapes outside the hierarchy, ascending
    uint32_t n_nodes, n_unbounded;
    uint32_t n_leaf;             // records of leaf_scan
    uint32_t by_octant;          // 1: `nodes` holds the 8 layouts; 0: one layout (small trees: 8 copies would only thrash the L1)
};

// The traversal of one ray as a resumable state (per lane), so that a kernel may keep rays IN FLIGHT across its rounds:
// a wave's rays need very different numbers of node visits (incoherent bounce rays: 16 of 64 lanes were active per
// traversal instruction when every round waited for its slowest ray), so the throughput body lets the lanes whose
// traversal has finished go on to shading and to their next ray while the others keep walking (bvh_advance's min_done).
struct BvhTrav {
    uint32_t i, i_end;        // next node of this ray's layout and the layout's end, as byte offsets into the node array (i >= i_end: finished)
    int best;                 // winner so far (-1 none) and its record, as the linear scan keeps them
    float best_t;
    Vec3 best_o, best_d;
    uint32_t best_kind;
};

// one shape's own test; the winner is the minimum by (t, index).  `diag` is wave-uniform.
template <bool AFFINE>
__device__ __forceinline__ void bvh_test_record(const DScan* __restrict__ S, const DWRows* __restrict__ wrows, uint32_t i,
                                                Vec3 ro, Vec3 rd, float tmin, float tmax, bool diag, BvhTrav& T) {
    const Rows3 R = load_rows(S->inv);
    const uint32_t kind = S->kind;
    Vec3 o, d;
    if (AFFINE && diag) {
        o = rows_point_diag(R, ro.x, ro.y, ro.z);
        d = rows_vector_diag(R, rd.x, rd.y, rd.z);
    } else {
        o = rows_point<AFFINE>(R, AFFINE ? nullptr : wrows[i].inv_w, ro.x, ro.y, ro.z);
        d = rows_vector(R, rd.x, rd.y, rd.z);
    }
    float t = 0.0f;
    bool hit;
    if (kind == RT_SHAPE_SPHERE) hit = sphere_t_nb(o, d, tmin, tmax, t);
    else hit = plane_t_nb(o, d, tmin, tmax, t);
    if (hit && (T.best < 0 || t < T.best_t || (t == T.best_t && (int)i < T.best))) {
        T.best = (int)i;
        T.best_t = t;
        T.best_o = o;
        T.best_d = d;
        T.best_kind = kind;
    }
}

template <bool AFFINE>
__device__ __forceinline__ void bvh_test_shape(const DScan* __restrict__ scan, const DWRows* __restrict__ wrows, uint32_t i,
                                               Vec3 ro, Vec3 rd, float tmin, float tmax, bool diag, BvhTrav& T) {
    bvh_test_record<AFFINE>(scan + i, wrows, i, ro, rd, tmin, tmax, diag, T);
}

// Start a ray (every ACTIVE lane starts one; call it inside the branch of the starting lanes): the unbounded shapes,
// the fall-backs to the linear scan, the entry node.  diag: wave-uniform over the starting lanes (sparse form allowed,
// diag_ray_ok).  cached, per lane: -2 unknown; else the known answer to this very query (-1 no hit, >= 0 the winning
// shape): only that shape is tested (the children of a mirror hit all trace the same ray, src/material.rs:225-246;
// the same inputs through the same arithmetic give the same t and object-space ray).
template <bool AFFINE>
__device__ __forceinline__ void bvh_begin(const DBvh B, const DScan* __restrict__ scan, const DWRows* __restrict__ wrows,
                                          uint32_t n_shapes, Vec3 ro, Vec3 rd, float tmin, float tmax, const bool diag,
                                          const int cached, BvhTrav& T) {
    T.i = 0u;
    T.i_end = 0u;
    const bool finite = (__builtin_fabsf(ro.x) < __builtin_inff()) && (__builtin_fabsf(ro.y) < __builtin_inff()) &&
                        (__builtin_fabsf(ro.z) < __builtin_inff()) && (__builtin_fabsf(rd.x) < __builtin_inff()) &&
                        (__builtin_fabsf(rd.y) < __builtin_inff()) && (__builtin_fabsf(rd.z) < __builtin_inff());
    if (__ballot(!finite) != 0ull) {
        T.best = scan_closest_hit_g<AFFINE, 2>(scan, wrows, n_shapes, ro, rd, tmin, tmax, T.best_t, T.best_o, T.best_d, T.best_kind);
        return;
    }
    T.best = -1;
    T.best_t = 0.0f;
    T.best_kind = 0u;
    T.best_o = {0.f, 0.f, 0.f};
    T.best_d = {0.f, 0.f, 0.f};
    const bool walk = (cached == -2);
    for (uint32_t u = 0; u < B.n_unbounded; ++u)
        if (walk) bvh_test_shape<AFFINE>(scan, wrows, B.unbounded[u], ro, rd, tmin, tmax, diag, T);
    // Plane::ray_intersection reports a hit whose t is NaN (its two rejections are comparisons that NaN fails,
    // src/shape.rs:198-204); the linear scan then keeps that hit iff no earlier shape is hit at all, which is not
    // a minimum over (t, index).  Only unbounded shapes can do this: such a wave takes the linear scan.
    if (__ballot(T.best >= 0 && T.best_t != T.best_t) != 0ull) {
        T.best = scan_closest_hit_g<AFFINE, 2>(scan, wrows, n_shapes, ro, rd, tmin, tmax, T.best_t, T.best_o, T.best_d, T.best_kind);
        return;
    }
    // this ray's near-to-far layout: nodes [i, i_end) of the concatenated array
    const uint32_t octant = B.by_octant ? ((rd.x < 0.0f ? 1u : 0u) | (rd.y < 0.0f ? 2u : 0u) | (rd.z < 0.0f ? 4u : 0u)) : 0u;
    T.i_end = (octant + 1u) * B.n_nodes * (uint32_t)sizeof(DBvhNode);
    T.i = walk ? octant * B.n_nodes * (uint32_t)sizeof(DBvhNode) : T.i_end;
    if (cached >= 0) bvh_test_shape<AFFINE>(scan, wrows, (uint32_t)cached, ro, rd, tmin, tmax, diag, T);
}

// Walk the rays in flight (`act`: this lane has one; ro / rd / tmin / tmax / T are its ray and state) until at least
// min_done of them have finished or none has work left.  "while-while" traversal: every lane walks inner nodes until it
// stands on a leaf whose box it passes (or is done), then the wave runs the shape tests of all those leaves together -- a
// lane's leaf visits are few, so the expensive tests execute a few times per ray instead of once per node step of the
// slowest lane.  lane_diag: this lane's ray admits the sparse form (diag_ray_ok; only read if scene_diag).
template <bool AFFINE>
__device__ __forceinline__ void bvh_advance(const DBvh B, const DScan* __restrict__ scan, const DWRows* __restrict__ wrows,
                                            Vec3 ro, Vec3 rd, float tmin, float tmax, const bool scene_diag, const bool lane_diag,
                                            const bool act, const uint32_t min_done, BvhTrav& T) {
    if (__ballot(act && T.i < T.i_end) == 0ull) return;
    const float big = 1.0e30f;
    const float idx = (__builtin_fabsf(rd.x) < 1.0e-30f) ? __builtin_copysignf(big, rd.x) : 1.0f / rd.x;
    const float idy = (__builtin_fabsf(rd.y) < 1.0e-30f) ? __builtin_copysignf(big, rd.y) : 1.0f / rd.y;
    const float idz = (__builtin_fabsf(rd.z) < 1.0e-30f) ? __builtin_copysignf(big, rd.z) : 1.0f / rd.z;
    // the slabs as (c -/+ e) * (1 / d) - o * (1 / d): the second product once per ray.  A ray for which it is not comfortably
    // finite (origin beyond 1e8 with a direction component below 1e-22, ...) culls nothing
    const float rix = ro.x * idx, riy = ro.y * idy, riz = ro.z * idz;
    const bool wide = !((__builtin_fabsf(rix) < big) && (__builtin_fabsf(riy) < big) && (__builtin_fabsf(riz) < big));
    const float o2x2 = 2.0f * (ro.x * ro.x + ro.y * ro.y + ro.z * ro.z);
    const float o1e = 1.0e-6f * (__builtin_fabsf(ro.x) + __builtin_fabsf(ro.y) + __builtin_fabsf(ro.z));
    for (;;) {
      uint32_t leaf_first = 0, leaf_count = 0;
      uint32_t i = T.i;
      const uint32_t i_end = act ? T.i_end : 0u;
      // a node that starts beyond the best hit so far (or beyond tmax) is culled; best only changes between the inner loops
      const float tlim = (T.best >= 0) ? __builtin_fminf(tmax, T.best_t) : tmax;
      while (i < i_end) {
        // the conservative node test is this library's own arithmetic, not the reference's: fused multiply-adds are welcome
        // (every bound below carries a relative margin far above one rounding)
#pragma clang fp contract(fast)
        const float4* np = reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(B.nodes) + i);
        const float4 n0 = np[0], n1 = np[1], n2 = np[2];
        // D: distance to the farthest corner of the (padded) box
        const float fx = __builtin_fabsf(ro.x - n0.x) + n1.x, fy = __builtin_fabsf(ro.y - n0.y) + n1.y, fz = __builtin_fabsf(ro.z - n0.z) + n1.z;
        const float D2 = fx * fx + fy * fy + fz * fz;
        const float DW = __builtin_amdgcn_sqrtf(D2 * (o2x2 + n2.w));
        const float S2 = n1.w * D2 + (n2.y * DW + n2.z);                                     // S / 2
        const float grow = __builtin_fminf(S2, __builtin_amdgcn_sqrtf(S2) * 1.4285f);         // sqrt(S) = sqrt(2) sqrt(S / 2), the estimate padded by 1 %
#ifdef RT_BVH_NO_MARGIN   // experiment: how many reference hits a geometric (unpadded) hierarchy would lose
        const float e = 0.0f * grow;
#else
        const float e = n0.w * grow + o1e;
#endif
        const float ex = n1.x + e, ey = n1.y + e, ez = n1.z + e;
        const float ax = (n0.x - ex) * idx - rix, bx = (n0.x + ex) * idx - rix;
        const float ay = (n0.y - ey) * idy - riy, by = (n0.y + ey) * idy - riy;
        const float az = (n0.z - ez) * idz - riz, bz = (n0.z + ez) * idz - riz;
        const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
        const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
        const float slack = 1.0e-3f * __builtin_fabsf(tn) + 1.0e-4f;
        // (no short-circuit: three compares and their conjunction cost less than the branches around them)
        const bool p_order = tn <= tf, p_near = tf + 1.0e-3f * __builtin_fabsf(tf) + 1.0e-4f >= tmin, p_far = tn - slack <= tlim;
        const bool pass = (p_order & p_near & p_far) | wide;
        const uint32_t link = __float_as_uint(n2.x);
        const bool is_leaf = (int)link < 0;
        i = (pass | is_leaf) ? i + (uint32_t)sizeof(DBvhNode) : link;
        if (pass & is_leaf) {
            leaf_first = (link & 0x7FFFFFFFu) >> 5;
            leaf_count = link & 31u;
            break;
        }
      }
      if (act) T.i = i;
      const unsigned long long at_leaf = __ballot(leaf_count != 0u);
      if (at_leaf == 0ull) break;
      const bool diag = AFFINE && scene_diag && __ballot(leaf_count != 0u && !lane_diag) == 0ull;
      for (uint32_t k = 0; k < leaf_count; ++k) {
          const DScan* S = B.leaf_scan + leaf_first + k;
          bvh_test_record<AFFINE>(S, wrows, S->shape, ro, rd, tmin, tmax, diag, T);
      }
      if ((uint32_t)__popcll(__ballot(act && T.i >= T.i_end)) >= min_done) break;
    }
}

// World::ray_intersection through the hierarchy in one call (every active lane traces one ray to the end).
template <bool AFFINE>
__device__ __forceinline__ int bvh_closest_hit(const DBvh B, const DScan* __restrict__ scan, const DWRows* __restrict__ wrows,
                                               uint32_t n_shapes, Vec3 ro, Vec3 rd, float tmin, float tmax,
                                               float& best_t, Vec3& best_o, Vec3& best_d, uint32_t& best_kind,
                                               const bool diag = false /* wave-uniform: sparse form allowed (diag_ray_ok) */,
                                               const int cached = -2 /* per lane, see bvh_begin */) {
    BvhTrav T;
    bvh_begin<AFFINE>(B, scan, wrows, n_shapes, ro, rd, tmin, tmax, diag, cached, T);
    bvh_advance<AFFINE>(B, scan, wrows, ro, rd, tmin, tmax, diag, true, true, 65u, T);
    best_t = T.best_t;
    best_o = T.best_o;
    best_d = T.best_d;
    best_kind = T.best_kind;
    return T.best;
}

// HitRecord fields of the winner (src/shape.rs:134-142, :205-213) on the split layout.  want_geom
// false skips world_point and normal (a node that cannot scatter never reads them).
template <bool AFFINE>
__device__ __forceinline__ HitRec scan_hit_record(const DScan* __restrict__ S, const DMRows* __restrict__ MR,
                                                  const DWRows* __restrict__ WR, Vec3 ro, Vec3 rd, float t,
                                                  bool want_uv, bool want_geom) {
    HitRec h;
    const Rows3 I = load_rows(S->inv);
    const Vec3 o = rows_point<AFFINE>(I, AFFINE ? nullptr : WR->inv_w, ro.x, ro.y, ro.z);
    const Vec3 d = rows_vector(I, rd.x, rd.y, rd.z);
    const float hx = o.x + d.x * t, hy = o.y + d.y * t, hz = o.z + d.z * t;   // inv_ray.at(t)
    h.world_point = {0.f, 0.f, 0.f};
    h.normal = {0.f, 0.f, 1.f};
    h.u = 0.0f;
    h.v = 0.0f;
    const bool sphere = (S->kind == RT_SHAPE_SPHERE);
    if (want_geom) {
        const Rows3 M = load_rows(MR->m);
        h.world_point = rows_point<AFFINE>(M, AFFINE ? nullptr : WR->m_w, hx, hy, hz);
        float nx, ny, nz, dp;
        if (sphere) {
            dp = hx * d.x + hy * d.y + hz * d.z;
            nx = hx; ny = hy; nz = hz;
        } else {
            dp = 0.0f * d.x + 0.0f * d.y + 1.0f * d.z;
            nx = 0.0f; ny = 0.0f; nz = 1.0f;
        }
        if (!(dp < 0.0f)) { nx = -nx; ny = -ny; nz = -nz; }
        h.normal = rows_normal(I, nx, ny, nz);
    }
    if (want_uv) {
        if (sphere) {
            float u = rt_atan2f(hy, hx) / 6.2831854820251465f;  // 2.0 * PI in f32
            const float v = rt_acosf(hz) / 3.1415927410125732f; // PI in f32
            if (u < 0.0f) u += 1.0f;
            h.u = u;
            h.v = v;
        } else {
            h.u = hx - __builtin_floorf(hx);
            h.v = hy - __builtin_floorf(hy);
        }
    }
    return h;
}

// ---------------------------------------------------------------------------------------------
// Pigment::get_color (src/material.rs:32-36, :54-70, :86-96, :106-115)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ Vec3 pigment_color(const DPigment* __restrict__ p, const float* __restrict__ texels, float u, float v) {
    // all fields are fetched up front (one batch of loads, no dependent round trips)
    const uint32_t kind = p->kind;
    const float c1x = p->c1[0], c1y = p->c1[1], c1z = p->c1[2];
    const float c2x = p->c2[0], c2y = p->c2[1], c2z = p->c2[2];
    const uint32_t isteps = p->steps;
    Vec3 c = {c1x, c1y, c1z};                         // UniformPigment (material.rs:32-36)
    if (kind == RT_PIGMENT_CHECKERED) {               // material.rs:86-96
        const float steps = (float)isteps;
        const uint32_t int_u = rt_f32_to_u32_sat(__builtin_floorf(u * steps));
        const uint32_t int_v = rt_f32_to_u32_sat(__builtin_floorf(v * steps));
        if ((int_u % 2u) != (int_v % 2u)) { c.x = c2x; c.y = c2y; c.z = c2z; }
    } else if (kind == RT_PIGMENT_IMAGE) {            // material.rs:54-70
        const uint32_t w = p->tex_w, h = p->tex_h;
        uint32_t col = rt_f32_to_u32_sat(u * (float)w);
        uint32_t row = rt_f32_to_u32_sat(v * (float)h);
        if (col >= w) col = w - 1;
        if (row >= h) row = h - 1;
        const float* px = texels + p->tex_off + 3ull * ((uint64_t)row * w + col);
        c.x = px[0]; c.y = px[1]; c.z = px[2];
    }
    return c;
}

// ---------------------------------------------------------------------------------------------
// Pigments of a hit: the colours of the brdf pigment and of the emitted radiance at the hit's (u, v)
// (src/shape.rs:101-108, :187-192; src/material.rs:54-70, :86-96).
//
// A pigment never reads (u, v) itself, only the CELL it falls in: floor(u * steps) and floor(v * steps) of a checkered
// pigment, the texel (u * w) as u32, (v * h) as u32 of an image.  On a sphere (u, v) = (atan2(y, x) / 2 pi, acos(z) / pi)
// costs four binary64 divides, a binary64 square root and two polynomials (rt_strict_math.h) -- 30 % of the textured
// sphere-flake's frame.  So the cell is first looked for with single-precision estimates of u and v whose distance from
// the reference's f32 values is bounded (RT_UV_DELTA_*); if no cell boundary lies within that bound of the scaled
// estimate, the reference's value falls in the same cell and the colour is the reference's, bit for bit.  Lanes for
// which a boundary is that close, or anything is not finite, or the hit is near a pole, evaluate the reference
// functions as before (a wave-level rarity: 0.1-0.5 % of the lanes per coordinate).  The bounds are checked against the
// strict functions on the GPU (tests/test_gpu_parity.py: rt_strict_math_eval fn 9, 10).
// ---------------------------------------------------------------------------------------------
#define RT_UV_DELTA_ATAN 1.5e-6f   // |atan2_est - rt_atan2f| <= this (radians); measured maximum 2.4e-7
#define RT_UV_DELTA_U 6.0e-7f      // |u_est - u_reference|: DELTA_ATAN / 2 pi, the reference's divide and wrap, this multiply
#define RT_UV_DELTA_V 8.0e-7f      // |v_est - v_reference|: (DELTA_ATAN + the square root's 2.4e-7) / pi, the divide, the multiply

// atan2 for finite operands, not both zero: octant reduction, a * P(a^2) with P of degree 7 (|a P - atan a| < 4e-8 in
// exact arithmetic, < 1.5e-7 evaluated in f32), v_rcp_f32 for the ratio (1 ulp).  NaN when y == x == 0 or an operand is not finite.
__device__ __forceinline__ float atan2_est(float y, float x) {
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const float mx = __builtin_fmaxf(ax, ay), mn = __builtin_fminf(ax, ay);
    const float a = mn * __builtin_amdgcn_rcpf(mx);
    const float z = a * a;
    float p = -0.004053469747304916f;
    p = p * z + 0.021858856081962585f;
    p = p * z + -0.055906206369400024f;
    p = p * z + 0.09641734510660172f;
    p = p * z + -0.13908442854881287f;
    p = p * z + 0.19946527481079102f;
    p = p * z + -0.33329856395721436f;
    p = p * z + 0.9999993443489075f;
    float r = a * p;
    if (ay > ax) r = 1.5707963705062866f - r;
    if (x < 0.0f) r = 3.1415927410125732f - r;
    return (y < 0.0f) ? -r : r;
}
// acos for |z| < 1 the same way the reference defines it: atan2(sqrt((1 - z)(1 + z)), z); NaN beyond
__device__ __forceinline__ float acos_est(float z) {
    return atan2_est(__builtin_amdgcn_sqrtf((1.0f - z) * (1.0f + z)), z);    // v_sqrt_f32: 1 ulp
}
// the cell floor(c * scale) if it is certain: c_est within delta of the reference's c, the reference rounds c * scale once
__device__ __forceinline__ bool cell_sure(float c_est, float delta, float scale, uint32_t& cell) {
    const float x = c_est * scale;
    const float eps = delta * scale + x * 2.4e-7f + 1.0e-30f;
    const float lo = __builtin_floorf(x - eps), hi = __builtin_floorf(x + eps);
    cell = (uint32_t)hi;                                   // only used when sure (then 0 <= hi < 2^31)
    return (lo == hi) && (x - eps > 0.0f) && (hi < 2147483648.0f);   // a NaN fails every comparison
}
// Pigment::get_color from certain cells; false if this pigment needs the reference's (u, v)
__device__ __forceinline__ bool pigment_color_est(const DPigment* __restrict__ p, const float* __restrict__ texels, float u_est, float v_est, Vec3& c) {
    const uint32_t kind = p->kind;
    c = {p->c1[0], p->c1[1], p->c1[2]};
    if (kind == RT_PIGMENT_CHECKERED) {
        const float steps = (float)p->steps;
        uint32_t iu, iv;
        const bool ok_u = cell_sure(u_est, RT_UV_DELTA_U, steps, iu), ok_v = cell_sure(v_est, RT_UV_DELTA_V, steps, iv);
        const bool ok = ok_u && ok_v;
        if (ok && (iu % 2u) != (iv % 2u)) c = {p->c2[0], p->c2[1], p->c2[2]};
        return ok;
    }
    if (kind == RT_PIGMENT_IMAGE) {
        const uint32_t w = p->tex_w, h = p->tex_h;
        uint32_t col, row;
        const bool ok_u = cell_sure(u_est, RT_UV_DELTA_U, (float)w, col), ok_v = cell_sure(v_est, RT_UV_DELTA_V, (float)h, row);
        const bool ok = ok_u && ok_v;
        if (ok) {
            if (col >= w) col = w - 1;
            if (row >= h) row = h - 1;
            const float* px = texels + p->tex_off + 3ull * ((uint64_t)row * w + col);
            c = {px[0], px[1], px[2]};
        }
        return ok;
    }
    return true;
}
// hit_color = brdf pigment, emitted = emitted pigment at the hit (render.rs:146-147); flags bit 1 / bit 2: that pigment
// reads (u, v); (hx, hy, hz): the hit point in object space (inv_ray.at(t)); hc / em come in holding the uniform colours
__device__ __forceinline__ void hit_pigments(const DMaterial* __restrict__ M, const float* __restrict__ texels, uint32_t sflags,
                                             bool sphere, float hx, float hy, float hz, Vec3& hc, Vec3& em) {
    if (!sphere) {
        const float u = hx - __builtin_floorf(hx), v = hy - __builtin_floorf(hy);     // shape.rs:187-192
        if (sflags & 2u) hc = pigment_color(&M->brdf_pigment, texels, u, v);
        if (sflags & 4u) em = pigment_color(&M->emitted, texels, u, v);
        return;
    }
    bool need_b = (sflags & 2u) != 0u, need_e = (sflags & 4u) != 0u;
#ifndef RT_NO_UV_ESTIMATE
    {
        float u_est = atan2_est(hy, hx) * 0.15915494309189535f;       // / 2 pi
        if (u_est < 0.0f) u_est += 1.0f;
        // a u within the bound of the wrap at 0 could be the reference's u + 1; a hit within 1e-6 of a pole, or off the unit
        // sphere (|z| > 1: the reference's acos is NaN there), is left to the reference functions
        const bool tame = (__builtin_fabsf(u_est) > 2.0f * RT_UV_DELTA_U) && (__builtin_fabsf(hz) < 0.999999f);
        const float v_est = acos_est(hz) * 0.3183098861837907f;       // / pi
        Vec3 c;
        if (need_b && tame && pigment_color_est(&M->brdf_pigment, texels, u_est, v_est, c)) { hc = c; need_b = false; }
        if (need_e && tame && pigment_color_est(&M->emitted, texels, u_est, v_est, c)) { em = c; need_e = false; }
    }
    if (__ballot(need_b || need_e) == 0ull) return;
#endif
    if (need_b || need_e) {
        float u = rt_atan2f(hy, hx) / 6.2831854820251465f;  // 2.0 * PI in f32 (shape.rs:101-108)
        const float v = rt_acosf(hz) / 3.1415927410125732f; // PI in f32
        if (u < 0.0f) u += 1.0f;
        if (need_b) hc = pigment_color(&M->brdf_pigment, texels, u, v);
        if (need_e) em = pigment_color(&M->emitted, texels, u, v);
    }
}

// ---------------------------------------------------------------------------------------------
// BRDF::scatter_ray (src/material.rs:161-186, :225-246) + create_onb_from_z (src/normal.rs:143-158)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ Vec3 diffuse_scatter_dir(float rf0, float rf1, Vec3 n) {
    const float sign = __builtin_copysignf(1.0f, n.z);
    const float a = -1.0f / (sign + n.z);
    const float b = n.x * n.y * a;
    const float e1x = 1.0f + sign * n.x * n.x * a, e1y = sign * b, e1z = -sign * n.x;
    const float e2x = b, e2y = sign + n.y * n.y * a, e2z = -n.y;
    // rf0 = k * 2^-32 rounded to f32, in {0} U [2^-32, 1]; 1 - rf0 in {0} U [2^-24, 1]
    const float cos_theta = sqrt_rn_normal(rf0);
    const float sin_theta = sqrt_rn_normal(1.0f - rf0);
    const float phi = 6.2831854820251465f * rf1;  // (2.0 * PI) * pcg.1
    float sn, cs;
    rt_sincosf(phi, &sn, &cs);
    Vec3 d;
    d.x = e1x * cs * cos_theta + e2x * sn * cos_theta + n.x * sin_theta;
    d.y = e1y * cs * cos_theta + e2y * sn * cos_theta + n.y * sin_theta;
    d.z = e1z * cs * cos_theta + e2z * sn * cos_theta + n.z * sin_theta;
    return d;
}

__device__ __forceinline__ Vec3 specular_scatter_dir(Vec3 in, Vec3 n) {
    // Vector::normalize = v * (1 / norm) (src/vector.rs:72-75)
    const float inv_r = 1.0f / __builtin_sqrtf(in.x * in.x + in.y * in.y + in.z * in.z);
    const float rx = in.x * inv_r, ry = in.y * inv_r, rz = in.z * inv_r;
    const float inv_n = 1.0f / __builtin_sqrtf(n.x * n.x + n.y * n.y + n.z * n.z);
    const float nx = n.x * inv_n, ny = n.y * inv_n, nz = n.z * inv_n;
    const float dot_prod = nx * rx + ny * ry + nz * rz;
    Vec3 d;
    d.x = rx - nx * 2.0f * dot_prod;
    d.y = ry - ny * 2.0f * dot_prod;
    d.z = rz - nz * 2.0f * dot_prod;
    return d;
}

}  // namespace rtd
