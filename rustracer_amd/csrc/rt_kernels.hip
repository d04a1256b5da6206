// rt_kernels.hip — gfx950 kernels of the render core.
//
// k_render is a persistent, register-resident path-state machine (one lane = one pixel in
// flight, exactly one ray in flight per lane):
//
//   generate : lanes without a ray take the next sample of their pixel, or a new pixel from the
//              wave's private chunk of the pixel queue (one global atomic per 64-pixel chunk);
//              the master PCG of ImageTracer::all_rays is reproduced by LCG jump-ahead.
//   intersect: World::ray_intersection — wave-uniform scan of the shape table (scalar loads).
//   shade    : HitRecord of the winner, pigments, renderer logic, Russian roulette; advances the
//              pixel's depth-first recursion by one node, consuming PCG draws in reference order.
//   compact  : finished lanes are refilled in place (wave64 ballot + prefix rank), so a wave
//              stays full until the pixel queue is empty.
//
// The recursion of PathTracer::solve (src/render.rs:136-172) is kept per lane as: the deepest
// open hit ("current level") in registers, its ancestors in an LDS stack [slot][field][thread]
// (global memory when max_depth is too deep for LDS).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device.h"

#pragma clang fp contract(off)

namespace rtd {

#define RT_BLOCK 256
#define RT_CHUNK 64u
#define RT_NF 19           // dwords per saved recursion level
#define RT_INVALID 0xFFFFFFFFu

// Everything a render / solve launch needs, passed by value (kernarg -> SGPRs).
struct RenderParams {
    uint32_t n_shapes;
    uint32_t scene_in_lds;           // 1: shapes + per-shape materials are staged in LDS
    // camera (src/camera.rs)
    uint32_t cam_kind;
    float cam_distance, cam_aspect;
    float cam_m[16];
    // renderer (src/render.rs)
    uint32_t algorithm;
    float bg[3], fg[3];
    uint32_t num_of_rays, max_depth, rr_limit;
    float inv_num_of_rays;           // 1. / (num_of_rays as f32)      (render.rs:171)
    uint64_t skip_mult, skip_plus;   // PCG advance by 2*num_of_rays draws: s' = s*mult + inc*plus
    // image (src/imagetracer.rs)
    uint32_t width, height, aa;
    float inv_samples;               // 1. / (aa.pow(2) as f32)        (imagetracer.rs:121)
    uint64_t init_state;
    uint64_t master_state0, master_inc;   // Pcg::new(init_state, init_seq)
    uint64_t pix_mult[32], pix_plus[32];  // master jump by 2^i pixels (2*aa^2+1 draws each)
    uint64_t seq_mult, seq_plus;          // master jump by 2*aa^2 draws (to the pixel's seq draw)
    // partition (row tiles dealt round-robin)
    uint32_t part, n_parts, tile_rows;
    uint32_t n_local;                // pixels of this part
    // solve mode
    const rt_ray* in_rays;
    rt_pcg* io_pcgs;
    // outputs
    float* out;
    uint32_t* queue_head;            // pixel-queue head (zeroed before launch)
    unsigned long long* counters;    // [0] rays [1] samples [2] diffuse [3] specular
    // recursion stack
    uint32_t stack_slots;            // max ancestors saved per lane
    float* gstack;                   // global stack [slot][field][global thread] when not in LDS
    uint32_t total_threads;
};

// Saved level state accessors: STACK_LDS ? LDS : global.
template <bool STACK_LDS>
struct LevelStack {
    float* base;       // LDS: &lds[tid]; global: &gstack[gtid]
    uint32_t stride;   // blockDim or total threads
    __device__ __forceinline__ void st(uint32_t slot, uint32_t f, float v) const { base[(slot * RT_NF + f) * stride] = v; }
    __device__ __forceinline__ float ld(uint32_t slot, uint32_t f) const { return base[(slot * RT_NF + f) * stride]; }
};

// MODE 0: image render (pixel queue).  MODE 1: batch Renderer::solve on caller-supplied (ray, pcg).
// NS > 0: the scene has exactly NS shapes (affine): the closest-hit scan is fully unrolled with the
// inverse matrices held in SGPRs for the whole launch, and the scene is staged in LDS.
// The scene tables are separate __restrict__ const arguments so that wave-uniform reads of them
// are scalar (SMEM) loads.
template <bool AFFINE, bool STACK_LDS, int MODE, int NS>
__global__ __launch_bounds__(RT_BLOCK) void k_render(const RenderParams P, const DShape* __restrict__ g_shapes,
                                                     const DMaterial* __restrict__ g_materials,
                                                     const float* __restrict__ texels) {
    extern __shared__ float4 lds_raw[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t gtid = blockIdx.x * RT_BLOCK + tid;

    // ---- LDS: [DShape x n][DMaterial x n (one per shape)][recursion stack] ----
    const bool scene_lds = (NS > 0) ? true : (P.scene_in_lds != 0u);
    const uint32_t n_staged = scene_lds ? P.n_shapes : 0u;
    DShape* lds_shapes = reinterpret_cast<DShape*>(lds_raw);
    DMaterial* lds_mats = reinterpret_cast<DMaterial*>(lds_shapes + n_staged);
    float* lds_stack = reinterpret_cast<float*>(lds_mats + n_staged);
    if (scene_lds) {
        uint32_t* dst_s = reinterpret_cast<uint32_t*>(lds_shapes);
        const uint32_t* src_s = reinterpret_cast<const uint32_t*>(g_shapes);
        for (uint32_t i = tid; i < n_staged * (uint32_t)(sizeof(DShape) / 4); i += RT_BLOCK) dst_s[i] = src_s[i];
        uint32_t* dst_m = reinterpret_cast<uint32_t*>(lds_mats);
        constexpr uint32_t MW = sizeof(DMaterial) / 4;
        for (uint32_t i = tid; i < n_staged * MW; i += RT_BLOCK) {
            const uint32_t sh = i / MW, j = i - sh * MW;
            dst_m[i] = reinterpret_cast<const uint32_t*>(g_materials + g_shapes[sh].material)[j];
        }
        __syncthreads();
    }

    LevelStack<STACK_LDS> stack;
    if (STACK_LDS) { stack.base = lds_stack + tid; stack.stride = RT_BLOCK; }
    else { stack.base = P.gstack + gtid; stack.stride = P.total_threads; }

    // NS > 0: inverse matrices (rows 0..2) and kinds of all NS shapes, wave-uniform for the launch
    float sinv[NS > 0 ? NS : 1][12];
    uint32_t skind[NS > 0 ? NS : 1];
    if (NS > 0) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
#pragma unroll
            for (int j = 0; j < 12; ++j)
                sinv[i][j] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(g_shapes[i].inv[j])));
            skind[i] = __builtin_amdgcn_readfirstlane(g_shapes[i].kind);
        }
    }

    const uint32_t aa2 = P.aa * P.aa;
    const uint32_t n_rays_per_hit = P.num_of_rays;

    // ---- per-lane path state ----
    uint32_t q = RT_INVALID;        // local pixel (MODE 0) / batch index (MODE 1) owned by this lane
    uint32_t samp = 0;              // samples of q already started
    uint64_t mstate = 0;            // master PCG state at the next jitter draw of q
    uint64_t pstate = 0, pinc = 1;  // the pixel's PCG
    float acc_r = 0.f, acc_g = 0.f, acc_b = 0.f;
    uint32_t grow = 0, gcol = 0;    // global row / column of q
    bool has_ray = false;
    Vec3 ro = {0.f, 0.f, 0.f}, rd = {1.f, 0.f, 0.f};
    float rtmin = 1e-5f, rtmax = __builtin_inff();
    uint32_t lvl = 0;               // recursion level of the ray in flight (0 = top-level solve)
    uint32_t base_depth = 0;        // Ray.depth of the level-0 ray
    // current level = the open hit at level lvl-1
    float hc_r = 0.f, hc_g = 0.f, hc_b = 0.f, em_r = 0.f, em_g = 0.f, em_b = 0.f;
    float cu_r = 0.f, cu_g = 0.f, cu_b = 0.f;
    uint32_t kcount = 0, cur_brdf = 0;
    Vec3 cw = {0.f, 0.f, 0.f}, cn = {0.f, 0.f, 1.f}, cin = {1.f, 0.f, 0.f};
    // counters
    uint32_t n_rays = 0, n_samples = 0, n_diffuse = 0, n_specular = 0;

    // wave-private slice of the pixel queue
    uint32_t wcur = 0, wend = 0;
    bool exhausted = false;

    for (;;) {
        // =====================================================================================
        // generate / compact: refill lanes that have no ray in flight
        // =====================================================================================
        if (__ballot(!has_ray) != 0ull) {
            bool want_pixel = false;
            if (!has_ray) {
                if (q != RT_INVALID && samp == aa2) {
                    // pixel (or solve item) finished: color * (1 / aa^2)  (imagetracer.rs:121)
                    if (MODE == 0) {
                        float* o = P.out + 3ull * q;
                        o[0] = acc_r * P.inv_samples;
                        o[1] = acc_g * P.inv_samples;
                        o[2] = acc_b * P.inv_samples;
                    } else {
                        float* o = P.out + 3ull * q;
                        o[0] = acc_r; o[1] = acc_g; o[2] = acc_b;
                        P.io_pcgs[q].state = pstate;
                    }
                    q = RT_INVALID;
                }
                want_pixel = (q == RT_INVALID) && !exhausted;
            }
            // wave64 ballot + prefix rank hand out consecutive queue entries to the idle lanes
            unsigned long long wantmask = __ballot(want_pixel);
            bool newpix = false;
            while (wantmask != 0ull) {
                if (wcur == wend) {
                    uint32_t b = 0;
                    if (lane == 0) b = atomicAdd(P.queue_head, RT_CHUNK);
                    b = __shfl(b, 0);
                    if (b >= P.n_local) { exhausted = true; break; }
                    wcur = b;
                    wend = (b + RT_CHUNK < P.n_local) ? b + RT_CHUNK : P.n_local;
                }
                const uint32_t avail = wend - wcur;
                const uint32_t wanted = (uint32_t)__popcll(wantmask);
                const uint32_t cnt = wanted < avail ? wanted : avail;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(wantmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wantmask, 0u));
                if (want_pixel && rank < cnt) {
                    q = wcur + rank;
                    want_pixel = false;
                    newpix = true;
                }
                wcur += cnt;
                wantmask = __ballot(want_pixel);
            }
            if (!has_ray && q != RT_INVALID) {
                if (MODE == 0) {
                    if (newpix) {
                        // local pixel -> global (row, col) through the row-tile partition
                        const uint32_t lrow = q / P.width;
                        gcol = q - lrow * P.width;
                        const uint32_t tile = lrow / P.tile_rows;
                        grow = (tile * P.n_parts + P.part) * P.tile_rows + (lrow - tile * P.tile_rows);
                        // master PCG at this pixel: jump ahead p*(2*aa^2+1) draws (imagetracer.rs:63-81)
                        uint32_t p = grow * P.width + gcol;
                        uint64_t s = P.master_state0;
                        for (uint32_t i = 0; p != 0u; ++i, p >>= 1) {
                            const uint64_t s2 = s * P.pix_mult[i] + P.pix_plus[i];
                            s = (p & 1u) ? s2 : s;
                        }
                        mstate = s;
                        // seq = the u32 drawn after the pixel's 2*aa^2 jitter draws (imagetracer.rs:79)
                        const uint64_t sq = s * P.seq_mult + P.seq_plus;
                        const uint64_t seq = (uint64_t)pcg_output(sq);
                        pcg_new(P.init_state, seq, pstate, pinc);   // imagetracer.rs:117
                        samp = 0;
                        acc_r = 0.f; acc_g = 0.f; acc_b = 0.f;
                    }
                    // ImageTracer::all_rays jitter + fire_ray (imagetracer.rs:52-56, :67-75)
                    const uint32_t sub_row = samp / P.aa;
                    const uint32_t sub_col = samp - sub_row * P.aa;
                    const float rf0 = pcg_random_float(mstate, P.master_inc);
                    const float rf1 = pcg_random_float(mstate, P.master_inc);
                    const float u_pixel = ((float)sub_row + rf0) / (float)P.aa;
                    const float v_pixel = ((float)sub_col + rf1) / (float)P.aa;
                    const float u = ((float)gcol + u_pixel) / (float)P.width;
                    const float v = 1.0f - ((float)grow + v_pixel) / (float)P.height;
                    // Camera::fire_ray (camera.rs:68-75, :132-143) then Transformation * Ray
                    float cox, coy, coz, cdx, cdy, cdz;
                    if (P.cam_kind == RT_CAMERA_ORTHOGONAL) {
                        cox = -1.0f; coy = (1.0f - 2.0f * u) * P.cam_aspect; coz = 2.0f * v - 1.0f;
                        cdx = 1.0f; cdy = 0.0f; cdz = 0.0f;
                    } else {
                        cox = -P.cam_distance; coy = 0.0f; coz = 0.0f;
                        cdx = P.cam_distance; cdy = (1.0f - 2.0f * u) * P.cam_aspect; cdz = 2.0f * v - 1.0f;
                    }
                    ro = xform_point<AFFINE>(P.cam_m, cox, coy, coz);
                    rd = xform_vector(P.cam_m, cdx, cdy, cdz);
                    rtmin = 1e-5f;
                    rtmax = __builtin_inff();
                    base_depth = 0;
                } else {
                    const rt_ray r = P.in_rays[q];
                    ro = {r.origin[0], r.origin[1], r.origin[2]};
                    rd = {r.dir[0], r.dir[1], r.dir[2]};
                    rtmin = r.tmin;
                    rtmax = r.tmax;
                    base_depth = r.depth;
                    pstate = P.io_pcgs[q].state;
                    pinc = P.io_pcgs[q].inc;
                    samp = 0;
                    acc_r = 0.f; acc_g = 0.f; acc_b = 0.f;
                }
                samp += 1;
                n_samples += 1;
                lvl = 0;
                has_ray = true;
            }
        }
        if (__ballot(has_ray) == 0ull) break;  // queue empty and every lane retired

        if (has_ray) {
            // =================================================================================
            // intersect + shade one node of the lane's recursion
            // =================================================================================
            float vr = 0.f, vg = 0.f, vb = 0.f;  // value returned by this node, if it is a leaf
            bool leaf = true;
            const uint32_t depth = base_depth + lvl;

            if (P.algorithm == RT_ALGO_DUMMY) {
                vr = 1.0f; vg = 2.0f; vb = 3.0f;   // render.rs:180-185
            } else if (P.algorithm == RT_ALGO_PATHTRACER && depth > P.max_depth) {
                // render.rs:137-139 (only reachable for a caller-supplied deep ray)
            } else {
                float t;
                int best;
                if (NS > 0) {
                    best = -1;
                    t = 0.0f;
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        const Vec3 o = xform_point<true>(sinv[i], ro.x, ro.y, ro.z);
                        const Vec3 d = xform_vector(sinv[i], rd.x, rd.y, rd.z);
                        float ti = 0.0f;
                        bool hit;
                        if (skind[i] == RT_SHAPE_SPHERE) hit = sphere_t(o, d, rtmin, rtmax, ti);
                        else hit = plane_t(o, d, rtmin, rtmax, ti);
                        if (hit && (best < 0 || ti < t)) { best = i; t = ti; }
                    }
                } else {
                    best = closest_hit<AFFINE>(g_shapes, P.n_shapes, ro, rd, rtmin, rtmax, t);
                }
                n_rays += 1;
                if (best < 0) {
                    vr = P.bg[0]; vg = P.bg[1]; vb = P.bg[2];
                } else if (P.algorithm == RT_ALGO_ONOFF) {
                    vr = P.fg[0]; vg = P.fg[1]; vb = P.fg[2];   // render.rs:61-66
                } else {
                    const DShape* S;
                    const DMaterial* M;
                    if (scene_lds) { S = lds_shapes + best; M = lds_mats + best; }
                    else { S = g_shapes + best; M = g_materials + S->material; }
                    const bool want_uv = S->needs_uv != 0u;
                    const HitRec h = hit_record<AFFINE>(S, ro, rd, t, want_uv);
                    Vec3 hc = pigment_color(&M->brdf_pigment, texels, h.u, h.v);
                    const Vec3 em = pigment_color(&M->emitted, texels, h.u, h.v);
                    if (P.algorithm == RT_ALGO_FLAT) {
                        vr = em.x + hc.x; vg = em.y + hc.y; vb = em.z + hc.z;   // render.rs:83-84
                    } else {
                        // PathTracer::solve (render.rs:146-171)
                        const float lum = rt_maxf(hc.x, rt_maxf(hc.y, hc.z));
                        bool alive = true;
                        if (depth >= P.rr_limit) {
                            const float rq = rt_maxf(1.0f - lum, 0.05f);
                            if (pcg_random_float(pstate, pinc) > rq) {
                                const float s = 1.0f / (1.0f - rq);
                                hc.x = hc.x * s; hc.y = hc.y * s; hc.z = hc.z * s;
                            } else {
                                vr = em.x; vg = em.y; vb = em.z;
                                alive = false;
                            }
                        }
                        if (alive) {
                            if (lum > 0.0f && n_rays_per_hit != 0u) {
                                if (depth + 1u > P.max_depth) {
                                    // every child returns black without an intersection (render.rs:137-139)
                                    // but its two scatter draws are consumed (render.rs:161).
                                    pstate = pstate * P.skip_mult + pinc * P.skip_plus;
                                    const float cr = 0.0f + hc.x * 0.0f, cg = 0.0f + hc.y * 0.0f, cb = 0.0f + hc.z * 0.0f;
                                    vr = em.x + cr * P.inv_num_of_rays;
                                    vg = em.y + cg * P.inv_num_of_rays;
                                    vb = em.z + cb * P.inv_num_of_rays;
                                } else {
                                    // this hit becomes the current level; save the previous one
                                    if (lvl > 0u) {
                                        const uint32_t s = lvl - 1u;
                                        stack.st(s, 0, hc_r); stack.st(s, 1, hc_g); stack.st(s, 2, hc_b);
                                        stack.st(s, 3, em_r); stack.st(s, 4, em_g); stack.st(s, 5, em_b);
                                        stack.st(s, 6, cu_r); stack.st(s, 7, cu_g); stack.st(s, 8, cu_b);
                                        stack.st(s, 9, cw.x); stack.st(s, 10, cw.y); stack.st(s, 11, cw.z);
                                        stack.st(s, 12, cn.x); stack.st(s, 13, cn.y); stack.st(s, 14, cn.z);
                                        stack.st(s, 15, cin.x); stack.st(s, 16, cin.y); stack.st(s, 17, cin.z);
                                        stack.st(s, 18, __uint_as_float((kcount << 1) | cur_brdf));
                                    }
                                    hc_r = hc.x; hc_g = hc.y; hc_b = hc.z;
                                    em_r = em.x; em_g = em.y; em_b = em.z;
                                    cu_r = 0.f; cu_g = 0.f; cu_b = 0.f;
                                    kcount = 0;
                                    cur_brdf = M->brdf;
                                    cw = h.world_point;
                                    cn = h.normal;
                                    cin = rd;          // hit.ray.dir (render.rs:162)
                                    lvl += 1u;         // level of the children
                                    leaf = false;
                                }
                            } else {
                                // no children: emitted + 0 * (1/n) (render.rs:157, :171)
                                vr = em.x + 0.0f * P.inv_num_of_rays;
                                vg = em.y + 0.0f * P.inv_num_of_rays;
                                vb = em.z + 0.0f * P.inv_num_of_rays;
                            }
                        }
                    }
                }
            }

            // ---- return: fold the leaf value into the open levels (render.rs:167-171) ----
            if (leaf) {
                for (;;) {
                    if (lvl == 0u) {
                        acc_r = acc_r + vr; acc_g = acc_g + vg; acc_b = acc_b + vb;   // imagetracer.rs:119
                        has_ray = false;
                        break;
                    }
                    cu_r = cu_r + hc_r * vr; cu_g = cu_g + hc_g * vg; cu_b = cu_b + hc_b * vb;
                    kcount += 1u;
                    if (kcount < n_rays_per_hit) break;
                    vr = em_r + cu_r * P.inv_num_of_rays;
                    vg = em_g + cu_g * P.inv_num_of_rays;
                    vb = em_b + cu_b * P.inv_num_of_rays;
                    lvl -= 1u;
                    if (lvl > 0u) {
                        const uint32_t s = lvl - 1u;
                        hc_r = stack.ld(s, 0); hc_g = stack.ld(s, 1); hc_b = stack.ld(s, 2);
                        em_r = stack.ld(s, 3); em_g = stack.ld(s, 4); em_b = stack.ld(s, 5);
                        cu_r = stack.ld(s, 6); cu_g = stack.ld(s, 7); cu_b = stack.ld(s, 8);
                        cw.x = stack.ld(s, 9); cw.y = stack.ld(s, 10); cw.z = stack.ld(s, 11);
                        cn.x = stack.ld(s, 12); cn.y = stack.ld(s, 13); cn.z = stack.ld(s, 14);
                        cin.x = stack.ld(s, 15); cin.y = stack.ld(s, 16); cin.z = stack.ld(s, 17);
                        const uint32_t kb = __float_as_uint(stack.ld(s, 18));
                        kcount = kb >> 1;
                        cur_brdf = kb & 1u;
                    }
                }
            }

            // ---- next child of the current level: BRDF::scatter_ray (render.rs:160-166) ----
            if (has_ray) {
                const float rf0 = pcg_random_float(pstate, pinc);
                const float rf1 = pcg_random_float(pstate, pinc);
                if (cur_brdf == RT_BRDF_DIFFUSE) {
                    rd = diffuse_scatter_dir(rf0, rf1, cn);
                    rtmin = 1.0e-3f;                  // material.rs:181
                    n_diffuse += 1u;
                } else {
                    rd = specular_scatter_dir(cin, cn);
                    rtmin = 1e-5f;                    // Ray::default (ray.rs:50)
                    n_specular += 1u;
                }
                ro = cw;
                rtmax = __builtin_inff();
            }
        }
    }

    // ---- counters: wave reduction, one atomic per wave and counter ----
    if (P.counters) {
        uint32_t c[4] = {n_rays, n_samples, n_diffuse, n_specular};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned long long v = c[j];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0 && v != 0ull) atomicAdd(P.counters + j, v);
        }
    }
}

// World::ray_intersection for a batch of rays, full HitRecord (src/world.rs:32-50).
template <bool AFFINE>
__global__ __launch_bounds__(RT_BLOCK) void k_intersect(const DShape* __restrict__ shapes, uint32_t n_shapes,
                                                        const rt_ray* __restrict__ rays, uint64_t n, rt_hit* __restrict__ hits) {
    const uint64_t i = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const rt_ray r = rays[i];
    const Vec3 ro = {r.origin[0], r.origin[1], r.origin[2]};
    const Vec3 rd = {r.dir[0], r.dir[1], r.dir[2]};
    float t;
    const int best = closest_hit<AFFINE>(shapes, n_shapes, ro, rd, r.tmin, r.tmax, t);
    rt_hit h;
    h.hit = 0; h.shape = 0; h.t = 0.f;
    h.world_point[0] = h.world_point[1] = h.world_point[2] = 0.f;
    h.normal[0] = h.normal[1] = h.normal[2] = 0.f;
    h.uv[0] = h.uv[1] = 0.f;
    if (best >= 0) {
        const HitRec rec = hit_record<AFFINE>(shapes + best, ro, rd, t, true);
        h.hit = 1;
        h.shape = (uint32_t)best;
        h.t = t;
        h.world_point[0] = rec.world_point.x; h.world_point[1] = rec.world_point.y; h.world_point[2] = rec.world_point.z;
        h.normal[0] = rec.normal.x; h.normal[1] = rec.normal.y; h.normal[2] = rec.normal.z;
        h.uv[0] = rec.u; h.uv[1] = rec.v;
    }
    hits[i] = h;
}

// Pcg::new + n draws per seed (src/random.rs:26-43).
__global__ __launch_bounds__(RT_BLOCK) void k_pcg(const uint64_t* __restrict__ init_state, const uint64_t* __restrict__ init_seq,
                                                  uint64_t n, uint32_t n_draws, uint32_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    uint64_t s, inc;
    pcg_new(init_state[i], init_seq[i], s, inc);
    for (uint32_t k = 0; k < n_draws; ++k) out[i * n_draws + k] = pcg_random(s, inc);
}

// include/rt_strict_math.h on the device.
__global__ __launch_bounds__(RT_BLOCK) void k_math(int fn, const float* __restrict__ a, const float* __restrict__ b, uint64_t n,
                                                   float* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    float r;
    switch (fn) {
        case 0: r = rt_sinf(a[i]); break;
        case 1: r = rt_cosf(a[i]); break;
        case 2: r = rt_atan2f(a[i], b[i]); break;
        case 3: r = rt_acosf(a[i]); break;
        case 4: r = a[i] / b[i]; break;
        default: r = __builtin_sqrtf(a[i]); break;
    }
    out[i] = r;
}

// Row-tile partition inverse: image pixel -> (part, local pixel) -> copy.
__global__ __launch_bounds__(RT_BLOCK) void k_unpack(const float* __restrict__ parts, uint64_t part_stride_pixels, uint32_t n_parts,
                                                     uint32_t tile_rows, uint32_t width, uint32_t height, float* __restrict__ image) {
    const uint64_t i = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x;   // float index in the image
    const uint64_t n = 3ull * width * height;
    if (i >= n) return;
    const uint64_t pix = i / 3ull;
    const uint32_t ch = (uint32_t)(i - pix * 3ull);
    const uint32_t row = (uint32_t)(pix / width);
    const uint32_t col = (uint32_t)(pix - (uint64_t)row * width);
    const uint32_t tile = row / tile_rows;
    const uint32_t part = tile % n_parts;
    const uint32_t lrow = (tile / n_parts) * tile_rows + (row - tile * tile_rows);
    image[i] = parts[3ull * ((uint64_t)part * part_stride_pixels + (uint64_t)lrow * width + col) + ch];
}

}  // namespace rtd

#include "rt_host.inc"
