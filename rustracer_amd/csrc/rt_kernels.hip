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

#include <type_traits>

#include "rt_device.h"

#pragma clang fp contract(off)

namespace rtd {

#define RT_BLOCK 256
#define RT_CHUNK 64u
#define RT_NCLASS 3          // 0 open specular hit, 1 open diffuse hit, 2 leaf (one ray)
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
    // cost-ordered pixel queue built by k_classify (path tracer only; sorted == 0: identity order)
    uint32_t sorted;
    const uint32_t* class_counts;    // [RT_NCLASS] pixels per cost class
    const uint32_t* order;           // [RT_NCLASS][n_local] local pixel indices of each class
    const uint64_t* pix_mstate;      // [n_local] master PCG state at each pixel's first jitter draw
    uint32_t idle_min;               // refill when at least this many teams of a wave are idle
    uint64_t surv_mult[16], surv_geom[16];   // PCG advance by 2*num_of_rays*s draws, s = 0..15
    uint32_t surv_spec;              // 1: enumerate survivor counts at the last traced level
    // diagnostics (RUSTRACER_DEBUG_WAVES): per wave {t_start, t_end (100 MHz), rounds, active-lane rounds, refills}
    unsigned long long* debug_waves;
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


// Master PCG of ImageTracer::all_rays at global pixel p: Pcg::new(init_state, init_seq) advanced by
// p*(2*aa^2+1) draws (imagetracer.rs:63-81), by binary LCG jump-ahead.
__device__ __forceinline__ uint64_t master_state_at(const RenderParams& P, uint32_t p) {
    uint64_t s = P.master_state0;
    for (uint32_t i = 0; p != 0u; ++i, p >>= 1) {
        const uint64_t s2 = s * P.pix_mult[i] + P.pix_plus[i];
        s = (p & 1u) ? s2 : s;
    }
    return s;
}

// local pixel -> global (row, col) through the row-tile partition
__device__ __forceinline__ void local_to_global(const RenderParams& P, uint32_t q, uint32_t& grow, uint32_t& gcol) {
    const uint32_t lrow = q / P.width;
    gcol = q - lrow * P.width;
    const uint32_t tile = lrow / P.tile_rows;
    grow = (tile * P.n_parts + P.part) * P.tile_rows + (lrow - tile * P.tile_rows);
}

// ImageTracer::fire_ray + Camera::fire_ray for sample `samp` of pixel (grow, gcol)
// (imagetracer.rs:52-56, :67-75; camera.rs:68-75, :132-143); advances the master PCG by two draws.
template <bool AFFINE>
__device__ __forceinline__ void primary_ray(const RenderParams& P, uint32_t grow, uint32_t gcol, uint32_t samp,
                                            uint64_t& mstate, Vec3& ro, Vec3& rd) {
    const uint32_t sub_row = samp / P.aa;
    const uint32_t sub_col = samp - sub_row * P.aa;
    const float rf0 = pcg_random_float(mstate, P.master_inc);
    const float rf1 = pcg_random_float(mstate, P.master_inc);
    const float u_pixel = ((float)sub_row + rf0) / (float)P.aa;
    const float v_pixel = ((float)sub_col + rf1) / (float)P.aa;
    const float u = ((float)gcol + u_pixel) / (float)P.width;
    const float v = 1.0f - ((float)grow + v_pixel) / (float)P.height;
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
}

// generate + intersect + compact (path tracer): one lane per pixel fires the pixel's first camera
// ray, finds its closest hit and files the pixel under a cost class — open specular hit, open
// diffuse hit, or leaf — by wave64 ballot + prefix rank, one atomic per wave and class.  k_render
// then drains the classes heaviest first.  Also stores each pixel's master-PCG state so that the
// render kernel does not repeat the jump-ahead.  Purely a scheduling aid: radiance never depends
// on the order pixels are processed in.
template <bool AFFINE>
__global__ __launch_bounds__(RT_BLOCK) void k_classify(const RenderParams P, const DShape* __restrict__ g_shapes,
                                                       const DMaterial* __restrict__ g_materials,
                                                       const float* __restrict__ texels,
                                                       uint64_t* __restrict__ pix_mstate, uint32_t* __restrict__ order,
                                                       uint32_t* __restrict__ class_counts) {
    const uint32_t q = blockIdx.x * RT_BLOCK + threadIdx.x;
    const bool valid = q < P.n_local;
    uint32_t cls = RT_NCLASS;   // none
    if (valid) {
        uint32_t grow, gcol;
        local_to_global(P, q, grow, gcol);
        uint64_t ms = master_state_at(P, grow * P.width + gcol);
        pix_mstate[q] = ms;
        Vec3 ro, rd;
        primary_ray<AFFINE>(P, grow, gcol, 0u, ms, ro, rd);
        float t;
        const int best = closest_hit<AFFINE>(g_shapes, P.n_shapes, ro, rd, 1e-5f, __builtin_inff(), t);
        cls = 2u;
        if (best >= 0 && P.num_of_rays != 0u && P.max_depth >= 1u) {
            const DShape* S = g_shapes + best;
            const DMaterial* M = g_materials + S->material;
            const HitRec h = hit_record<AFFINE>(S, ro, rd, t, S->needs_uv != 0u);
            const Vec3 hc = pigment_color(&M->brdf_pigment, texels, h.u, h.v);
            if (rt_maxf(hc.x, rt_maxf(hc.y, hc.z)) > 0.0f) cls = (M->brdf == RT_BRDF_SPECULAR) ? 0u : 1u;
        }
    }
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (uint32_t c = 0; c < RT_NCLASS; ++c) {
        const unsigned long long m = __ballot(cls == c);
        if (m == 0ull) continue;
        const uint32_t cnt = (uint32_t)__popcll(m);
        uint32_t base = 0;
        const uint32_t first = (uint32_t)__builtin_ctzll(m);
        if (lane == first) base = atomicAdd(class_counts + c, cnt);
        base = __shfl(base, first);
        if (cls == c) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            order[c * P.n_local + base + rank] = q;
        }
    }
}

// (A^m, sum_{j<m} A^j) for m = 0..63: per-lane PCG skip-ahead of the speculated siblings.
struct JumpTable { uint64_t mult[64]; uint64_t geom[64]; };
constexpr JumpTable make_jump_table() {
    JumpTable t{};
    uint64_t m = 1, g = 0;
    for (int i = 0; i < 64; ++i) {
        t.mult[i] = m;
        t.geom[i] = g;
        g = g * RT_PCG_MULT + 1ull;
        m = m * RT_PCG_MULT;
    }
    return t;
}
__constant__ JumpTable c_jump = make_jump_table();

// MODE 0: image render (pixel queue).  MODE 1: batch Renderer::solve on caller-supplied (ray, pcg).
// NS > 0: the scene has exactly NS shapes (affine): the closest-hit scan is fully unrolled with the
//         inverse matrices held in SGPRs for the whole launch.
// T:      team size (1, 2, 4, 8, 16 lanes).  A team owns one pixel; its lanes hold identical copies
//         of the pixel's recursion state and trace up to T SIBLING rays of the open hit per round.
//         Sibling j's scatter draws sit at PCG offset j*(2+pred) past the level's current state
//         provided every earlier sibling is a leaf that consumed exactly `pred` further draws
//         (0 below the Russian-roulette depth, 1 at or beyond it).  After the round the team keeps
//         the longest prefix for which that held (ballot + count-trailing-zeros), folds its
//         radiances into the level in sibling order, adopts the PCG state of the last kept lane,
//         and re-speculates the rest.  Draw order and every f32 operation equal the sequential
//         recursion of PathTracer::solve (src/render.rs:136-172); only the latency changes.
// The scene tables are separate __restrict__ const arguments so that wave-uniform reads of them
// are scalar (SMEM) loads.
template <bool AFFINE, bool STACK_LDS, int MODE, int NS, int T, int T2>
__global__ __launch_bounds__(RT_BLOCK) void k_render(const RenderParams P, const DShape* __restrict__ g_shapes,
                                                     const DMaterial* __restrict__ g_materials,
                                                     const float* __restrict__ texels) {
    static_assert(T == 1 || T == 2 || T == 4 || T == 8 || T == 16, "team size");
    static_assert(T2 == 0 || ((T2 == 2 || T2 == 4 || T2 == 8 || T2 == 16) && T == 1 && MODE == 0 && STACK_LDS), "tail team size");
    extern __shared__ float4 lds_raw[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t gtid = blockIdx.x * RT_BLOCK + tid;
    constexpr bool ANY_TEAMS = (T > 1) || (T2 > 1);

    // ---- LDS: [jump table (T>1)][DShape x n][DMaterial x n (one per shape)][recursion stack] ----
    uint64_t* lds_jump = reinterpret_cast<uint64_t*>(lds_raw);
    constexpr uint32_t JUMP_BYTES = ANY_TEAMS ? (uint32_t)sizeof(JumpTable) + 256u : 0u;   // + survivor tables
    const bool scene_lds = (NS > 0) ? true : (P.scene_in_lds != 0u);
    const uint32_t n_staged = scene_lds ? P.n_shapes : 0u;
    DShape* lds_shapes = reinterpret_cast<DShape*>(reinterpret_cast<unsigned char*>(lds_raw) + JUMP_BYTES);
    DMaterial* lds_mats = reinterpret_cast<DMaterial*>(lds_shapes + n_staged);
    float* lds_stack = reinterpret_cast<float*>(lds_mats + n_staged);
    if (ANY_TEAMS) {
        const uint64_t* src = reinterpret_cast<const uint64_t*>(&c_jump);
        for (uint32_t i = tid; i < 128u; i += RT_BLOCK) lds_jump[i] = src[i];
        if (tid < 16u) { lds_jump[128u + tid] = P.surv_mult[tid]; lds_jump[144u + tid] = P.surv_geom[tid]; }
    }
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
    }
    if (ANY_TEAMS || scene_lds) __syncthreads();

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
    const uint32_t n_children = P.num_of_rays;

    // ---- team state, replicated in every lane of the team ----
    uint32_t q = RT_INVALID;        // local pixel (MODE 0) / batch index (MODE 1) owned by the team
    uint32_t samp = 0;              // samples of q already started
    uint64_t mstate = 0;            // master PCG state at the next jitter draw of q
    uint64_t pstate = 0, pinc = 1;  // the pixel's PCG (state before this round's draws)
    float acc_r = 0.f, acc_g = 0.f, acc_b = 0.f;
    uint32_t grow = 0, gcol = 0;    // global row / column of q
    bool busy = false;              // a top-level solve of q is in flight
    Vec3 ro = {0.f, 0.f, 0.f}, rd = {1.f, 0.f, 0.f};   // the level-0 ray of that solve
    float rtmin = 1e-5f, rtmax = __builtin_inff();
    uint32_t lvl = 0;               // recursion level of the rays traced next (0 = the level-0 ray)
    uint32_t base_depth = 0;        // Ray.depth of the level-0 ray
    // current level = the open hit at level lvl-1
    float hc_r = 0.f, hc_g = 0.f, hc_b = 0.f, em_r = 0.f, em_g = 0.f, em_b = 0.f;
    float cu_r = 0.f, cu_g = 0.f, cu_b = 0.f;
    uint32_t kcount = 0, cur_brdf = 0;   // kcount: children of the current level already started
    Vec3 cw = {0.f, 0.f, 0.f}, cn = {0.f, 0.f, 1.f}, cin = {1.f, 0.f, 0.f};
    // counters (kept by the team leader)
    uint32_t n_rays = 0, n_samples = 0, n_diffuse = 0, n_specular = 0;

    // wave-private slice of the pixel queue
    uint32_t wcur = 0, wend = 0;
    bool exhausted = false;
    uint32_t dbg_pix_rounds = 0, dbg_max_pix_rounds = 0;
    // diagnostics
    unsigned long long dbg_t0 = 0;
    uint32_t dbg_rounds = 0, dbg_lanes = 0;
    if (P.debug_waves) dbg_t0 = __builtin_amdgcn_s_memrealtime();
    // cost-ordered queue: position h of the concatenated class lists -> local pixel
    uint32_t cls_end0 = 0, cls_end1 = 0;
    if (MODE == 0 && P.sorted) {
        cls_end0 = P.class_counts[0];
        cls_end1 = cls_end0 + P.class_counts[1];
    }

    // The round loop, instantiated per team size TT.  stop_live > 0: return true (instead of running
    // to completion) once the queue is exhausted and at most stop_live pixels are left in the wave.
    auto run_rounds = [&](auto team_tag, const uint32_t stop_live) -> bool {
    constexpr int TT = decltype(team_tag)::value;
    const uint32_t jj = (TT > 1) ? (lane & (uint32_t)(TT - 1)) : 0u;   // candidate slot of this lane
    const uint32_t team0 = lane & ~(uint32_t)(TT - 1);                 // first lane of the team
    const bool leader = (jj == 0u);
    for (;;) {
        // =====================================================================================
        // generate / compact: teams without a solve in flight take the next sample or pixel
        // =====================================================================================
        const unsigned long long idle_leaders = __ballot(!busy && leader);
        const unsigned long long busy_any = __ballot(busy);
        if (idle_leaders != 0ull && ((uint32_t)__popcll(idle_leaders) >= P.idle_min || busy_any == 0ull)) {
            bool want_pixel = false;
            if (!busy) {
                if (q != RT_INVALID && samp == aa2) {
                    // pixel (or solve item) finished: color * (1 / aa^2)  (imagetracer.rs:121)
                    if (leader) {
                        float* o = P.out + 3ull * q;
                        if (MODE == 0) {
                            o[0] = acc_r * P.inv_samples;
                            o[1] = acc_g * P.inv_samples;
                            o[2] = acc_b * P.inv_samples;
                        } else {
                            o[0] = acc_r; o[1] = acc_g; o[2] = acc_b;
                            P.io_pcgs[q].state = pstate;
                        }
                    }
                    q = RT_INVALID;
                    if (dbg_pix_rounds > dbg_max_pix_rounds) dbg_max_pix_rounds = dbg_pix_rounds;
                    dbg_pix_rounds = 0;
                }
                want_pixel = (q == RT_INVALID) && !exhausted;
            }
            // wave64 ballot + prefix rank hand out consecutive queue entries to the idle teams
            unsigned long long wantmask = __ballot(want_pixel && leader);
            bool newpix = false;
            bool lead_want = want_pixel && leader;
            while (wantmask != 0ull) {
                if (wcur == wend) {
                    // heavy classes are handed out in small chunks (longest-processing-time first)
                    constexpr uint32_t TEAMS = 64u / (uint32_t)TT;     // pixels a wave holds at once
                    uint32_t chunk = TEAMS;
                    if (P.sorted) {
                        chunk = (wend < cls_end0) ? 4u : (wend < cls_end1) ? 16u : RT_CHUNK;
                        if (chunk > TEAMS) chunk = TEAMS;
                    }
                    uint32_t b = 0;
                    if (lane == 0) b = atomicAdd(P.queue_head, chunk);
                    b = __shfl(b, 0);
                    if (b >= P.n_local) { exhausted = true; break; }
                    wcur = b;
                    wend = (b + chunk < P.n_local) ? b + chunk : P.n_local;
                }
                const uint32_t avail = wend - wcur;
                const uint32_t wanted = (uint32_t)__popcll(wantmask);
                const uint32_t cnt = wanted < avail ? wanted : avail;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(wantmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wantmask, 0u));
                if (lead_want && rank < cnt) {
                    const uint32_t h = wcur + rank;
                    q = h;
                    if (MODE == 0 && P.sorted) {
                        if (h < cls_end0) q = P.order[h];
                        else if (h < cls_end1) q = P.order[P.n_local + (h - cls_end0)];
                        else q = P.order[2u * P.n_local + (h - cls_end1)];
                    }
                    lead_want = false;
                    newpix = true;
                }
                wcur += cnt;
                wantmask = __ballot(lead_want);
            }
            if (TT > 1) {
                q = __shfl(q, team0);
                newpix = __shfl((int)newpix, team0) != 0;
            }
            if (!busy && q != RT_INVALID) {
                if (MODE == 0) {
                    if (newpix) {
                        local_to_global(P, q, grow, gcol);
                        // master PCG at this pixel: jump ahead p*(2*aa^2+1) draws (imagetracer.rs:63-81)
                        uint64_t s;
                        if (P.sorted) s = P.pix_mstate[q];
                        else s = master_state_at(P, grow * P.width + gcol);
                        mstate = s;
                        // seq = the u32 drawn after the pixel's 2*aa^2 jitter draws (imagetracer.rs:79)
                        const uint64_t sq = s * P.seq_mult + P.seq_plus;
                        const uint64_t seq = (uint64_t)pcg_output(sq);
                        pcg_new(P.init_state, seq, pstate, pinc);   // imagetracer.rs:117
                        samp = 0;
                        acc_r = 0.f; acc_g = 0.f; acc_b = 0.f;
                    }
                    primary_ray<AFFINE>(P, grow, gcol, samp, mstate, ro, rd);
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
                if (leader) n_samples += 1;
                lvl = 0;
                busy = true;
            }
        }
        if (__ballot(busy) == 0ull) return false;  // queue empty and every team retired
        if (stop_live != 0u && exhausted) {
            const uint32_t live = (uint32_t)__popcll(__ballot(busy || (q != RT_INVALID && samp < aa2)));
            if (live <= stop_live) return true;
        }
        // =====================================================================================
        // this round's candidate rays: the level-0 ray, or up to T siblings of the open hit
        // =====================================================================================
        const uint32_t depth = base_depth + lvl;     // Ray.depth of the rays traced this round
        // Candidate (cj, cs) of this lane: sibling kcount+cj, assuming cs of the earlier siblings of this
        // round consumed the long draw count.  Below the last traced level cs = 0 and cj = jj.  At the
        // last level (depth == max_depth: a hit cannot open, it consumes pred or pred + 2n draws) the
        // team enumerates all (cj, cs <= cj) in triangular order, so several siblings resolve per round
        // whatever their roulette outcomes were.
        const bool last_level = (TT > 1) && (lvl > 0u) && (depth == P.max_depth) && (P.surv_spec != 0u);
        uint32_t cj = jj, cs = 0u;
        if (last_level) {
            cj = (jj >= 15u) ? 5u : (jj >= 10u) ? 4u : (jj >= 6u) ? 3u : (jj >= 3u) ? 2u : (jj >= 1u) ? 1u : 0u;
            cs = jj - ((cj * (cj + 1u)) >> 1);
        }
        uint32_t n_rem = 0;          // rays this level still has to start
        if (busy) n_rem = (lvl == 0u) ? 1u : (n_children - kcount);
        const bool act = busy && ((lvl == 0u) ? (jj == 0u) : (cj < n_rem));
        if (P.debug_waves) {
            dbg_rounds += 1u; dbg_lanes += (uint32_t)__popcll(__ballot(act));
            if (busy) dbg_pix_rounds += 1u;
        }
        // draws an earlier sibling is assumed to consume after its two scatter draws
        const uint32_t pred = (lvl > 0u && depth >= P.rr_limit) ? 1u : 0u;

        uint64_t st = pstate;        // this lane's PCG, advanced through its own ray's draws
        bool leaf = true;            // the ray returns a value this round (no open hit below it)
        bool traced = false;         // World::ray_intersection was called
        uint32_t consumed = 0;       // draws after the scatter draws
        float vr = 0.f, vg = 0.f, vb = 0.f;
        // open-hit data if !leaf
        Vec3 ihc = {0.f, 0.f, 0.f}, iem = {0.f, 0.f, 0.f}, iwp = {0.f, 0.f, 0.f}, inr = {0.f, 0.f, 1.f};
        Vec3 idir = {1.f, 0.f, 0.f};
        uint32_t ibrdf = 0;

        if (act) {
            Vec3 o, d;
            float tmin, tmax;
            if (lvl == 0u) {
                o = ro; d = rd; tmin = rtmin; tmax = rtmax;
            } else {
                // BRDF::scatter_ray of sibling kcount+jj (render.rs:160-166)
                if (TT > 1) {
                    const uint32_t m = cj * (2u + pred);
                    st = st * lds_jump[m] + pinc * lds_jump[64u + m];
                    if (last_level) st = st * lds_jump[128u + cs] + pinc * lds_jump[144u + cs];
                }
                const float rf0 = pcg_random_float(st, pinc);
                const float rf1 = pcg_random_float(st, pinc);
                if (cur_brdf == RT_BRDF_DIFFUSE) {
                    d = diffuse_scatter_dir(rf0, rf1, cn);
                    tmin = 1.0e-3f;                   // material.rs:181
                } else {
                    d = specular_scatter_dir(cin, cn);
                    tmin = 1e-5f;                     // Ray::default (ray.rs:50)
                }
                o = cw;
                tmax = __builtin_inff();
            }

            // ---- intersect + shade one node ----
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
                        const Vec3 so = xform_point<true>(sinv[i], o.x, o.y, o.z);
                        const Vec3 sd = xform_vector(sinv[i], d.x, d.y, d.z);
                        float ti = 0.0f;
                        bool hit;
                        if (skind[i] == RT_SHAPE_SPHERE) hit = sphere_t(so, sd, tmin, tmax, ti);
                        else hit = plane_t(so, sd, tmin, tmax, ti);
                        if (hit && (best < 0 || ti < t)) { best = i; t = ti; }
                    }
                } else {
                    best = closest_hit<AFFINE>(g_shapes, P.n_shapes, o, d, tmin, tmax, t);
                }
                traced = true;
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
                    const HitRec h = hit_record<AFFINE>(S, o, d, t, want_uv);
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
                            consumed = 1u;
                            if (pcg_random_float(st, pinc) > rq) {
                                const float s = 1.0f / (1.0f - rq);
                                hc.x = hc.x * s; hc.y = hc.y * s; hc.z = hc.z * s;
                            } else {
                                vr = em.x; vg = em.y; vb = em.z;
                                alive = false;
                            }
                        }
                        if (alive) {
                            if (lum > 0.0f && n_children != 0u) {
                                if (depth + 1u > P.max_depth) {
                                    // every child returns black without an intersection (render.rs:137-139)
                                    // but its two scatter draws are consumed (render.rs:161).
                                    st = st * P.skip_mult + pinc * P.skip_plus;
                                    consumed += 2u * n_children;
                                    const float cr = 0.0f + hc.x * 0.0f, cg = 0.0f + hc.y * 0.0f, cb = 0.0f + hc.z * 0.0f;
                                    vr = em.x + cr * P.inv_num_of_rays;
                                    vg = em.y + cg * P.inv_num_of_rays;
                                    vb = em.z + cb * P.inv_num_of_rays;
                                } else {
                                    // an open hit: its children are traced in later rounds
                                    leaf = false;
                                    ihc = hc; iem = em; iwp = h.world_point; inr = h.normal;
                                    idir = d;          // hit.ray.dir (render.rs:162)
                                    ibrdf = M->brdf;
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
        }

        // =====================================================================================
        // resolve: keep the longest prefix of siblings whose PCG offsets were right
        // =====================================================================================
        if (busy) {
            uint32_t n_valid = 1u, last = 0u;
            bool last_leaf = leaf;
            uint32_t traced_valid = traced ? 1u : 0u;
            // fold the kept leaves into the level, in sibling order (render.rs:168 / imagetracer.rs:119)
            float pr, pg, pb;
            if (lvl == 0u) { pr = vr; pg = vg; pb = vb; }
            else { pr = hc_r * vr; pg = hc_g * vg; pb = hc_b * vb; }
            float tr = (lvl == 0u) ? acc_r : cu_r, tg = (lvl == 0u) ? acc_g : cu_g, tb_ = (lvl == 0u) ? acc_b : cu_b;
            if (TT > 1) {
                constexpr uint32_t tmask = (1u << TT) - 1u;
                const uint32_t b_leaf = (uint32_t)(__ballot(act && leaf) >> team0) & tmask;
                const uint32_t b_traced = (uint32_t)(__ballot(traced) >> team0) & tmask;
                if (!last_level) {
                    // linear layout: keep siblings up to and including the first that is not a
                    // leaf with the predicted draw count
                    const bool bad = act && !(leaf && (lvl == 0u || consumed == pred));
                    const uint32_t tb = (uint32_t)(__ballot(bad) >> team0) & tmask;
                    uint32_t n_cand = n_rem < (uint32_t)TT ? n_rem : (uint32_t)TT;
                    const uint32_t first_bad = tb ? (uint32_t)__builtin_ctz(tb) : n_cand;
                    n_valid = first_bad + 1u < n_cand ? first_bad + 1u : n_cand;
                    last = n_valid - 1u;
                    last_leaf = ((b_leaf >> last) & 1u) != 0u;
                    traced_valid = (uint32_t)__popc(b_traced & ((1u << n_valid) - 1u));
                    const uint32_t n_acc = n_valid - (last_leaf ? 0u : 1u);
#pragma unroll
                    for (int i = 0; i < TT; ++i) {
                        const float qr = __shfl(pr, team0 + i), qg = __shfl(pg, team0 + i), qb = __shfl(pb, team0 + i);
                        if ((uint32_t)i < n_acc) { tr = tr + qr; tg = tg + qg; tb_ = tb_ + qb; }
                    }
                } else {
                    // triangular layout: walk the siblings in order, following the survivor count
                    constexpr int JM = (TT >= 16) ? 6 : (TT >= 8) ? 4 : (TT >= 4) ? 3 : 2;
                    const uint32_t b_act = (uint32_t)(__ballot(act) >> team0) & tmask;
                    const uint32_t b_lo = (uint32_t)(__ballot(act && leaf && consumed == pred) >> team0) & tmask;
                    const uint32_t b_hi = (uint32_t)(__ballot(act && leaf && consumed == pred + 2u * n_children) >> team0) & tmask;
                    uint32_t ch[JM];
                    uint32_t surv = 0u;
                    bool cont = true;
                    n_valid = 0u;
                    traced_valid = 0u;
#pragma unroll
                    for (int j = 0; j < JM; ++j) {
                        const uint32_t idx = (uint32_t)((j * (j + 1)) >> 1) + surv;
                        const bool ok = cont && idx < (uint32_t)TT && ((b_act >> idx) & 1u) != 0u;
                        ch[j] = idx < (uint32_t)TT ? idx : 0u;
                        if (ok) {
                            n_valid = (uint32_t)j + 1u;
                            last = idx;
                            traced_valid += (b_traced >> idx) & 1u;
                        }
                        const bool hi = ((b_hi >> idx) & 1u) != 0u;
                        cont = ok && ((((b_lo >> idx) & 1u) != 0u) || hi);
                        if (ok && hi) surv += 1u;
                    }
                    last_leaf = ((b_leaf >> last) & 1u) != 0u;
                    const uint32_t n_acc = n_valid - (last_leaf ? 0u : 1u);
#pragma unroll
                    for (int j = 0; j < JM; ++j) {
                        const float qr = __shfl(pr, team0 + ch[j]), qg = __shfl(pg, team0 + ch[j]), qb = __shfl(pb, team0 + ch[j]);
                        if ((uint32_t)j < n_acc) { tr = tr + qr; tg = tg + qg; tb_ = tb_ + qb; }
                    }
                }
                pstate = __shfl(st, team0 + last);
            } else {
                if (last_leaf) { tr = tr + pr; tg = tg + pg; tb_ = tb_ + pb; }
                pstate = st;
            }
            if (lvl == 0u) { acc_r = tr; acc_g = tg; acc_b = tb_; }
            else { cu_r = tr; cu_g = tg; cu_b = tb_; }
            if (leader) {
                n_rays += traced_valid;
                if (lvl > 0u) {
                    if (cur_brdf == RT_BRDF_DIFFUSE) n_diffuse += n_valid; else n_specular += n_valid;
                }
            }
            if (lvl > 0u) kcount += n_valid;

            if (!last_leaf) {
                // the last kept sibling is an open hit: it becomes the current level
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
                if (TT > 1) {
                    const uint32_t src = team0 + last;
                    hc_r = __shfl(ihc.x, src); hc_g = __shfl(ihc.y, src); hc_b = __shfl(ihc.z, src);
                    em_r = __shfl(iem.x, src); em_g = __shfl(iem.y, src); em_b = __shfl(iem.z, src);
                    cw.x = __shfl(iwp.x, src); cw.y = __shfl(iwp.y, src); cw.z = __shfl(iwp.z, src);
                    cn.x = __shfl(inr.x, src); cn.y = __shfl(inr.y, src); cn.z = __shfl(inr.z, src);
                    cin.x = __shfl(idir.x, src); cin.y = __shfl(idir.y, src); cin.z = __shfl(idir.z, src);
                    cur_brdf = __shfl(ibrdf, src);
                } else {
                    hc_r = ihc.x; hc_g = ihc.y; hc_b = ihc.z;
                    em_r = iem.x; em_g = iem.y; em_b = iem.z;
                    cw = iwp; cn = inr; cin = idir;
                    cur_brdf = ibrdf;
                }
                cu_r = 0.f; cu_g = 0.f; cu_b = 0.f;
                kcount = 0;
                lvl += 1u;
            } else if (lvl == 0u) {
                busy = false;      // the level-0 ray returned: this solve is done
            } else {
                // levels whose children have all returned close and return to their parent
                while (kcount == n_children) {
                    const float wr = em_r + cu_r * P.inv_num_of_rays;   // render.rs:171
                    const float wg = em_g + cu_g * P.inv_num_of_rays;
                    const float wb = em_b + cu_b * P.inv_num_of_rays;
                    lvl -= 1u;
                    if (lvl == 0u) {
                        acc_r = acc_r + wr; acc_g = acc_g + wg; acc_b = acc_b + wb;   // imagetracer.rs:119
                        busy = false;
                        break;
                    }
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
                    cu_r = cu_r + hc_r * wr; cu_g = cu_g + hc_g * wg; cu_b = cu_b + hc_b * wb;   // render.rs:168
                }
            }
        }
    }

    };   // run_rounds

    if (T2 == 0) {
        (void)run_rounds(std::integral_constant<int, T>{}, 0u);
    } else if (run_rounds(std::integral_constant<int, 1>{}, 64u / (uint32_t)(T2 > 0 ? T2 : 1))) {
        // ---- tail: the queue is empty and at most 64/T2 pixels of this wave are unfinished.  Pack
        // them into teams of T2 lanes (each lane of a team gets a copy of the pixel's state and of its
        // LDS stack) and finish them with sibling / survivor speculation: the frame ends when the
        // longest per-pixel chain of dependent rays ends, and idle lanes are free. ----
        constexpr uint32_t TT2 = (uint32_t)(T2 > 0 ? T2 : 1);
        // retire finished pixels still waiting for a refill event
        if (!busy && q != RT_INVALID && samp == aa2) {
            float* o = P.out + 3ull * q;
            o[0] = acc_r * P.inv_samples; o[1] = acc_g * P.inv_samples; o[2] = acc_b * P.inv_samples;
            q = RT_INVALID;
            if (dbg_pix_rounds > dbg_max_pix_rounds) dbg_max_pix_rounds = dbg_pix_rounds;
            dbg_pix_rounds = 0;
        }
        const bool live = busy || (q != RT_INVALID);
        unsigned long long lm = __ballot(live);
        const uint32_t n_live = (uint32_t)__popcll(lm);
        const uint32_t tt = lane / TT2;
        for (uint32_t i = 0; i < tt; ++i) lm &= lm - 1ull;       // drop the tt lowest live lanes
        const bool has = tt < n_live;
        const uint32_t src = has ? (uint32_t)__builtin_ctzll(lm) : lane;
        // counters stay with their lane; everything else moves
        q = __shfl(q, src); samp = __shfl(samp, src);
        mstate = __shfl(mstate, src); pstate = __shfl(pstate, src); pinc = __shfl(pinc, src);
        acc_r = __shfl(acc_r, src); acc_g = __shfl(acc_g, src); acc_b = __shfl(acc_b, src);
        grow = __shfl(grow, src); gcol = __shfl(gcol, src);
        busy = __shfl((int)busy, src) != 0;
        ro.x = __shfl(ro.x, src); ro.y = __shfl(ro.y, src); ro.z = __shfl(ro.z, src);
        rd.x = __shfl(rd.x, src); rd.y = __shfl(rd.y, src); rd.z = __shfl(rd.z, src);
        rtmin = __shfl(rtmin, src); rtmax = __shfl(rtmax, src);
        lvl = __shfl(lvl, src); base_depth = __shfl(base_depth, src);
        hc_r = __shfl(hc_r, src); hc_g = __shfl(hc_g, src); hc_b = __shfl(hc_b, src);
        em_r = __shfl(em_r, src); em_g = __shfl(em_g, src); em_b = __shfl(em_b, src);
        cu_r = __shfl(cu_r, src); cu_g = __shfl(cu_g, src); cu_b = __shfl(cu_b, src);
        kcount = __shfl(kcount, src); cur_brdf = __shfl(cur_brdf, src);
        cw.x = __shfl(cw.x, src); cw.y = __shfl(cw.y, src); cw.z = __shfl(cw.z, src);
        cn.x = __shfl(cn.x, src); cn.y = __shfl(cn.y, src); cn.z = __shfl(cn.z, src);
        cin.x = __shfl(cin.x, src); cin.y = __shfl(cin.y, src); cin.z = __shfl(cin.z, src);
        dbg_pix_rounds = __shfl(dbg_pix_rounds, src);
        if (!has) { q = RT_INVALID; busy = false; }
        // LDS stack: every (slot, field) is read by all lanes before any lane overwrites it
        // (LDS operations of one wave execute in order)
        const float* sbase = lds_stack + (tid - lane + src);
        for (uint32_t i = 0; i < P.stack_slots * RT_NF; ++i) {
            const float v = sbase[i * RT_BLOCK];
            stack.base[i * RT_BLOCK] = v;
        }
        exhausted = true;
        (void)run_rounds(std::integral_constant<int, (T2 > 0 ? T2 : 1)>{}, 0u);
    }

    if (P.debug_waves && lane == 0) {
        unsigned long long* d = P.debug_waves + 5ull * (gtid >> 6);
        d[0] = dbg_t0; d[1] = __builtin_amdgcn_s_memrealtime(); d[2] = dbg_rounds; d[3] = dbg_lanes;
    }
    if (P.debug_waves) {
        uint32_t mx = dbg_max_pix_rounds;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_down(mx, off); mx = o > mx ? o : mx; }
        if (lane == 0) P.debug_waves[5ull * (gtid >> 6) + 4] = mx;
    }
    // ---- counters: wave reduction, one atomic per wave and counter ----
    if (P.counters) {
        uint32_t c[4] = {n_rays, n_samples, n_diffuse, n_specular};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t v = c[j];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0 && v != 0u) atomicAdd(P.counters + j, (unsigned long long)v);
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
