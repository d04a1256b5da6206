// rt_kernels.hip — gfx950 kernels of the render core.
//
// Wavefront path tracing over PATH-STATE queues.  The reference threads ONE Pcg through the whole
// recursion of PathTracer::solve (src/render.rs:136-172), so the draw order inside a pixel is
// depth-first and data dependent: a queue entry is therefore a pixel (its whole recursion state,
// exactly one ray in flight), never a single ray.  A frame is a short sequence of passes of one
// kernel, k_pass:
//
//   generate : pass 0 takes fresh pixels (master PCG of ImageTracer::all_rays reproduced by LCG
//              jump-ahead, src/imagetracer.rs:61-84); later passes take parked path states from
//              the SoA queue in HBM the previous pass wrote.
//   intersect: World::ray_intersection — wave-uniform scan of the shape table staged in LDS.
//   shade    : HitRecord of the winner, pigments, renderer logic, Russian roulette; advances the
//              pixel's depth-first recursion by one node, consuming PCG draws in reference order.
//   compact  : a lane whose pixel finished refills in place (wave64 ballot + prefix rank over the
//              wave's chunk of the queue); a pixel that used up the pass's round budget is PARKED:
//              its state is appended to the next pass's queue (ballot + prefix rank, one atomic per
//              wave).  The next pass therefore starts with full waves again, and with fewer of them.
//
// Passes whose queue is short run TEAMS: T = 2..64 lanes share one pixel and trace sibling rays
// of the open hit speculatively (see run_rounds); that shortens the per-pixel dependency chain
// with lanes that would otherwise idle.  Radiance never depends on pass budgets or team sizes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "rt_device.h"

#pragma clang fp contract(off)

namespace rtd {

#define RT_BLOCK 256
// Cost probes (diagnostics builds only, -DRT_DUP_BLOCK=n): block n of the FAST round body is executed twice on laundered
// inputs, its second result kept alive but unused, so the slow-down of the build IS that block's cost at the kernel's real
// occupancy and divergence.  1 scan  2 diffuse scatter direction  3 open-hit geometry  4 primary ray  5 two PCG draws
#ifdef RT_DUP_BLOCK
#define RT_DUP_IS(n) (RT_DUP_BLOCK == (n))
#else
#define RT_DUP_IS(n) false
#endif
#define RT_LAUNDER3(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z))
#define RT_KEEP3(v) asm volatile("" ::"v"((v).x), "v"((v).y), "v"((v).z))
#define RT_NF 16             // dwords per saved recursion level
#define RT_QHEAD 13          // dwords of a parked path before its current level
#define RT_INVALID 0xFFFFFFFFu
#define RT_MAX_PASSES 24
#define RT_TRI_J 22          // candidate table: siblings j < RT_TRI_J

// Per-frame control block in HBM (zeroed before pass 0).
struct PassCtl {
    uint32_t head[RT_MAX_PASSES];        // consumption head of pass p's input
    uint32_t count[RT_MAX_PASSES + 1];   // entries parked for pass p
    uint32_t pad[5];
    unsigned long long counters[5];      // [0] rays [1] samples [2] diffuse [3] specular [4] rays answered from a sibling's query
};
static_assert(sizeof(PassCtl) == 256, "PassCtl layout");

// Team tables (device memory, staged in LDS by team passes).  PCG advance by k draws:
// s' = s * mult + inc * geom.
struct TeamTables {
    uint64_t j2_mult[64], j2_geom[64];   // 2j draws
    uint64_t j3_mult[64], j3_geom[64];   // 3j draws
    uint64_t sv_mult[32], sv_geom[32];   // 2 * num_of_rays * s draws
    uint16_t fwd[64];                    // team lane k -> sibling j | survivors s << 8 (0xFFFF: none); inverse: tri_index
    uint8_t pad[256];
};
static_assert(sizeof(TeamTables) == 2944, "TeamTables layout");

// One frame of a multi-frame launch (rt_render_frames): its camera and its output buffer.
struct FrameEntry {
    float cam_m[16];                 // Camera transformation.m (src/camera.rs)
    uint32_t cam_kind;
    float cam_distance, cam_aspect;
    uint32_t pad;
    float* out;
    uint64_t pad2;
};
static_assert(sizeof(FrameEntry) == 96, "FrameEntry layout");

// Experiment (profiles/r03_dropped_open_hit_mailbox.txt; off in the shipped library): the geometry of a hit that opens --
// needed by one lane in eleven per round -- is posted to a workgroup mailbox in LDS and computed by ONE wave of the block
// over the compacted list, instead of by every wave for its few lanes.  Two workgroup barriers per round.
#ifndef RT_MAILBOX
#define RT_MAILBOX 0
#endif
#define RT_MB_SLOTS 40u          // postings per round and block; lanes beyond compute their hit themselves
#define RT_MB_FIELDS 16u         // in [0, 10): shape | flags, hit point (3), object-space direction (3), world direction (3);
                                 // out [10, 16): world point (3), child direction / normal (3) -- an area of their own: a wave may post the
                                 // next round's inputs while a slower one still reads this round's results
#define RT_MB_BYTES (16u + RT_MB_FIELDS * RT_MB_SLOTS * 4u)

#define RT_INLINE_FRAMES 8   // launches of up to this many frames carry their frame table in the kernarg

// Everything a pass needs, passed by value (kernarg -> SGPRs).
struct PassParams {
    uint32_t n_shapes;
    uint32_t scene_diag;             // every shape's invm is affine with a diagonal 3x3 in [2^-60, 2^60] (see diag_ray_ok)
    uint32_t mode;                   // 0: image render; 1: batch Renderer::solve on caller (ray, pcg)
    // frames of this launch (mode 0): item id = frame * n_per_frame + local pixel
    const FrameEntry* frames;        // device table, or NULL: the frames are in frames_inline
    uint32_t n_frames, n_per_frame;
    FrameEntry frames_inline[RT_INLINE_FRAMES];
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
    // partition (row tiles dealt round-robin); part_rotate: frame f of the launch renders part (part + f) % n_parts
    uint32_t part, n_parts, tile_rows;
    uint32_t part_rotate;
    uint32_t mg_parts[3];            // magic division by n_parts
    uint32_t out_in_place;           // FrameEntry.out is the FULL image: a pixel is stored at its global (row, column)
    // exact u32 division by invariant divisors (Granlund-Montgomery, host: magic_u32): width, n_per_frame, tile_rows
    uint32_t mg_width[3], mg_npf[3], mg_tile[3];
    uint32_t n_items;                // n_frames * (pixels of this part) / solve items
    uint32_t n_in0;                  // length of pass 0's (virtual) queue: n_items, or whole 64-pixel granules
    uint32_t perm_stride;            // 0, or granules per stratum of the pass-0 queue order (see perm_entry)
    uint32_t mix_from;               // FAST == 3: hand-outs from this granule of the queue on are mixed (runs from several strata)
    // solve mode
    const rt_ray* in_rays;
    rt_pcg* io_pcgs;
    // outputs (mode 1; mode 0 writes FrameEntry.out)
    float* out;
    // pass control
    uint32_t pass;                   // 0: fresh items; p > 0: path states parked by pass p-1
    uint32_t budget;                 // rounds a path may run in this pass before it is parked (0: no limit)
    uint32_t lanes_target;           // team size = lanes_target / queue length (power of two, <= max_team)
    uint32_t max_team;
    uint32_t trav_min_done;          // FAST over a hierarchy: a round's traversal ends once this many of the wave's rays have returned
                                     // (0: every round waits for all of them)
    uint32_t idle_div;               // refill once 1/idle_div of a wave's teams are idle
    uint32_t backlog;                // queue entries per team slot of the grid (>= 1): fewer, fuller waves
    PassCtl* ctl;
    DBvh bvh;                        // exact-winner hierarchy of a scene read from HBM (n_nodes == 0: linear scan)
    float* qbuf[2];                  // SoA path-state queues [field][entry]; pass p reads (p-1)&1, writes p&1
    uint32_t q_capacity;
    const TeamTables* tables;
    // recursion stack
    uint32_t stack_slots;            // max ancestors saved per lane
    float* gstack;                   // global stack [slot][field][global thread] when not in LDS
    uint32_t total_threads;
    // diagnostics (RUSTRACER_DEBUG_WAVES): per pass and wave {t_start, t_end (100 MHz), rounds, active-lane rounds, team}
    unsigned long long* debug_waves;
    uint32_t debug_stride;           // waves per pass in debug_waves
};

// Parameters only the generate / retire / park paths read are fetched from the kernarg segment at
// the point of use (scalar loads) instead of living in SGPRs for the whole launch: the round loop
// is long and every resident scalar costs spills.
typedef const __attribute__((address_space(4))) PassParams* ColdParams;
__device__ __forceinline__ ColdParams cold_params() {
    ColdParams p = (ColdParams)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// Frame table entry: in the kernarg segment for small launches (no upload), else in device memory.
__device__ __forceinline__ const FrameEntry* frame_entry(ColdParams C, uint32_t frame) {
    const FrameEntry* tab = C->frames;
    if (tab) return tab + frame;
    return reinterpret_cast<const FrameEntry*>(reinterpret_cast<uintptr_t>(&C->frames_inline[0])) + frame;
}

// Saved recursion levels: slot s of a lane is four float4 quads at [(s * 4 + quad) * stride] from the lane's base
// (LDS: stride = block size, lanes 16 B apart: conflict-free 128-bit accesses; global: stride = total threads).
template <bool STACK_LDS>
struct LevelStack {
    float4* base;      // LDS: &lds[tid]; global: &gstack[gtid]
    uint32_t stride;   // blockDim or total threads
    __device__ __forceinline__ void st4(uint32_t slot, uint32_t quad, float4 v) const { base[(slot * (RT_NF / 4) + quad) * stride] = v; }
    __device__ __forceinline__ float4 ld4(uint32_t slot, uint32_t quad) const { return base[(slot * (RT_NF / 4) + quad) * stride]; }
    // dword i = slot * RT_NF + field (the order in which a parked path stores its saved levels)
    __device__ __forceinline__ float& dword(uint32_t i) const { return reinterpret_cast<float*>(base + (i >> 2) * stride)[i & 3u]; }
};

// Master PCG of ImageTracer::all_rays at global pixel p: Pcg::new(init_state, init_seq) advanced by
// p*(2*aa^2+1) draws (imagetracer.rs:63-81), by binary LCG jump-ahead.
__device__ __forceinline__ uint64_t master_state_at(ColdParams P, uint32_t p) {
    uint64_t s = P->master_state0;
    for (uint32_t i = 0; p != 0u; ++i, p >>= 1) {
        const uint64_t s2 = s * P->pix_mult[i] + P->pix_plus[i];
        s = (p & 1u) ? s2 : s;
    }
    return s;
}

// Master PCG `delta` pixels after state s (per-lane delta, wave-uniform trip count so that the jump
// table is read with scalar loads).
__device__ __forceinline__ uint64_t master_jump(ColdParams P, uint64_t s, uint32_t delta) {
    for (uint32_t i = 0; i < 32u; ++i) {
        if (__ballot((delta >> i) != 0u) == 0ull) break;
        const uint64_t s2 = s * P->pix_mult[i] + P->pix_plus[i];
        s = ((delta >> i) & 1u) ? s2 : s;
    }
    return s;
}

// Pass 0 hands out pixels in a strided order so that every wave sees a mix of cheap (sky) and
// expensive image regions: queue entry h = (granule g, offset r) with 64-pixel granules; granule g
// maps to pixel granule s * stride + (g / 64 + s * (stride / 64)) % stride with s = g % 64.  Entries past
// the last pixel are skipped.
__device__ __forceinline__ uint32_t perm_entry(uint32_t h, uint32_t stride) {
    if (stride == 0u) return h;
    const uint32_t g = h >> 6, s = g & 63u;
    uint32_t o = (g >> 6) + s * (stride >> 6);   // each stratum starts a 64th of a stratum further on, so strata
    if (o >= stride) o -= stride;                // that begin at the same place in different frames do not sweep the image in step
    return ((s * stride + o) << 6) | (h & 63u);
}

// n / d for every u32 n, d >= 1, with mg = {m', sh1, sh2} of d: one mul_hi instead of a ~25-instruction expansion
__device__ __forceinline__ uint32_t div_magic(uint32_t n, uint32_t m, uint32_t sh1, uint32_t sh2) {
    const uint32_t t1 = __umulhi(m, n);
    return (t1 + ((n - t1) >> sh1)) >> sh2;
}

// local pixel -> global (row, col) through the row-tile partition
__device__ __forceinline__ void local_to_global(ColdParams P, uint32_t item, uint32_t& frame, uint32_t& lq,
                                                uint32_t& grow, uint32_t& gcol) {
    const uint32_t width = P->width, tile_rows = P->tile_rows, npf = P->n_per_frame;
    frame = (P->n_frames > 1u) ? div_magic(item, P->mg_npf[0], P->mg_npf[1], P->mg_npf[2]) : 0u;
    lq = item - frame * npf;
    const uint32_t lrow = div_magic(lq, P->mg_width[0], P->mg_width[1], P->mg_width[2]);
    gcol = lq - lrow * width;
    const uint32_t tile = (tile_rows > 1u) ? div_magic(lrow, P->mg_tile[0], P->mg_tile[1], P->mg_tile[2]) : lrow;
    uint32_t part = P->part;
    if (P->part_rotate != 0u) {
        const uint32_t x = part + frame;
        part = x - div_magic(x, P->mg_parts[0], P->mg_parts[1], P->mg_parts[2]) * P->n_parts;
    }
    grow = (tile * P->n_parts + part) * tile_rows + (lrow - tile * tile_rows);
}

// ImageTracer::fire_ray + Camera::fire_ray for the point (u_pixel, v_pixel) of pixel (grow, gcol)
// (imagetracer.rs:52-56; camera.rs:68-75, :132-143)
template <bool AFFINE>
__device__ __forceinline__ void camera_ray(ColdParams P, const FrameEntry* __restrict__ F, uint32_t grow, uint32_t gcol,
                                           float u_pixel, float v_pixel, Vec3& ro, Vec3& rd) {
    const float u = ((float)gcol + u_pixel) / (float)P->width;
    const float v = 1.0f - ((float)grow + v_pixel) / (float)P->height;
    const float dist = F->cam_distance, aspect = F->cam_aspect;
    float cox, coy, coz, cdx, cdy, cdz;
    if (F->cam_kind == RT_CAMERA_ORTHOGONAL) {
        cox = -1.0f; coy = (1.0f - 2.0f * u) * aspect; coz = 2.0f * v - 1.0f;
        cdx = 1.0f; cdy = 0.0f; cdz = 0.0f;
    } else {
        cox = -dist; coy = 0.0f; coz = 0.0f;
        cdx = dist; cdy = (1.0f - 2.0f * u) * aspect; cdz = 2.0f * v - 1.0f;
    }
    float M[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) M[i] = F->cam_m[i];
    ro = xform_point<AFFINE>(M, cox, coy, coz);
    rd = xform_vector(M, cdx, cdy, cdz);
}

// the ray of sample `samp` of pixel (grow, gcol) (imagetracer.rs:67-75); advances the master PCG by two draws.
template <bool AFFINE>
__device__ __forceinline__ void primary_ray(ColdParams P, const FrameEntry* __restrict__ F, uint32_t grow, uint32_t gcol,
                                            uint32_t samp, uint64_t& mstate, Vec3& ro, Vec3& rd) {
    const uint32_t aa = P->aa;
    const uint64_t minc = P->master_inc;
    const float rf0 = pcg_random_float(mstate, minc);
    const float rf1 = pcg_random_float(mstate, minc);
    float u_pixel, v_pixel;
    if (aa == 1u) {
        u_pixel = 0.0f + rf0;        // (0 + rf) / 1 == 0 + rf exactly
        v_pixel = 0.0f + rf1;
    } else {
        const uint32_t sub_row = samp / aa;
        const uint32_t sub_col = samp - sub_row * aa;
        u_pixel = ((float)sub_row + rf0) / (float)aa;
        v_pixel = ((float)sub_col + rf1) / (float)aa;
    }
    camera_ray<AFFINE>(P, F, grow, gcol, u_pixel, v_pixel, ro, rd);
}

// Lane (within its team) of last-level candidate (sibling j, s earlier survivors): rows of the
// triangle in order, inside a row the most probable survivor counts first (|2s - j| ascending).
__device__ __forceinline__ uint32_t tri_index(uint32_t j, uint32_t s) {
    const int dlt = 2 * (int)s - (int)j;
    const uint32_t rank = (dlt < 0) ? (uint32_t)(-dlt - 1) : (uint32_t)dlt;
    return ((j * (j + 1u)) >> 1) + rank;
}

__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, uint32_t src) {
    const uint32_t lo = __shfl((uint32_t)v, src), hi = __shfl((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}

// One node of the recursion: World::ray_intersection + the renderer's shading of the hit.
struct NodeOut {
    bool leaf;            // the ray returns `v` now (no open hit below it)
    bool traced;          // World::ray_intersection was called
    uint32_t consumed;    // PCG draws after the ray's own scatter draws
    float vr, vg, vb;
    // open hit (leaf == false): its children are traced in later rounds
    Vec3 ihc, iem, iwp, icv;
    uint32_t ibrdf;
};

// k_pass — one pass of the frame (see the file header).
//
// run_rounds<TEAMS>: TEAMS == false is one lane per pixel.  TEAMS == true: T lanes (wave-uniform power
// of two) hold identical copies of one pixel's recursion state and trace up to T SIBLING rays of
// the open hit per round, each under a hypothesis about the PCG draws its earlier siblings consume:
//   * below the last traced level a sibling is assumed to be a leaf consuming `pred` draws after
//     its two scatter draws (0 below the Russian-roulette depth, 1 at or beyond it): lane j traces
//     sibling j at offset j*(2+pred);
//   * at the last traced level (depth == max_depth: a hit cannot open, it consumes pred or
//     pred + 2n draws) the team enumerates (sibling j, s of the earlier siblings survived the
//     roulette) pairs, most probable first (TeamTables::fwd), at offset j*(2+pred) + 2n*s.
// After the round every lane walks the siblings in order along the hypotheses that turned out
// true, folds their radiances IN SIBLING ORDER, adopts the PCG state of the last sibling kept and
// re-speculates the rest.  Every draw and every f32 operation equals the sequential recursion of
// PathTracer::solve (src/render.rs:136-172); only the latency changes.
#ifndef RT_WAVES_PER_EU
#define RT_WAVES_PER_EU 4
#endif
#ifndef RT_SCAN_GROUP
#define RT_SCAN_GROUP 2
#endif
#ifndef RT_MIRROR_RUNS
#define RT_MIRROR_RUNS 1   // closed-form runs of a mirror hit's children whose value no draw can change (throughput body)
#endif
#ifndef RT_DRAIN_HELPERS
// Throughput body, queue dry: lanes without a pixel trace later siblings of another lane's last-level hit under hypotheses about
// the draws the earlier siblings consume (see "drain helpers" in the round loop).  0 compiles it out.
#define RT_DRAIN_HELPERS 1
#endif
#ifndef RT_HELP_MIN
#define RT_HELP_MIN 2      // free lanes per speculating lane from which a wave speculates (1 / 2 / 3: 2.26 / 2.24 / 2.37 ms for a 1080p frame)
#endif
#ifndef RT_SUBCHUNKS
#define RT_SUBCHUNKS 4     // FAST == 3: runs of consecutive pixels per 64-entry hand-out (1 / 4 / 8 / 16: 2.33 / 2.07 / 2.11 / 2.02 ms for a 1080p frame)
#endif
#ifndef RT_CHUNK
#define RT_CHUNK 64        // throughput body: pixels per queue hand-out (a divisor of 64; 16 / 8: 2.4x / 5x slower, every refill
                           // then waits for an atomic round trip)
#endif
#ifndef RT_MIRROR_DRAIN
#define RT_MIRROR_DRAIN 1  // throughput body, queue dry: all siblings of a mirror hit's leaf child in one round (2: hierarchy scenes too)
#endif
#ifndef RT_MIRROR_SIBS
// siblings of a mirror hit's leaf child resolved in the same round through the general roulette / skip code (throughput body, scene
// in LDS).  Measured before the closed-form run of siblings whose value no draw can change existed: 0 / 1 / 2 / 4 / 9 -> 0.709 /
// 0.641 / 0.652 / 0.648 / 0.742 ms per frame; with it: 0 / 1 / 2 / 4 -> 0.533 / 0.547 / 0.549 / 0.553.  So: off.
#define RT_MIRROR_SIBS 0
#endif
// FAST: the throughput variant (image render, path tracer, one pass without a round budget, one lane per
// pixel, no diagnostics): these launch parameters fold to constants, so their registers and branches go.
#define K_ALGO (FAST ? (uint32_t)RT_ALGO_PATHTRACER : P.algorithm)
#define K_MODE (FAST ? 0u : P.mode)
#define K_BUDGET (FAST ? 0u : P.budget)
#define K_PASS (FAST ? 0u : P.pass)
#ifdef RT_DEBUG_SECTIONS
#define K_DEBUG (P.debug_waves)
#else
#define K_DEBUG (FAST ? (unsigned long long*)nullptr : P.debug_waves)
#endif
template <bool AFFINE, bool STACK_LDS, bool SCENE_LDS, bool TEAMS, int FAST, bool BVH_LDS = false>
__global__ __launch_bounds__(RT_BLOCK, RT_WAVES_PER_EU) void k_pass(const PassParams P, const DScan* __restrict__ g_scan,
                                                   const DMRows* __restrict__ g_mrows,
                                                   const DWRows* __restrict__ g_wrows,
                                                   const DShade* __restrict__ g_shade,
                                                   const DMaterial* __restrict__ g_materials,
                                                   const float* __restrict__ texels) {
    extern __shared__ float4 lds_raw[];
    // FAST >= 2: the throughput body with drain helpers once the queue is dry; FAST == 3: mixed hand-outs too, for launches short
    // enough that their end matters more than the steady state (a frame or a few)
    constexpr bool HELP_K = (FAST == 2 || FAST == 3) && (RT_DRAIN_HELPERS != 0);
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t gtid = blockIdx.x * RT_BLOCK + tid;

    // ---- this pass's input and team size (wave-uniform, the same in every wave) ----
    ColdParams C0 = cold_params();
    uint32_t n_in = C0->n_in0;
    if (K_PASS != 0u) {
        n_in = P.ctl->count[K_PASS];
        if (n_in > C0->q_capacity) n_in = C0->q_capacity;
    }
    n_in = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_in);
    if (n_in == 0u) return;
    uint32_t T = 1u;
    if (C0->max_team > 1u) {
        uint32_t r = C0->lanes_target / n_in;
        if (r > C0->max_team) r = C0->max_team;
        while (T * 2u <= r) T *= 2u;
    }
    // the team variant of the kernel also runs T == 1 (later passes, whose queue length only the
    // device knows); the single variant is launched for passes the host pins to one lane per pixel
    if (!TEAMS) T = 1u;
    static_assert(!FAST || !TEAMS, "the FAST variant is one lane per pixel");
    if ((uint64_t)blockIdx.x * (RT_BLOCK / T) * C0->backlog >= (uint64_t)n_in && blockIdx.x != 0u) return;   // more blocks than work

    // ---- LDS: [team tables][DScan x n][DMRows x n][DShade x n][DMaterial x n (one per shape)][recursion stack] ----
    TeamTables* lds_tab = reinterpret_cast<TeamTables*>(lds_raw);
    const uint32_t n_staged = SCENE_LDS ? P.n_shapes : 0u;
    // FAST keeps, in the slot of the team tables, the master PCG's jump by 0..63 pixels: (mult, plus) pairs
    uint64_t* lds_jump = reinterpret_cast<uint64_t*>(lds_raw);
    uint64_t* lds_skip = lds_jump + 128;      // FAST: next to it, the pixel PCG's jump by 0..63 draws: (mult, geom) pairs
    uint64_t* lds_sv = lds_jump + 256;        // FAST: the pixel PCG's jump by s * 2 * num_of_rays draws, s = 0..15: (mult, geom) pairs
    uint8_t* lds_tri = reinterpret_cast<uint8_t*>(lds_jump + 288);   // FAST: candidate slot c -> sibling j (c = j (j + 1) / 2 + s, s <= j)
    // FAST: per wave, the lanes that speculate [0..63] and the free lanes [64..127], each by rank
    volatile uint8_t* lds_wv = lds_tri + 64u + (tid >> 6) * 128u;
    static_assert(sizeof(TeamTables) >= 2048 + 256 + 64 + (RT_BLOCK / 64) * 128, "the FAST tables live in the team tables' slot");
    DScan* lds_scan = reinterpret_cast<DScan*>(reinterpret_cast<unsigned char*>(lds_raw) + ((TEAMS || FAST) ? sizeof(TeamTables) : 0));
    DMRows* lds_mrows = reinterpret_cast<DMRows*>(lds_scan + n_staged);
    DShade* lds_shade = reinterpret_cast<DShade*>(lds_mrows + n_staged);
    DMaterial* lds_mats = reinterpret_cast<DMaterial*>(lds_shade + n_staged);
    float4* lds_stack = reinterpret_cast<float4*>(lds_mats + n_staged);
    // BVH_LDS: the hierarchy's nodes and leaf records instead (the recursion stack is in HBM then)
    static_assert(!BVH_LDS || (FAST == 1 && !SCENE_LDS && !STACK_LDS && !TEAMS), "the LDS hierarchy serves the plain throughput body");
    DBvhNode* lds_nodes = reinterpret_cast<DBvhNode*>(lds_scan);
    DScan* lds_leaf = reinterpret_cast<DScan*>(lds_nodes + (BVH_LDS ? P.bvh.n_nodes : 0u));
    if (BVH_LDS) {
        uint32_t* dst_n = reinterpret_cast<uint32_t*>(lds_nodes);
        const uint32_t* src_n = reinterpret_cast<const uint32_t*>(P.bvh.nodes);
        for (uint32_t i = tid; i < P.bvh.n_nodes * (uint32_t)(sizeof(DBvhNode) / 4); i += RT_BLOCK) dst_n[i] = src_n[i];
        uint32_t* dst_l = reinterpret_cast<uint32_t*>(lds_leaf);
        const uint32_t* src_l = reinterpret_cast<const uint32_t*>(P.bvh.leaf_scan);
        for (uint32_t i = tid; i < P.bvh.n_leaf * (uint32_t)(sizeof(DScan) / 4); i += RT_BLOCK) dst_l[i] = src_l[i];
    }
    if (TEAMS) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(C0->tables);
        uint32_t* dst = reinterpret_cast<uint32_t*>(lds_tab);
        for (uint32_t i = tid; i < (uint32_t)(sizeof(TeamTables) / 4); i += RT_BLOCK) dst[i] = src[i];
    }
    if (SCENE_LDS) {
        uint32_t* dst_s = reinterpret_cast<uint32_t*>(lds_scan);
        const uint32_t* src_s = reinterpret_cast<const uint32_t*>(g_scan);
        constexpr uint32_t SW = sizeof(DScan) / 4;
        for (uint32_t i = tid; i < n_staged * SW; i += RT_BLOCK) {
            const uint32_t sh = i / SW, j = i - sh * SW;
            dst_s[i] = (j == 13u) ? sh : src_s[i];        // staged materials are indexed by shape
        }
        uint32_t* dst_r = reinterpret_cast<uint32_t*>(lds_mrows);
        const uint32_t* src_r = reinterpret_cast<const uint32_t*>(g_mrows);
        for (uint32_t i = tid; i < n_staged * (uint32_t)(sizeof(DMRows) / 4); i += RT_BLOCK) dst_r[i] = src_r[i];
        uint32_t* dst_h = reinterpret_cast<uint32_t*>(lds_shade);
        const uint32_t* src_h = reinterpret_cast<const uint32_t*>(g_shade);
        for (uint32_t i = tid; i < n_staged * (uint32_t)(sizeof(DShade) / 4); i += RT_BLOCK) dst_h[i] = src_h[i];
        uint32_t* dst_m = reinterpret_cast<uint32_t*>(lds_mats);
        constexpr uint32_t MW = sizeof(DMaterial) / 4;
        for (uint32_t i = tid; i < n_staged * MW; i += RT_BLOCK) {
            const uint32_t sh = i / MW, j = i - sh * MW;
            dst_m[i] = reinterpret_cast<const uint32_t*>(g_materials + g_scan[sh].material)[j];
        }
    }
    if (FAST) {
        if (tid < 64u) {
            // compose the power-of-two jumps of the bits of tid: s -> s * m + p
            uint64_t m = 1, p = 0;
            for (uint32_t i = 0; i < 6u; ++i)
                if ((tid >> i) & 1u) {
                    p = p * C0->pix_mult[i] + C0->pix_plus[i];
                    m = m * C0->pix_mult[i];
                }
            lds_jump[2u * tid] = m;
            lds_jump[2u * tid + 1u] = p;
            // the pixel PCG advanced by tid draws: s -> s * M^tid + inc * (M^(tid-1) + ... + 1)
            uint64_t sm = 1, sg = 0;
            for (uint32_t i = 0; i < tid; ++i) {
                sg = sg * RT_PCG_MULT + 1ull;
                sm = sm * RT_PCG_MULT;
            }
            lds_skip[2u * tid] = sm;
            lds_skip[2u * tid + 1u] = sg;
            if (tid < 16u) {
                // tid skips of 2 * num_of_rays draws composed: s -> s * vm + inc * vg
                uint64_t vm = 1, vg = 0;
                for (uint32_t i = 0; i < tid; ++i) {
                    vg = vg * P.skip_mult + P.skip_plus;
                    vm = vm * P.skip_mult;
                }
                lds_sv[2u * tid] = vm;
                lds_sv[2u * tid + 1u] = vg;
            }
            uint32_t tj = 0;
            while ((tj + 1u) * (tj + 2u) / 2u <= tid) ++tj;
            lds_tri[tid] = (uint8_t)tj;
        }
    }
#if RT_MAILBOX
    constexpr bool MB_K = SCENE_LDS && STACK_LDS && (FAST == 1 || FAST == 2 || FAST == 3);
    volatile uint32_t* lds_mb = reinterpret_cast<volatile uint32_t*>(lds_stack + (size_t)P.stack_slots * (RT_NF / 4) * RT_BLOCK);
    if (MB_K && tid < 4u) lds_mb[tid] = 0u;        // postings [2], live waves [2]: one of each per round parity
#endif
    if (TEAMS || SCENE_LDS || FAST) __syncthreads();
    const DScan* __restrict__ scan = SCENE_LDS ? lds_scan : g_scan;
    const DMRows* __restrict__ mrows = SCENE_LDS ? lds_mrows : g_mrows;
    const DShade* __restrict__ shade = SCENE_LDS ? lds_shade : g_shade;
    const DMaterial* __restrict__ mats = SCENE_LDS ? lds_mats : g_materials;

    LevelStack<STACK_LDS> stack;
    if (STACK_LDS) { stack.base = lds_stack + tid; stack.stride = RT_BLOCK; }
    else { stack.base = reinterpret_cast<float4*>(C0->gstack) + gtid; stack.stride = C0->total_threads; }

    const uint32_t aa2 = C0->aa * C0->aa;
    const uint32_t n_children = P.num_of_rays;
    const uint32_t n_qfields = RT_QHEAD + RT_NF * (1u + P.stack_slots);
    (void)n_qfields;

    // ---- path state, replicated in every lane of the team ----
    uint32_t q = RT_INVALID;        // local pixel (mode 0) / batch index (mode 1) owned by the team
    uint32_t samp = 0;              // samples of q already started
    uint64_t mstate = 0;            // master PCG state at the next jitter draw of q
    uint64_t pstate = 0, pinc = 1;  // the pixel's PCG (state before this round's draws)
    uint64_t pskip = 0;             // FAST: pinc * skip_plus (the increment part of skipping 2*num_of_rays draws)
    uint32_t px_fr = 0, px_row = 0, px_col = 0;   // FAST: frame, image row and column of q (computed once per pixel)
    float acc_r = 0.f, acc_g = 0.f, acc_b = 0.f;
    bool busy = false;              // a top-level solve of q is in flight
    uint32_t lvl = 0;               // recursion level of the rays traced next (0 = the level-0 ray)
    uint32_t base_depth = 0;        // Ray.depth of the level-0 ray
    uint32_t rounds_left = 0;       // of this pass's budget
    // current level = the open hit at level lvl-1
    float hc_r = 0.f, hc_g = 0.f, hc_b = 0.f, em_r = 0.f, em_g = 0.f, em_b = 0.f;
    float cu_r = 0.f, cu_g = 0.f, cu_b = 0.f;
    uint32_t kcount = 0, cur_brdf = 0;   // kcount: children of the current level already started
    // FAST over a hierarchy: the children of a mirror hit all trace the same ray, so the current level remembers the answer
    // of the first one (0 unknown, 1 no hit, shape + 2) and the other children only re-run the winner's own test
    uint32_t c_hit = 0;
    // ... and a hit opened by such a repeated ray is its sibling's hit, so its own children repeat the sibling's children: per
    // lane, tmpl[l] (l = 1..4, one quad behind the saved levels) is what the level whose children are traced at level l had
    // learned when it last closed -- valid for the next hit at that level iff that hit is opened by a repeated ray and itself
    // started from a template (the body voids the entry below every hit that starts knowing nothing, whichever ray opened it:
    // a repeated ray whose siblings the roulette killed has no predecessor below).  A pure mirror chain then needs one
    // traversal per level, not one per hit.  FAST lanes own their pixel to its end (K_BUDGET 0), so the entries are the lane's.
    bool ray_cached = false;
    const uint32_t tmpl_base = P.stack_slots * RT_NF;
    const bool mirror_cache = FAST && !SCENE_LDS && P.bvh.n_nodes != 0u && P.num_of_rays < 256u && P.n_shapes < 0x7FFFF0u;
    if (mirror_cache) stack.st4(P.stack_slots, 0, make_float4(0.f, 0.f, 0.f, 0.f));   // nothing learned yet
    Vec3 cw = {0.f, 0.f, 0.f};           // hit.world_point
    Vec3 cv = {0.f, 0.f, 1.f};           // diffuse: hit.normal (raw); specular: the mirrored direction
    // counters (kept by the team leader)
    uint32_t n_rays = 0, n_samples = 0, n_diffuse = 0, n_specular = 0;
    uint32_t u_rays = 0, u_samples = 0, u_diffuse = 0, u_specular = 0, u_shared = 0;   // FAST: the same, wave-uniform (ballot counts)
    // FAST over a hierarchy: the lane's ray may stay in flight over several rounds (bvh_begin / bvh_advance, rt_device.h)
    constexpr bool INFL_C = FAST == 4;
    static_assert(!INFL_C || !SCENE_LDS, "rays in flight are for scenes traced through the hierarchy");
    const bool infl = INFL_C && P.bvh.n_nodes != 0u && P.trav_min_done != 0u;
    const uint32_t trav_min_done = P.trav_min_done;
    BvhTrav tr;
    tr.i = 0u; tr.i_end = 0u; tr.best = -1; tr.best_t = 0.f; tr.best_kind = 0u;
    tr.best_o = {0.f, 0.f, 0.f}; tr.best_d = {0.f, 0.f, 0.f};
    Vec3 tr_d = {0.f, 0.f, 0.f};       // direction of the ray in flight (its origin is cw)
    bool tr_fly = false, tr_ok = false; // a ray is in flight / it admits the sparse closest-hit form
#if RT_MAILBOX
    bool wave_done = false;             // mailbox experiment: this wave has nothing left but keeps the block's barriers company
    uint32_t mb_par = 0u;               // ... the round's parity
    bool open_req = false;              // ... this lane's hit opens: cw / cv hold the hit point and the object-space direction until the exchange
    uint32_t open_tag = 0u;             // ... shape | sphere << 30 | mirror << 31
    uint32_t dbg_mb_round = 0u;         // ... the round's number (the executor rotates)
#endif
    uint32_t l_sibs = 0, l_extra = 0;   // FAST, per lane: mirror siblings resolved beside their first child / siblings taken from drain helpers

    // wave-private slice of the input queue
    uint32_t wcur = 0, wend = 0;
    uint32_t wbase_q = 0, wbase_h = 0, wbase_p = 0, wbase_f = 0;   // pass 0: first pixel of the wave's current chunk,
    uint64_t wbase_state = 0;                        // its queue entry, global index and master PCG state
    // throughput body, scene in LDS: a hand-out is SUBS runs of consecutive pixels; lanes 0..SUBS-1 hold one run's base each
    constexpr uint32_t SUBS = (FAST == 3 && SCENE_LDS) ? (uint32_t)RT_SUBCHUNKS : 1u;   // short launches only: +7 % in the steady state
    static_assert(SUBS >= 1u && SUBS <= 64u && (SUBS & (SUBS - 1u)) == 0u, "RT_SUBCHUNKS is a power of two");
    uint32_t sub_q = 0, sub_p = 0, sub_f = 0;
    bool wmix = false;                               // the wave's current hand-out is a mixed one
    uint64_t sub_state = 0;
    bool exhausted = false;
    // diagnostics
    unsigned long long dbg_t0 = 0;
    uint32_t dbg_rounds = 0, dbg_lanes = 0;
#ifdef RT_HELP_DEBUG
    uint32_t hd_ex = 0, hd_few = 0, hd_eng = 0, hd_own = 0;
#endif
    if (K_DEBUG) dbg_t0 = __builtin_amdgcn_s_memrealtime();
    // section cycle counters (diagnostics): 0 generate 1 scatter 2 scan 3 hit+shade 4 resolve 5 park
    uint32_t dbg_sec0 = 0, dbg_sec1 = 0, dbg_sec2 = 0, dbg_sec3 = 0, dbg_sec4 = 0, dbg_sec5 = 0;
    // finer marks: 6 hit point + pigments 7 roulette 8 leaf value / draw skip 9 open hit 10 push 11 pop
    uint32_t dbg_sec6 = 0, dbg_sec7 = 0, dbg_sec8 = 0, dbg_sec9 = 0, dbg_sec10 = 0, dbg_sec11 = 0;
    uint64_t dbg_prev = 0;
#ifdef RT_DEBUG_SECTIONS   // per-section cycle counters cost ~10 VGPRs: diagnostics builds only (make EXTRA=-DRT_DEBUG_SECTIONS)
#define RT_DBG_MARK(i) do { if (K_DEBUG) { const uint64_t now_ = __builtin_readcyclecounter(); dbg_sec##i += (uint32_t)(now_ - dbg_prev); dbg_prev = now_; } } while (0)
#else
#define RT_DBG_MARK(i) do { (void)dbg_prev; } while (0)
#endif

    // World::ray_intersection: the linear scan of the shape table, or the hierarchy of a large scene
    auto closest_hit = [&](const Vec3 o, const Vec3 d, const float tmin, const float tmax, float& t, Vec3& lo, Vec3& ld,
                           uint32_t& bkind, const int cached = -2) -> int {
        if constexpr (!SCENE_LDS) {
            if (P.bvh.n_nodes != 0u) {
                // image renders trace with tmin 1e-5 / 1e-3 (>= 0, as the sparse form's proof needs); batch solves bring their own
                const bool dg = AFFINE && P.scene_diag != 0u && K_MODE == 0u && __ballot(!diag_ray_ok(o, d)) == 0ull;
                if constexpr (BVH_LDS) {
                    DBvh Bl = P.bvh;
                    Bl.nodes = lds_nodes;
                    Bl.leaf_scan = lds_leaf;
                    return bvh_closest_hit<AFFINE>(Bl, g_scan, g_wrows, P.n_shapes, o, d, tmin, tmax, t, lo, ld, bkind, dg, cached);
                }
                return bvh_closest_hit<AFFINE>(P.bvh, g_scan, g_wrows, P.n_shapes, o, d, tmin, tmax, t, lo, ld, bkind, dg, cached);
            }
        }
        if constexpr (FAST && AFFINE) {
            // scalings and translations only: the sparse form of the same arithmetic, when every live lane's ray qualifies
            // (tmin is 1e-5 or 1e-3 here)
            if (P.scene_diag != 0u && __ballot(!diag_ray_ok(o, d)) == 0ull)
                return scan_closest_hit_g<AFFINE, RT_SCAN_GROUP, true>(scan, g_wrows, P.n_shapes, o, d, tmin, tmax, t, lo, ld, bkind);
        }
        return scan_closest_hit_g<AFFINE, RT_SCAN_GROUP>(scan, g_wrows, P.n_shapes, o, d, tmin, tmax, t, lo, ld, bkind);
    };

#if RT_MAILBOX
    // world_point and normal of a hit that opens and the direction its children take (src/shape.rs:88-95, :134-138, :175-182,
    // :205-209; material.rs:225-246): the mailbox experiment's copy of the block in rt_fast_body.inc
    auto open_geometry = [&](const uint32_t shape_, const bool sphere_, const bool mirror_, const float hx_, const float hy_, const float hz_,
                             const Vec3 ld_, const Vec3 d_, Vec3& wp_, Vec3& ncv_) {
        const Rows3 MW = load_rows(mrows[shape_].m);
        const Rows3 IW = load_rows(scan[shape_].inv);
        wp_ = rows_point<AFFINE>(MW, AFFINE ? nullptr : g_wrows[shape_].m_w, hx_, hy_, hz_);
        float nx, ny, nz, dp;
        if (sphere_) {
            dp = hx_ * ld_.x + hy_ * ld_.y + hz_ * ld_.z;
            nx = hx_; ny = hy_; nz = hz_;
        } else {
            dp = 0.0f * ld_.x + 0.0f * ld_.y + 1.0f * ld_.z;
            nx = 0.0f; ny = 0.0f; nz = 1.0f;
        }
        if (!(dp < 0.0f)) { nx = -nx; ny = -ny; nz = -nz; }
        const Vec3 nrm = rows_normal(IW, nx, ny, nz);
        ncv_ = mirror_ ? specular_scatter_dir(d_, nrm) : nrm;
    };
#endif

    // ---------------------------------------------------------------------------------------------
    // one node: intersect + shade (render.rs:57-67, :76-89, :136-171)
    // ---------------------------------------------------------------------------------------------
    auto trace_node = [&](const Vec3 o, const Vec3 d, const float tmin, const float tmax, const uint32_t depth,
                          uint64_t& st, NodeOut& R) {
        R.leaf = true; R.traced = false; R.consumed = 0u;
        R.vr = 0.f; R.vg = 0.f; R.vb = 0.f;
        if (K_ALGO == RT_ALGO_DUMMY) {
            R.vr = 1.0f; R.vg = 2.0f; R.vb = 3.0f;   // render.rs:180-185
            return;
        }
        if (K_ALGO == RT_ALGO_PATHTRACER && depth > P.max_depth) return;   // render.rs:137-139
        float t;
        Vec3 lo, ld;                 // the winner's object-space ray
        uint32_t bkind;
        const int best = closest_hit(o, d, tmin, tmax, t, lo, ld, bkind);
        R.traced = true;
        RT_DBG_MARK(2);
        if (best < 0) {
            R.vr = P.bg[0]; R.vg = P.bg[1]; R.vb = P.bg[2];
            return;
        }
        if (K_ALGO == RT_ALGO_ONOFF) {
            R.vr = P.fg[0]; R.vg = P.fg[1]; R.vb = P.fg[2];   // render.rs:61-66
            return;
        }
        // HitRecord of the winner (src/shape.rs:134-142, :205-213): only the fields that are read
        const float4* shp = reinterpret_cast<const float4*>(shade + best);
        const float4 sh0 = shp[0], sh1 = shp[1];
        Vec3 hc = {sh0.x, sh0.y, sh0.z};
        Vec3 em = {sh1.x, sh1.y, sh1.z};
        const uint32_t sflags = __float_as_uint(sh0.w);
        const bool can_open = (K_ALGO == RT_ALGO_PATHTRACER) && (depth + 1u <= P.max_depth);
        const float hx = lo.x + ld.x * t, hy = lo.y + ld.y * t, hz = lo.z + ld.z * t;   // inv_ray.at(t)
        const bool sphere = (bkind == RT_SHAPE_SPHERE);
        if (sflags & 6u)   // some pigment of this material reads (u, v)
            hit_pigments(mats + (SCENE_LDS ? (uint32_t)best : scan[best].material), texels, sflags, sphere, hx, hy, hz, hc, em);
        RT_DBG_MARK(6);
        if (K_ALGO == RT_ALGO_FLAT) {
            R.vr = em.x + hc.x; R.vg = em.y + hc.y; R.vb = em.z + hc.z;   // render.rs:83-84
            return;
        }
        // PathTracer::solve (render.rs:146-171)
        const float lum = rt_maxf(hc.x, rt_maxf(hc.y, hc.z));
        if (depth >= P.rr_limit) {
            const float rq = rt_maxf(1.0f - lum, 0.05f);
            R.consumed = 1u;
            if (pcg_random_float(st, pinc) > rq) {
                const float s = 1.0f / (1.0f - rq);
                hc.x = hc.x * s; hc.y = hc.y * s; hc.z = hc.z * s;
            } else {
                R.vr = em.x; R.vg = em.y; R.vb = em.z;
                return;
            }
        }
        RT_DBG_MARK(7);
        if (lum > 0.0f && n_children != 0u) {
            if (!can_open) {
                // every child returns black without an intersection (render.rs:137-139) but its
                // two scatter draws are consumed (render.rs:161).
                st = st * P.skip_mult + pinc * P.skip_plus;
                R.consumed += 2u * n_children;
                const float cr = 0.0f + hc.x * 0.0f, cg = 0.0f + hc.y * 0.0f, cb = 0.0f + hc.z * 0.0f;
                R.vr = em.x + cr * P.inv_num_of_rays;
                R.vg = em.y + cg * P.inv_num_of_rays;
                R.vb = em.z + cb * P.inv_num_of_rays;
                RT_DBG_MARK(8);
            } else {
                R.leaf = false;
                R.ihc = hc; R.iem = em;
                R.ibrdf = sflags & 1u;
                // world_point and normal of the hit (src/shape.rs:88-95, :134-138, :175-182, :205-209)
                const Rows3 MW = load_rows(mrows[best].m);
                const Rows3 IW = load_rows(scan[best].inv);
                R.iwp = rows_point<AFFINE>(MW, AFFINE ? nullptr : g_wrows[best].m_w, hx, hy, hz);
                float nx, ny, nz, dp;
                if (sphere) {
                    dp = hx * ld.x + hy * ld.y + hz * ld.z;
                    nx = hx; ny = hy; nz = hz;
                } else {
                    dp = 0.0f * ld.x + 0.0f * ld.y + 1.0f * ld.z;
                    nx = 0.0f; ny = 0.0f; nz = 1.0f;
                }
                if (!(dp < 0.0f)) { nx = -nx; ny = -ny; nz = -nz; }
                const Vec3 nrm = rows_normal(IW, nx, ny, nz);
                // SpecularBRDF::scatter_ray ignores its draws: every child of this hit has the same
                // direction, computed once from hit.ray.dir and hit.normal (material.rs:225-246)
                if (R.ibrdf == RT_BRDF_SPECULAR) R.icv = specular_scatter_dir(d, nrm);
                else R.icv = nrm;
                RT_DBG_MARK(9);
            }
        } else {
            // no children: emitted + 0 * (1/n) (render.rs:157, :171)
            R.vr = em.x + 0.0f * P.inv_num_of_rays;
            R.vg = em.y + 0.0f * P.inv_num_of_rays;
            R.vb = em.z + 0.0f * P.inv_num_of_rays;
        }
    };

    // ---------------------------------------------------------------------------------------------
    // the round loop
    // ---------------------------------------------------------------------------------------------
    {
    const uint32_t TT = TEAMS ? T : 1u;
    const uint32_t jj = TEAMS ? (lane & (TT - 1u)) : 0u;       // candidate slot of this lane
    const uint32_t team0 = TEAMS ? (lane & ~(TT - 1u)) : lane; // first lane of the team
    const bool leader = (jj == 0u);
    const unsigned long long tmask = (TT >= 64u) ? ~0ull : ((1ull << TT) - 1ull);
    const uint32_t n_teams = 64u / TT;
    uint32_t idle_min = n_teams / cold_params()->idle_div;
    if (idle_min < 1u) idle_min = 1u;

    if (K_DEBUG) dbg_prev = __builtin_readcyclecounter();
    for (;;) {
        // =====================================================================================
        // generate / compact in: teams without a solve in flight retire their pixel, take the
        // next sample, or take a new entry of the pass's queue
        // =====================================================================================
        const unsigned long long idle_leaders = __ballot(!busy && leader);
        const unsigned long long busy_any = __ballot(busy);
        if (idle_leaders != 0ull && ((uint32_t)__popcll(idle_leaders) >= idle_min || busy_any == 0ull)) {
            bool want = false;
            ColdParams C = cold_params();
            if (!busy) {
                if (q != RT_INVALID && samp == aa2) {
                    // pixel (or solve item) finished: color * (1 / aa^2)  (imagetracer.rs:121)
                    if (leader) {
                        if (K_MODE == 0u) {
                            uint32_t fr, lq, gr_, gc_;
                            if (FAST) { fr = px_fr; lq = q - fr * C->n_per_frame; gr_ = px_row; gc_ = px_col; }
                            else local_to_global(C, q, fr, lq, gr_, gc_);
                            if (C->out_in_place != 0u) lq = gr_ * C->width + gc_;
                            float* o = frame_entry(C, fr)->out + 3ull * lq;
                            const float inv_samples = C->inv_samples;
                            o[0] = acc_r * inv_samples;
                            o[1] = acc_g * inv_samples;
                            o[2] = acc_b * inv_samples;
                        } else {
                            float* o = C->out + 3ull * q;
                            o[0] = acc_r; o[1] = acc_g; o[2] = acc_b;
                            C->io_pcgs[q].state = pstate;
                        }
                    }
                    q = RT_INVALID;
                }
                want = (q == RT_INVALID) && !exhausted;
            }
            // wave64 ballot + prefix rank hand out consecutive queue entries to the idle teams
            bool lead_want = want && leader;
            unsigned long long wantmask = __ballot(lead_want);
            uint32_t h = RT_INVALID;
            uint32_t my_base_p = 0, my_base_f = 0;   // the chunk base this lane's entry was taken from
            uint64_t my_base_state = 0;
            while (wantmask != 0ull) {
                if (wcur == wend) {
                    uint32_t b = 0;
                    // FAST hands out RT_CHUNK pixels at a time: a wave's lanes then hold pixels of several image regions, so the
                    // long pixels of one region do not all end up in the same wave
                    const uint32_t chunk = FAST ? (uint32_t)RT_CHUNK : n_teams;
                    if (lane == 0) b = atomicAdd(&P.ctl->head[K_PASS], chunk);
                    b = __shfl(b, 0);      // (readfirstlane instead makes the hand-out state scalar and the kernel 4 % slower: measured)
                    if (b >= n_in) { exhausted = true; break; }
                    wcur = b;
                    wend = (b + chunk < n_in) ? b + chunk : n_in;
                    wmix = SUBS > 1u && C->perm_stride != 0u && (b >> 6) >= C->mix_from;
                    if (SUBS > 1u && wmix) {
                        // The 64 entries of a hand-out are SUBS runs of 64 / SUBS consecutive pixels, run i taken from the
                        // granule 64 / SUBS places further on in the strided order (another stratum of the image: the long
                        // pixels of one image region do not all end up in one wave).  Entry (granule g, offset r) -> run i =
                        // r / (64 / SUBS) of granule (g + i * 64 / SUBS) mod G: a bijection of the entries, as g -> g + const is.
                        // Only the end of the queue is mixed, granules [mix_from, G) among themselves (the pixels in flight when
                        // the queue runs dry are the ones handed out last; mixing costs the steady state 7 %).
                        // Lanes 0..SUBS-1 prepare one run each: its first pixel and the master PCG there.
                        wbase_h = b;
                        const uint32_t G0 = C->mix_from, Gt = (C->perm_stride << 6) - G0;
                        const uint32_t si = lane & (SUBS - 1u);
                        const uint32_t g2 = G0 + (((b >> 6) - G0) + si * (64u / SUBS)) % Gt;
                        sub_q = perm_entry(g2 << 6, C->perm_stride) + si * (64u / SUBS);
                        uint32_t f0 = 0, lq0, r0 = 0, c0 = 0;
                        const bool okq = sub_q < P.n_items;
                        local_to_global(C, okq ? sub_q : 0u, f0, lq0, r0, c0);
                        sub_f = f0;
                        sub_p = r0 * C->width + c0;
                        sub_state = master_jump(C, C->master_state0, (okq && lane < SUBS) ? sub_p : 0u);
                    } else if (K_PASS == 0u && K_MODE == 0u) {
                        // master PCG at the chunk's first pixel, on the scalar unit (imagetracer.rs:63-81)
                        wbase_h = b;
                        wbase_q = (uint32_t)__builtin_amdgcn_readfirstlane((int)perm_entry(b, C->perm_stride));
                        if (wbase_q < P.n_items) {
                            uint32_t f0, lq0, r0, c0;
                            local_to_global(C, wbase_q, f0, lq0, r0, c0);
                            wbase_f = (uint32_t)__builtin_amdgcn_readfirstlane((int)f0);
                            wbase_p = (uint32_t)__builtin_amdgcn_readfirstlane((int)(r0 * C->width + c0));
                            wbase_state = master_state_at(C, wbase_p);
                        }
                    }
                }
                const uint32_t avail = wend - wcur;
                const uint32_t wanted = (uint32_t)__popcll(wantmask);
                const uint32_t cnt = wanted < avail ? wanted : avail;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(wantmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wantmask, 0u));
                uint32_t run_q = 0, run_p = 0, run_f = 0;
                uint64_t run_state = 0;
                const bool runs = SUBS > 1u && wmix;
                if (runs) {
                    // every lane reads the run its entry would belong to (the shuffles need the whole wave)
                    const uint32_t r = (wcur + rank - wbase_h) & 63u;
                    const uint32_t si = r / (64u / SUBS);
                    run_q = __shfl(sub_q, si) + (r & (64u / SUBS - 1u));
                    run_p = __shfl(sub_p, si);
                    run_f = __shfl(sub_f, si);
                    run_state = shfl_u64(sub_state, si);
                }
                if (lead_want && rank < cnt) {
                    h = wcur + rank;
                    // pass 0: the chunk's entries are consecutive pixels starting at wbase_q
                    if (runs) {
                        h = run_q;
                        my_base_p = run_p;
                        my_base_f = run_f;
                        my_base_state = run_state;
                        if (h >= P.n_items) h = RT_INVALID - 1u;   // ragged end of the strided order: nothing to do
                    } else if (K_PASS == 0u && K_MODE == 0u) {
                        h = wbase_q + (h - wbase_h);
                        my_base_p = wbase_p;
                        my_base_f = wbase_f;
                        my_base_state = wbase_state;
                        if (h >= P.n_items) h = RT_INVALID - 1u;   // ragged end of the strided order: nothing to do
                    }
                    lead_want = false;
                }
                wcur += cnt;
                wantmask = __ballot(lead_want);
            }
            if (TEAMS) {
                h = __shfl(h, team0);
                my_base_p = __shfl(my_base_p, team0);
                my_base_f = __shfl(my_base_f, team0);
                my_base_state = shfl_u64(my_base_state, team0);
            }
            if (h == RT_INVALID - 1u) h = RT_INVALID;
            if (!busy && h != RT_INVALID) {
                rounds_left = K_BUDGET;
                if (K_PASS == 0u) {
                    q = h;
                    samp = 0;
                    acc_r = 0.f; acc_g = 0.f; acc_b = 0.f;
                    lvl = 0;
                    if (K_MODE == 0u) {
                        uint32_t fr, lq, grow, gcol;
                        local_to_global(C, q, fr, lq, grow, gcol);
                        if (FAST) { px_fr = fr; px_row = grow; px_col = gcol; }
                        // master PCG at this pixel: jump ahead p*(2*aa^2+1) draws (imagetracer.rs:63-81),
                        // from the chunk's first pixel (a chunk that runs into the next frame restarts at 0)
                        const uint32_t gp = grow * C->width + gcol;
                        const bool same = (fr == my_base_f);
                        const uint64_t jbase = same ? my_base_state : C->master_state0;
                        const uint32_t jdelta = same ? gp - my_base_p : gp;
                        uint64_t s;
                        if (FAST && __ballot(jdelta >= 64u) == 0ull) {
                            s = jbase * lds_jump[2u * jdelta] + lds_jump[2u * jdelta + 1u];   // the usual case: one step
                        } else {
                            s = master_jump(C, jbase, jdelta);
                        }
                        mstate = s;
                        // seq = the u32 drawn after the pixel's 2*aa^2 jitter draws (imagetracer.rs:79)
                        const uint64_t sq = s * C->seq_mult + C->seq_plus;
                        const uint64_t seq = (uint64_t)pcg_output(sq);
                        pcg_new(C->init_state, seq, pstate, pinc);   // imagetracer.rs:117
                        if (FAST) pskip = pinc * P.skip_plus;
                    }
                } else {
                    // a path parked by the previous pass
                    const uint32_t qcap = C->q_capacity;
                    const float* e = C->qbuf[(K_PASS + 1u) & 1u] + h;
                    q = __float_as_uint(e[0ull * qcap]);
                    samp = __float_as_uint(e[1ull * qcap]);
                    lvl = __float_as_uint(e[2ull * qcap]);
                    base_depth = __float_as_uint(e[3ull * qcap]);
                    mstate = (uint64_t)__float_as_uint(e[4ull * qcap]) | ((uint64_t)__float_as_uint(e[5ull * qcap]) << 32);
                    pstate = (uint64_t)__float_as_uint(e[6ull * qcap]) | ((uint64_t)__float_as_uint(e[7ull * qcap]) << 32);
                    pinc = (uint64_t)__float_as_uint(e[8ull * qcap]) | ((uint64_t)__float_as_uint(e[9ull * qcap]) << 32);
                    acc_r = e[10ull * qcap]; acc_g = e[11ull * qcap]; acc_b = e[12ull * qcap];
                    if (lvl > 0u) {
                        const float* c = e + (uint64_t)RT_QHEAD * qcap;
                        hc_r = c[0ull * qcap]; hc_g = c[1ull * qcap]; hc_b = c[2ull * qcap];
                        em_r = c[3ull * qcap]; em_g = c[4ull * qcap]; em_b = c[5ull * qcap];
                        cu_r = c[6ull * qcap]; cu_g = c[7ull * qcap]; cu_b = c[8ull * qcap];
                        cw.x = c[9ull * qcap]; cw.y = c[10ull * qcap]; cw.z = c[11ull * qcap];
                        cv.x = c[12ull * qcap]; cv.y = c[13ull * qcap]; cv.z = c[14ull * qcap];
                        const uint32_t kb = __float_as_uint(c[15ull * qcap]);
                        kcount = kb >> 1;
                        cur_brdf = kb & 1u;
                        const uint32_t n_saved = (lvl - 1u) * RT_NF;
                        const float* sv = c + (uint64_t)RT_NF * qcap;
                        for (uint32_t i = 0; i < n_saved; ++i) stack.dword(i) = sv[(uint64_t)i * qcap];
                        busy = true;
                    }
                }
            }
            if (FAST) u_samples += (uint32_t)__popcll(__ballot(!busy && q != RT_INVALID));
            if (!busy && q != RT_INVALID) {
                // start the next top-level solve of q (imagetracer.rs:118-120); its level-0 ray is
                // produced in the candidate stage below
                if (K_MODE == 0u) {
                    base_depth = 0;
                } else {
                    base_depth = C->in_rays[q].depth;
                    pstate = C->io_pcgs[q].state;
                    pinc = C->io_pcgs[q].inc;
                }
                samp += 1;
                if (!FAST && leader) n_samples += 1;
                lvl = 0;
                busy = true;
            }
        }
#if RT_MAILBOX
        // every wave of the block meets the round's two barriers until all of them are done (the body's exchange ends the loop)
        if (__ballot(busy) == 0ull && exhausted) wave_done = true;
        if (!MB_K)
#endif
        if (__ballot(busy) == 0ull) {
            if (exhausted) break;           // queue empty and every team retired or parked
            continue;                       // only skipped entries so far: take more
        }
        RT_DBG_MARK(0);

        if constexpr (FAST) {
            // =================================================================================
            // FAST: one lane per pixel; intersect, shade and resolve fused.  The same draws and f32
            // operations as the general path below (trace_node + resolve), but a hit that opens
            // writes the recursion state directly (no NodeOut record to merge) and a leaf is folded
            // where its value is known.
            // =================================================================================
            // The body exists twice in the k_pass<..., 2> kernel: the helpers' code in the steady-state rounds costs 4 % (registers,
            // scheduling), so a wave runs the plain copy until its queue is dry and the copy with drain helpers afterwards.
            if (HELP_K && exhausted) {
#define RT_BODY_HELP true
#include "rt_fast_body.inc"
#undef RT_BODY_HELP
            } else {
#define RT_BODY_HELP false
#include "rt_fast_body.inc"
#undef RT_BODY_HELP
            }
            continue;
        }

        // =====================================================================================
        // this round's candidate rays: the level-0 ray, or siblings of the open hit
        // =====================================================================================
        const uint32_t depth = FAST ? lvl : base_depth + lvl;     // Ray.depth of the rays traced this round
        // draws an earlier sibling is assumed to consume after its two scatter draws
        const uint32_t pred = (lvl > 0u && depth >= P.rr_limit) ? 1u : 0u;
        const bool last_level = TEAMS && (lvl > 0u) && (depth == P.max_depth);
        uint32_t cj = jj, cs = 0u;   // this lane's candidate: sibling kcount+cj, cs earlier survivors
        bool cand_ok = true;
        if (TEAMS && last_level) {
            const uint32_t fw = lds_tab->fwd[jj];
            cj = fw & 0xFFu;
            cs = (fw >> 8) & 31u;
            cand_ok = (fw != 0xFFFFu);
        }
        uint32_t n_rem = 0;          // rays this level still has to start
        if (busy) n_rem = (lvl == 0u) ? 1u : (n_children - kcount);
        const bool act = busy && ((lvl == 0u) ? (jj == 0u) : (cand_ok && cj < n_rem));
        if (K_DEBUG) { dbg_rounds += 1u; dbg_lanes += (uint32_t)__popcll(__ballot(act)); }

        uint64_t st = pstate;        // this lane's PCG, advanced through its own ray's draws
        NodeOut R;
        R.leaf = true; R.traced = false; R.consumed = 0u;
        R.vr = 0.f; R.vg = 0.f; R.vb = 0.f;
        R.ihc = {0.f, 0.f, 0.f}; R.iem = {0.f, 0.f, 0.f}; R.iwp = {0.f, 0.f, 0.f}; R.icv = {0.f, 0.f, 1.f};
        R.ibrdf = 0u;
        Vec3 o = {0.f, 0.f, 0.f}, d = {1.f, 0.f, 0.f};
        float tmin = 1e-5f, tmax = __builtin_inff();
        if (busy && lvl == 0u) {
            // the level-0 ray of sample samp-1 (every lane of the team: the master PCG stays replicated)
            ColdParams C = cold_params();
            if (K_MODE == 0u) {
                uint32_t fr, lq, grow, gcol;
                local_to_global(C, q, fr, lq, grow, gcol);
                primary_ray<AFFINE>(C, frame_entry(C, fr), grow, gcol, samp - 1u, mstate, o, d);
            } else {
                const rt_ray r = C->in_rays[q];
                o = {r.origin[0], r.origin[1], r.origin[2]};
                d = {r.dir[0], r.dir[1], r.dir[2]};
                tmin = r.tmin;
                tmax = r.tmax;
            }
        }
        if (act) {
            if (lvl != 0u) {
                // BRDF::scatter_ray of this lane's sibling (render.rs:160-166)
                if (TEAMS) {
                    if (pred) st = st * lds_tab->j3_mult[cj] + pinc * lds_tab->j3_geom[cj];
                    else st = st * lds_tab->j2_mult[cj] + pinc * lds_tab->j2_geom[cj];
                    if (last_level) st = st * lds_tab->sv_mult[cs] + pinc * lds_tab->sv_geom[cs];
                }
                const float rf0 = pcg_random_float(st, pinc);
                const float rf1 = pcg_random_float(st, pinc);
                if (cur_brdf == RT_BRDF_DIFFUSE) {
                    d = diffuse_scatter_dir(rf0, rf1, cv);
                    tmin = 1.0e-3f;                   // material.rs:181
                } else {
                    d = cv;
                    tmin = 1e-5f;                     // Ray::default (ray.rs:50)
                }
                o = cw;
                tmax = __builtin_inff();
            }
            RT_DBG_MARK(1);
            trace_node(o, d, tmin, tmax, depth, st, R);
        }
        RT_DBG_MARK(3);

        // =====================================================================================
        // resolve: walk the siblings in order along the hypotheses that held, fold their radiances
        // =====================================================================================
        float pr, pg, pb;
        if (lvl == 0u) { pr = R.vr; pg = R.vg; pb = R.vb; }
        else { pr = hc_r * R.vr; pg = hc_g * R.vg; pb = hc_b * R.vb; }   // render.rs:168
        float tr = (lvl == 0u) ? acc_r : cu_r, tg = (lvl == 0u) ? acc_g : cu_g, tb_ = (lvl == 0u) ? acc_b : cu_b;
        uint32_t n_valid = 1u, last = 0u, traced_valid = R.traced ? 1u : 0u;
        bool last_leaf = R.leaf;
        uint64_t nstate = st;
        if (TEAMS) {
            const unsigned long long b_act = (__ballot(act) >> team0) & tmask;
            const unsigned long long b_leaf = (__ballot(act && R.leaf) >> team0) & tmask;
            const unsigned long long b_traced = (__ballot(R.traced) >> team0) & tmask;
            const unsigned long long b_lo = (__ballot(act && R.leaf && (lvl == 0u || R.consumed == pred)) >> team0) & tmask;
            const unsigned long long b_hi = (__ballot(act && R.leaf && last_level && R.consumed == pred + 2u * n_children) >> team0) & tmask;
            // phase 1: which candidates lie on the true path (bit operations only).  hi_hist bit j:
            // sibling j of this round survived the roulette; the lane of candidate (j, s) is
            // tri_index(j, s) at the last level and j elsewhere.
            uint32_t surv = 0u, j = 0u, hi_hist = 0u;
            bool cont = busy;            // sibling j of this round is on the true path
            n_valid = 0u; traced_valid = 0u;
            while (__ballot(cont) != 0ull) {
                const uint32_t idx = last_level ? tri_index(j, surv) : j;
                const bool ok = cont && idx < TT && ((b_act >> idx) & 1ull) != 0ull;
                if (ok) {
                    n_valid = j + 1u;
                    last = idx;
                    traced_valid += (uint32_t)((b_traced >> idx) & 1ull);
                }
                const bool hi = ok && ((b_hi >> idx) & 1ull) != 0ull;
                cont = ok && (hi || ((b_lo >> idx) & 1ull) != 0ull) && j < 31u;
                if (hi) { surv += 1u; hi_hist |= 1u << j; }
                j += 1u;
            }
            // phase 2: fold the kept leaves into the level, in sibling order (render.rs:168 / imagetracer.rs:119)
            for (j = 0u; __ballot(busy && j < n_valid) != 0ull; ++j) {
                const uint32_t sj = (uint32_t)__popc(hi_hist & ((1u << j) - 1u));
                const uint32_t idx = last_level ? tri_index(j, sj) : j;
                const bool ok = busy && j < n_valid;
                const uint32_t src = team0 + (ok ? idx : 0u);
                const float qr = __shfl(pr, src), qg = __shfl(pg, src), qb = __shfl(pb, src);
                if (ok && ((b_leaf >> idx) & 1ull) != 0ull) { tr = tr + qr; tg = tg + qg; tb_ = tb_ + qb; }
            }
            last_leaf = ((b_leaf >> last) & 1ull) != 0ull;
            nstate = shfl_u64(st, team0 + last);
            // open-hit data of the last sibling kept
            if (__ballot(busy && !last_leaf) != 0ull) {
                const uint32_t src = team0 + last;
                R.ihc.x = __shfl(R.ihc.x, src); R.ihc.y = __shfl(R.ihc.y, src); R.ihc.z = __shfl(R.ihc.z, src);
                R.iem.x = __shfl(R.iem.x, src); R.iem.y = __shfl(R.iem.y, src); R.iem.z = __shfl(R.iem.z, src);
                R.iwp.x = __shfl(R.iwp.x, src); R.iwp.y = __shfl(R.iwp.y, src); R.iwp.z = __shfl(R.iwp.z, src);
                R.icv.x = __shfl(R.icv.x, src); R.icv.y = __shfl(R.icv.y, src); R.icv.z = __shfl(R.icv.z, src);
                R.ibrdf = __shfl(R.ibrdf, src);
            }
        } else {
            if (R.leaf) { tr = tr + pr; tg = tg + pg; tb_ = tb_ + pb; }
        }
        if (busy) {
            pstate = nstate;
            if (lvl == 0u) { acc_r = tr; acc_g = tg; acc_b = tb_; }
            else { cu_r = tr; cu_g = tg; cu_b = tb_; }
            if (!FAST && leader) {
                n_rays += traced_valid;
                if (lvl > 0u) {
                    if (cur_brdf == RT_BRDF_DIFFUSE) n_diffuse += n_valid; else n_specular += n_valid;
                }
            }
            if (lvl > 0u) kcount += n_valid;
            if (rounds_left > 0u) rounds_left -= 1u;
            RT_DBG_MARK(4);

            if (!last_leaf) {
                // the last kept sibling is an open hit: it becomes the current level
                if (lvl > 0u) {
                    const uint32_t s = lvl - 1u;
                    stack.st4(s, 0, make_float4(hc_r, hc_g, hc_b, em_r));
                    stack.st4(s, 1, make_float4(em_g, em_b, cu_r, cu_g));
                    stack.st4(s, 2, make_float4(cu_b, cw.x, cw.y, cw.z));
                    stack.st4(s, 3, make_float4(cv.x, cv.y, cv.z, __uint_as_float((kcount << 1) | cur_brdf)));
                }
                hc_r = R.ihc.x; hc_g = R.ihc.y; hc_b = R.ihc.z;
                em_r = R.iem.x; em_g = R.iem.y; em_b = R.iem.z;
                cw = R.iwp; cv = R.icv;
                cur_brdf = R.ibrdf;
                cu_r = 0.f; cu_g = 0.f; cu_b = 0.f;
                kcount = 0;
                lvl += 1u;
                RT_DBG_MARK(10);
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
                    const float4 s0 = stack.ld4(s, 0), s1 = stack.ld4(s, 1), s2 = stack.ld4(s, 2), s3 = stack.ld4(s, 3);
                    hc_r = s0.x; hc_g = s0.y; hc_b = s0.z;
                    em_r = s0.w; em_g = s1.x; em_b = s1.y;
                    cu_r = s1.z; cu_g = s1.w; cu_b = s2.x;
                    cw.x = s2.y; cw.y = s2.z; cw.z = s2.w;
                    cv.x = s3.x; cv.y = s3.y; cv.z = s3.z;
                    const uint32_t kb = __float_as_uint(s3.w);
                    kcount = kb >> 1;
                    cur_brdf = kb & 1u;
                    cu_r = cu_r + hc_r * wr; cu_g = cu_g + hc_g * wg; cu_b = cu_b + hc_b * wb;   // render.rs:168
                }
                RT_DBG_MARK(11);
            }
        }

        RT_DBG_MARK(4);
        // =====================================================================================
        // compact out: paths that used up this pass's budget are parked for the next pass
        // =====================================================================================
        if (K_BUDGET != 0u) {
            const bool live = busy || (q != RT_INVALID && samp < aa2);
            const bool park = live && rounds_left == 0u;
            const unsigned long long pm = __ballot(park && leader);
            if (pm != 0ull) {
                ColdParams C = cold_params();
                const uint32_t qcap = C->q_capacity;
                float* q_out = C->qbuf[K_PASS & 1u];
                uint32_t base = 0;
                const uint32_t first = (uint32_t)__builtin_ctzll(pm);
                if (lane == first) base = atomicAdd(&P.ctl->count[K_PASS + 1u], (uint32_t)__popcll(pm));
                base = __shfl(base, first);
                uint32_t slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                if (TEAMS) slot = __shfl(slot, team0);
                if (park) {
                    if (slot < qcap) {
                        if (leader) {
                            float* e = q_out + slot;
                            const uint32_t plvl = busy ? lvl : 0u;
                            e[0ull * qcap] = __uint_as_float(q);
                            e[1ull * qcap] = __uint_as_float(samp);
                            e[2ull * qcap] = __uint_as_float(plvl);
                            e[3ull * qcap] = __uint_as_float(base_depth);
                            e[4ull * qcap] = __uint_as_float((uint32_t)mstate); e[5ull * qcap] = __uint_as_float((uint32_t)(mstate >> 32));
                            e[6ull * qcap] = __uint_as_float((uint32_t)pstate); e[7ull * qcap] = __uint_as_float((uint32_t)(pstate >> 32));
                            e[8ull * qcap] = __uint_as_float((uint32_t)pinc); e[9ull * qcap] = __uint_as_float((uint32_t)(pinc >> 32));
                            e[10ull * qcap] = acc_r; e[11ull * qcap] = acc_g; e[12ull * qcap] = acc_b;
                            if (plvl > 0u) {
                                float* c = e + (uint64_t)RT_QHEAD * qcap;
                                c[0ull * qcap] = hc_r; c[1ull * qcap] = hc_g; c[2ull * qcap] = hc_b;
                                c[3ull * qcap] = em_r; c[4ull * qcap] = em_g; c[5ull * qcap] = em_b;
                                c[6ull * qcap] = cu_r; c[7ull * qcap] = cu_g; c[8ull * qcap] = cu_b;
                                c[9ull * qcap] = cw.x; c[10ull * qcap] = cw.y; c[11ull * qcap] = cw.z;
                                c[12ull * qcap] = cv.x; c[13ull * qcap] = cv.y; c[14ull * qcap] = cv.z;
                                c[15ull * qcap] = __uint_as_float((kcount << 1) | cur_brdf);
                                const uint32_t n_saved = (plvl - 1u) * RT_NF;
                                float* sv = c + (uint64_t)RT_NF * qcap;
                                for (uint32_t i = 0; i < n_saved; ++i) sv[(uint64_t)i * qcap] = stack.dword(i);
                            }
                        }
                        q = RT_INVALID;
                        busy = false;
                    } else {
                        rounds_left = K_BUDGET;   // the queue is full: keep running this path here
                    }
                }
            }
        }
        RT_DBG_MARK(5);
    }
    }   // round loop

    if (K_DEBUG && lane == 0) {
        unsigned long long* d = K_DEBUG + 17ull * ((unsigned long long)K_PASS * P.debug_stride + (gtid >> 6));
        d[0] = dbg_t0; d[1] = __builtin_amdgcn_s_memrealtime(); d[2] = dbg_rounds; d[3] = dbg_lanes; d[4] = T;
        d[5] = dbg_sec0; d[6] = dbg_sec1; d[7] = dbg_sec2; d[8] = dbg_sec3; d[9] = dbg_sec4; d[10] = dbg_sec5;
        d[11] = dbg_sec6; d[12] = dbg_sec7; d[13] = dbg_sec8; d[14] = dbg_sec9; d[15] = dbg_sec10; d[16] = dbg_sec11;
#ifdef RT_HELP_DEBUG
        d[13] = hd_ex; d[14] = hd_few; d[15] = hd_eng; d[16] = hd_own;
#endif
    }
    // ---- counters: wave reduction, one atomic per wave and counter ----
    {
        auto flush = [&](uint32_t v, int j) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0 && v != 0u) atomicAdd(&P.ctl->counters[j], (unsigned long long)v);
        };
        if (FAST) {
            // per-lane counts: siblings of a mirror hit's child (a ray, a mirror scatter, answered from the sibling's query)
            // and siblings taken from drain helpers (a ray, a diffuse scatter)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { l_sibs += __shfl_down(l_sibs, off); l_extra += __shfl_down(l_extra, off); }
            // rays traced round by round: one level-0 ray per sample, the rest children of a diffuse or of a mirror hit
            u_specular += u_rays - u_samples - u_diffuse;
            u_rays += l_sibs + l_extra;
            u_specular += l_sibs;
            u_shared += l_sibs;
            u_diffuse += l_extra;
            if (lane == 0) {
                if (u_rays) atomicAdd(&P.ctl->counters[0], (unsigned long long)u_rays);
                if (u_samples) atomicAdd(&P.ctl->counters[1], (unsigned long long)u_samples);
                if (u_diffuse) atomicAdd(&P.ctl->counters[2], (unsigned long long)u_diffuse);
                if (u_specular) atomicAdd(&P.ctl->counters[3], (unsigned long long)u_specular);
                if (u_shared) atomicAdd(&P.ctl->counters[4], (unsigned long long)u_shared);
            }
        } else {
            flush(n_rays, 0); flush(n_samples, 1); flush(n_diffuse, 2); flush(n_specular, 3);
        }
    }
}

#undef K_ALGO
#undef K_MODE
#undef K_BUDGET
#undef K_PASS
#undef K_DEBUG

// The renderers without recursion (OnOff, Flat, Dummy: src/render.rs:57-89, :178-185) as a plain image kernel:
// one lane per pixel, its aa^2 samples in order (imagetracer.rs:112-123).  Every sample costs the same, so no
// queue and no refill: the master PCG of the lane's pixel is the wave's first pixel (scalar unit) jumped ahead by
// the lane's offset (LDS table: one step in the usual case), then per sample two jitter draws, fire_ray, the
// closest hit and the renderer's colour.
template <bool AFFINE, bool SCENE_LDS>
__global__ __launch_bounds__(RT_BLOCK) void k_simple(const PassParams P, const DScan* __restrict__ g_scan,
                                                     const DWRows* __restrict__ g_wrows, const DShade* __restrict__ g_shade,
                                                     const DMaterial* __restrict__ g_materials, const float* __restrict__ texels) {
    extern __shared__ float4 lds_raw[];
    ColdParams C = cold_params();
    const uint32_t tid = threadIdx.x;
    uint64_t* lds_jump = reinterpret_cast<uint64_t*>(lds_raw);                    // 64 x (mult, plus)
    DScan* lds_scan = reinterpret_cast<DScan*>(reinterpret_cast<unsigned char*>(lds_raw) + 1024);
    const uint32_t n_staged = SCENE_LDS ? P.n_shapes : 0u;
    DShade* lds_shade = reinterpret_cast<DShade*>(lds_scan + n_staged);
    DMaterial* lds_mats = reinterpret_cast<DMaterial*>(lds_shade + n_staged);
    if (tid < 64u) {
        uint64_t m = 1, p = 0;
        for (uint32_t i = 0; i < 6u; ++i)
            if ((tid >> i) & 1u) {
                p = p * C->pix_mult[i] + C->pix_plus[i];
                m = m * C->pix_mult[i];
            }
        lds_jump[2u * tid] = m;
        lds_jump[2u * tid + 1u] = p;
    }
    if (SCENE_LDS) {
        uint32_t* dst_s = reinterpret_cast<uint32_t*>(lds_scan);
        const uint32_t* src_s = reinterpret_cast<const uint32_t*>(g_scan);
        constexpr uint32_t SW = sizeof(DScan) / 4;
        for (uint32_t i = tid; i < n_staged * SW; i += RT_BLOCK) {
            const uint32_t sh = i / SW, j = i - sh * SW;
            dst_s[i] = (j == 13u) ? sh : src_s[i];        // staged materials are indexed by shape
        }
        uint32_t* dst_h = reinterpret_cast<uint32_t*>(lds_shade);
        const uint32_t* src_h = reinterpret_cast<const uint32_t*>(g_shade);
        for (uint32_t i = tid; i < n_staged * (uint32_t)(sizeof(DShade) / 4); i += RT_BLOCK) dst_h[i] = src_h[i];
        uint32_t* dst_m = reinterpret_cast<uint32_t*>(lds_mats);
        constexpr uint32_t MW = sizeof(DMaterial) / 4;
        for (uint32_t i = tid; i < n_staged * MW; i += RT_BLOCK) {
            const uint32_t sh = i / MW, j = i - sh * MW;
            dst_m[i] = reinterpret_cast<const uint32_t*>(g_materials + g_scan[sh].material)[j];
        }
    }
    __syncthreads();
    const DScan* __restrict__ scan = SCENE_LDS ? lds_scan : g_scan;
    const DShade* __restrict__ shade = SCENE_LDS ? lds_shade : g_shade;
    const DMaterial* __restrict__ mats = SCENE_LDS ? lds_mats : g_materials;

    const uint32_t n_items = P.n_items;
    const uint32_t i_raw = blockIdx.x * RT_BLOCK + tid;
    const uint32_t i = i_raw < n_items ? i_raw : n_items - 1u;      // lanes past the end redo the last pixel, unstored
    const uint32_t i0r = (uint32_t)__builtin_amdgcn_readfirstlane((int)(i_raw & ~63u));
    const uint32_t i0 = i0r < n_items ? i0r : n_items - 1u;
    uint32_t f0, lq0, r0, c0;
    local_to_global(C, i0, f0, lq0, r0, c0);
    const uint32_t base_f = (uint32_t)__builtin_amdgcn_readfirstlane((int)f0);
    const uint32_t base_p = (uint32_t)__builtin_amdgcn_readfirstlane((int)(r0 * C->width + c0));
    const uint64_t base_state = master_state_at(C, base_p);
    uint32_t fr, lq, grow, gcol;
    local_to_global(C, i, fr, lq, grow, gcol);
    const uint32_t gp = grow * C->width + gcol;
    const bool same = (fr == base_f) && (gp >= base_p);
    const uint64_t jbase = same ? base_state : C->master_state0;
    const uint32_t jdelta = same ? gp - base_p : gp;
    uint64_t ms;
    if (__ballot(jdelta >= 64u) == 0ull) ms = jbase * lds_jump[2u * jdelta] + lds_jump[2u * jdelta + 1u];
    else ms = master_jump(C, jbase, jdelta);

    const FrameEntry* F = frame_entry(C, fr);
    const uint32_t aa2 = C->aa * C->aa;
    float acc_r = 0.f, acc_g = 0.f, acc_b = 0.f;
    for (uint32_t s = 0; s < aa2; ++s) {
        Vec3 o, d;
        primary_ray<AFFINE>(C, F, grow, gcol, s, ms, o, d);
        float vr, vg, vb;
        if (P.algorithm == RT_ALGO_DUMMY) {
            vr = 1.0f; vg = 2.0f; vb = 3.0f;                          // render.rs:180-185
        } else {
            float t;
            Vec3 lo, ld;
            uint32_t bkind;
            int best;
            if (!SCENE_LDS && P.bvh.n_nodes != 0u)
                best = bvh_closest_hit<AFFINE>(P.bvh, g_scan, g_wrows, P.n_shapes, o, d, 1e-5f, __builtin_inff(), t, lo, ld, bkind);
            else if (AFFINE && P.scene_diag != 0u && __ballot(!diag_ray_ok(o, d)) == 0ull)
                best = scan_closest_hit_g<AFFINE, RT_SCAN_GROUP, true>(scan, g_wrows, P.n_shapes, o, d, 1e-5f, __builtin_inff(), t, lo, ld, bkind);
            else
                best = scan_closest_hit_g<AFFINE, RT_SCAN_GROUP>(scan, g_wrows, P.n_shapes, o, d, 1e-5f, __builtin_inff(), t, lo, ld, bkind);
            if (best < 0) {
                vr = P.bg[0]; vg = P.bg[1]; vb = P.bg[2];
            } else if (P.algorithm == RT_ALGO_ONOFF) {
                vr = P.fg[0]; vg = P.fg[1]; vb = P.fg[2];             // render.rs:61-66
            } else {
                // FlatRenderer: emitted + brdf colour at the hit's (u, v) (render.rs:80-85)
                const float4* shp = reinterpret_cast<const float4*>(shade + best);
                const float4 sh0 = shp[0], sh1 = shp[1];
                Vec3 hc = {sh0.x, sh0.y, sh0.z};
                Vec3 em = {sh1.x, sh1.y, sh1.z};
                const uint32_t sflags = __float_as_uint(sh0.w);
                if (sflags & 6u) {
                    const float hx = lo.x + ld.x * t, hy = lo.y + ld.y * t, hz = lo.z + ld.z * t;   // inv_ray.at(t)
                    hit_pigments(mats + (SCENE_LDS ? (uint32_t)best : g_scan[best].material), texels, sflags, bkind == RT_SHAPE_SPHERE, hx, hy, hz, hc, em);
                }
                vr = em.x + hc.x; vg = em.y + hc.y; vb = em.z + hc.z;   // render.rs:83-84
            }
        }
        acc_r = acc_r + vr; acc_g = acc_g + vg; acc_b = acc_b + vb;     // imagetracer.rs:119
    }
    if (i_raw < n_items) {
        float* out = F->out + 3ull * (C->out_in_place != 0u ? grow * C->width + gcol : lq);
        const float inv_samples = C->inv_samples;                       // imagetracer.rs:121
        out[0] = acc_r * inv_samples;
        out[1] = acc_g * inv_samples;
        out[2] = acc_b * inv_samples;
    }
    if (i_raw == 0u) {
        // every sample is one top-level solve and (but for Dummy) one World::ray_intersection
        const unsigned long long n = (unsigned long long)n_items * aa2;
        P.ctl->counters[0] = (P.algorithm == RT_ALGO_DUMMY) ? 0ull : n;
        P.ctl->counters[1] = n;
    }
}

// World::ray_intersection for a batch of rays, full HitRecord (src/world.rs:32-50).
template <bool AFFINE>
__global__ __launch_bounds__(RT_BLOCK) void k_intersect(const DScan* __restrict__ scan, const DMRows* __restrict__ mrows,
                                                        const DWRows* __restrict__ wrows, uint32_t n_shapes, const DBvh bvh,
                                                        const rt_ray* __restrict__ rays, uint64_t n, rt_hit* __restrict__ hits) {
    const uint64_t i = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const rt_ray r = rays[i];
    const Vec3 ro = {r.origin[0], r.origin[1], r.origin[2]};
    const Vec3 rd = {r.dir[0], r.dir[1], r.dir[2]};
    float t;
    Vec3 lo, ld;
    uint32_t bkind;
    const int best = bvh.n_nodes != 0u
                         ? bvh_closest_hit<AFFINE>(bvh, scan, wrows, n_shapes, ro, rd, r.tmin, r.tmax, t, lo, ld, bkind)
                         : scan_closest_hit_g<AFFINE, 2>(scan, wrows, n_shapes, ro, rd, r.tmin, r.tmax, t, lo, ld, bkind);
    rt_hit h;
    h.hit = 0; h.shape = 0; h.t = 0.f;
    h.world_point[0] = h.world_point[1] = h.world_point[2] = 0.f;
    h.normal[0] = h.normal[1] = h.normal[2] = 0.f;
    h.uv[0] = h.uv[1] = 0.f;
    if (best >= 0) {
        const HitRec rec = scan_hit_record<AFFINE>(scan + best, mrows + best, wrows + best, ro, rd, t, true, true);
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
        case 5: r = sqrt_rn(a[i]); break;                 // the hot path's sqrt (rt_device.h)
        case 6: r = sqrt_rn_normal(a[i]); break;          // its short form: 0, or >= 2^-96
        case 7: r = rt_maxf(a[i], b[i]); break;           // Rust f32::max (render.rs:148, :150)
        case 8: r = __uint_as_float(rt_f32_to_u32_sat(a[i])); break;   // Rust `as u32`, the result's bits (material.rs:56-57, :88-89)
        case 9: r = atan2_est(a[i], b[i]); break;         // the single-precision estimates behind the pigment cells (rt_device.h)
        case 10: r = acos_est(a[i]); break;
        default: r = __builtin_sqrtf(a[i]); break;
    }
    out[i] = r;
}

// BRDF::scatter_ray for a batch (src/material.rs:161-186, :225-246, :265-284): the render kernels' own
// diffuse_scatter_dir / specular_scatter_dir, with the origin, tmin, tmax and depth the reference gives the new ray.
__global__ __launch_bounds__(RT_BLOCK) void k_scatter(uint32_t brdf, const float* __restrict__ rf, const float* __restrict__ incoming,
                                                      const float* __restrict__ point, const float* __restrict__ normal,
                                                      uint32_t depth, uint64_t n, rt_ray* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const Vec3 nrm = {normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]};
    Vec3 d;
    rt_ray r;
    if (brdf == RT_BRDF_DIFFUSE) {
        d = diffuse_scatter_dir(rf[2 * i], rf[2 * i + 1], nrm);
        r.tmin = 1.0e-3f;                                   // material.rs:181
    } else {
        const Vec3 in = {incoming[3 * i], incoming[3 * i + 1], incoming[3 * i + 2]};
        d = specular_scatter_dir(in, nrm);
        r.tmin = 1e-5f;                                     // Ray::default (ray.rs:50)
    }
    r.origin[0] = point[3 * i]; r.origin[1] = point[3 * i + 1]; r.origin[2] = point[3 * i + 2];
    r.dir[0] = d.x; r.dir[1] = d.y; r.dir[2] = d.z;
    r.tmax = __builtin_inff();
    r.depth = depth;
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
