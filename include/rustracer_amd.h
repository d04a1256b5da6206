/* rustracer_amd.h — C-ABI of the MI355X render core (librustracer_amd.so).
 *
 * Drop-in boundary for rustracer's per-pixel render loop.  The reference has no FFI seam; the
 * narrowest call is ImageTracer::fire_all_rays (src/imagetracer.rs:105-125) whose only inputs
 * are the Camera, the World (shapes + materials), the Renderer parameters and the PCG seeds, and
 * whose only output is HdrImage::set_pixels (src/hdrimage.rs:86-96).  Every struct below is the
 * flattened, caller-owned, read-only form of the reference type named in its comment; every
 * entry point names the reference function it replaces.  All matrices are row-major 4x4 exactly
 * as Matrix.elements (src/transformation.rs:20-24).  Plain pointers and sizes only.
 *
 * All arithmetic on the render path is IEEE binary32, round-to-nearest, no FMA contraction,
 * except the PCG which is wrapping u64/u32 (src/random.rs:26-48).
 */
#ifndef RUSTRACER_AMD_H
#define RUSTRACER_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enumerations ----------------------------------------------------------------------- */
enum { RT_SHAPE_SPHERE = 0, RT_SHAPE_PLANE = 1 };                 /* src/shape.rs:61, :148 */
enum { RT_PIGMENT_UNIFORM = 0, RT_PIGMENT_CHECKERED = 1, RT_PIGMENT_IMAGE = 2 }; /* material.rs:100-104 */
enum { RT_BRDF_DIFFUSE = 0, RT_BRDF_SPECULAR = 1 };               /* material.rs:250-253 */
enum { RT_CAMERA_ORTHOGONAL = 0, RT_CAMERA_PERSPECTIVE = 1 };     /* camera.rs:148-151 */
enum { RT_ALGO_ONOFF = 0, RT_ALGO_FLAT = 1, RT_ALGO_PATHTRACER = 2, RT_ALGO_DUMMY = 3 }; /* render.rs:188-193 */

/* ---- flattened scene -------------------------------------------------------------------- */

/* Sphere / Plane { transformation, material } (src/shape.rs:61-70, :148-157).
 * Shape ORDER is semantic: World::ray_intersection keeps the earliest shape on ties
 * (src/world.rs:38). */
typedef struct rt_shape {
    uint32_t kind;      /* RT_SHAPE_* */
    uint32_t material;  /* index into the material table */
    float m[16];        /* Transformation.m    */
    float invm[16];     /* Transformation.invm */
} rt_shape;

/* Pigment enum (src/material.rs:27-104). */
typedef struct rt_pigment {
    uint32_t kind;      /* RT_PIGMENT_* */
    float c1[3];        /* uniform: color; checkered: color1 */
    float c2[3];        /* checkered: color2 */
    uint32_t steps;     /* checkered */
    uint32_t texture;   /* image: index into the texture table */
} rt_pigment;

/* Material { brdf, emitted_radiance } (src/material.rs:298-303); the BRDF carries its pigment
 * (DiffuseBRDF.pigment / SpecularBRDF.pigment, material.rs:136-139, :190-195).
 * SpecularBRDF.threshold_angle_rad is only read by Eval::eval, which no renderer calls. */
typedef struct rt_material {
    uint32_t brdf;              /* RT_BRDF_* */
    rt_pigment brdf_pigment;
    rt_pigment emitted;
} rt_material;

/* ImagePigment.hdr_img (src/material.rs:42-45): HdrImage pixel order, row 0 = top,
 * 3 floats per pixel (src/hdrimage.rs:21-29). */
typedef struct rt_texture {
    uint32_t w, h;
    const float* rgb;   /* w*h*3, host memory */
} rt_texture;

/* OrthogonalCamera / PerspectiveCamera (src/camera.rs:25-30, :82-89).  Only Transformation.m is
 * read by Transformation * Ray (src/transformation.rs:197-209). */
typedef struct rt_camera {
    uint32_t kind;      /* RT_CAMERA_* */
    float distance;     /* perspective only */
    float aspect;
    float m[16];
} rt_camera;

/* Renderer enum + its parameters (src/render.rs:26-33, :50-55, :95-108). */
typedef struct rt_renderer {
    uint32_t algorithm; /* RT_ALGO_* */
    float bg[3];        /* bg_color */
    float fg[3];        /* OnOffRenderer.fg_color */
    uint32_t num_of_rays;
    uint32_t max_depth;
    uint32_t rr_limit;  /* russian_roulette_limit */
} rt_renderer;

/* Ray (src/ray.rs:8-21). */
typedef struct rt_ray {
    float origin[3];
    float dir[3];
    float tmin, tmax;
    uint32_t depth;
} rt_ray;

/* HitRecord without the Material clone (src/shape.rs:32-45); `shape` replaces `material`
 * (the material is scene.shapes[shape].material), hit = 0 encodes None. */
typedef struct rt_hit {
    uint32_t hit;
    uint32_t shape;
    float world_point[3];
    float normal[3];
    float uv[2];
    float t;
} rt_hit;

/* Pcg { state, inc } (src/random.rs:7-10). */
typedef struct rt_pcg {
    uint64_t state, inc;
} rt_pcg;

/* Which rows of the image this call renders: row tiles of `tile_rows` rows dealt round-robin,
 * tile j of the image belongs to part (j % n_parts).  One part per GPU; {0,1,any,0} = whole image.
 * (Replaces rayon's par_iter over pixels, src/imagetracer.rs:112-123.)
 * rotate != 0 (multi-frame launches): frame f of the launch renders part (part + f) % n_parts instead, so over
 * n_parts consecutive frames every GPU renders every part once.  With tile_rows = height / n_parts a part is one
 * contiguous block of rows that lands in the final image with a plain copy (no rt_unpack_parts) and the rotation
 * balances the blocks' unequal costs over the frames.  Needs equal parts: height % (tile_rows * n_parts) == 0.
 * in_place != 0: d_out_rgb is the FULL image (width * height * 3 floats) and every pixel of the part is stored at its
 * global (row, column) -- the buffer may be another GPU's memory (an IPC mapping: the stores cross xGMI beside the
 * render) or pinned host memory, so the parts of all GPUs assemble in one image with no copy and no rt_unpack_parts.
 * in_place == 0: d_out_rgb holds only the part, rows in part-local order. */
typedef struct rt_partition {
    uint32_t part, n_parts, tile_rows;
    uint32_t rotate;
    uint32_t in_place;
} rt_partition;

/* Counters of one render call. */
typedef struct rt_stats {
    uint64_t rays;        /* World::ray_intersection calls (src/world.rs:32)            */
    uint64_t samples;     /* top-level Renderer::solve calls (src/imagetracer.rs:119)   */
    uint64_t sphere_tests, plane_tests; /* Shape::ray_intersection calls by kind        */
    uint64_t diffuse_scatters, specular_scatters; /* BRDF::scatter_ray calls whose ray is traced */
    double kernel_ms;     /* device time of the render kernels of this call (HIP events) */
    double total_ms;      /* host wall time of the call */
    uint64_t rays_shared; /* of `rays`: children of a mirror hit that took the answer of the sibling traced before them
                             (the same ray, src/material.rs:225-246) without running the shape tests again -- in scenes
                             traced through the hierarchy: without the traversal, only the winner's own test; counted in
                             `rays`, `sphere_tests` and `plane_tests` as the reference counts them */
} rt_stats;

typedef struct rt_scene rt_scene; /* device-resident World + textures (opaque) */

/* ---- entry points ----------------------------------------------------------------------- */

/* Select the GPU for this process (one process per GPU). 0 on success. */
int rt_init(int device);

/* Thread-local message of the last failing call. */
const char* rt_last_error(void);

/* Upload World (src/world.rs:16-21) + materials + textures to HBM.  NULL on failure. */
rt_scene* rt_scene_create(const rt_shape* shapes, uint32_t n_shapes,
                          const rt_material* materials, uint32_t n_materials,
                          const rt_texture* textures, uint32_t n_textures);
void rt_scene_destroy(rt_scene* scene);

/* Scheduling of rt_render on this scene (never changes a pixel; no counterpart in the reference,
 * where rayon schedules pixels, src/imagetracer.rs:112-123).  A path-traced frame is a sequence of
 * passes: a pixel runs at most budgets[p] rays deep into its recursion in pass p, then its path state
 * is parked in HBM and picked up, compacted into full wavefronts, by pass p+1; the pass after the
 * last budget has no limit.  n_budgets == 0 restores the default: the library chooses (one pass for
 * path-traced frames of an affine scene of up to 96 shapes from half a pixel per GPU lane on, and for
 * every launch of more than about a dozen pixels per lane; short passes otherwise);
 * budgets = {0} renders every pixel in one pass (best throughput when several frames are in flight
 * on different streams). */
int rt_scene_set_schedule(rt_scene* scene, const uint32_t* budgets, uint32_t n_budgets);

/* How World::ray_intersection runs on this scene (no counterpart in the reference; diagnostics and tests):
 * n_nodes > 0 means the exact-winner hierarchy is used (scenes of more than 96 shapes with at least 64
 * bounded ones), n_unbounded is the number of shapes outside it that every ray tests.  0 on success. */
int rt_scene_hierarchy_info(const rt_scene* scene, uint32_t* n_nodes, uint32_t* n_unbounded);

/* Number of pixels of `part` (rows * width), i.e. the length/3 of the buffer rt_render fills. */
uint64_t rt_partition_pixels(uint32_t width, uint32_t height, const rt_partition* part);

/* Replaces ImageTracer::fire_all_rays (src/imagetracer.rs:105-125) with everything resident in
 * HBM: renders the pixels of `part` into d_out_rgb (DEVICE memory, part-local row order, 3 f32
 * per pixel, row 0 = top) on `stream` (hipStream_t, NULL = default stream) and returns without
 * synchronising unless `stats` is non-NULL (then it blocks and fills the counters). */
int rt_render(const rt_scene* scene, const rt_camera* camera, const rt_renderer* renderer,
              uint32_t width, uint32_t height, uint64_t init_state, uint64_t init_seq,
              uint32_t antialiasing_level, const rt_partition* part /* NULL = whole image */,
              float* d_out_rgb, void* stream, rt_stats* stats /* nullable */);

/* rt_render for n_frames frames of the same World in ONE launch (an animation: the reference renders
 * its 360-frame demo.gif by calling `demo --angle-deg a` once per frame, makefile:13-15,
 * src/main.rs:197): frame f uses cameras[f] and fills d_out_rgb[f] (DEVICE buffers; the array of
 * pointers itself is host memory).  Every frame is ImageTracer::fire_all_rays with the same seeds,
 * renderer and partition; pixels of all frames share one queue, which keeps the GPU full where a
 * single frame leaves most lanes waiting on a few deep pixels.  rt_render == n_frames 1.
 * `stats` counts all frames of the launch. */
int rt_render_frames(const rt_scene* scene, const rt_camera* cameras, uint32_t n_frames,
                     const rt_renderer* renderer, uint32_t width, uint32_t height,
                     uint64_t init_state, uint64_t init_seq, uint32_t antialiasing_level,
                     const rt_partition* part /* NULL = whole image */,
                     float* const* d_out_rgb, void* stream, rt_stats* stats /* nullable */);

/* rt_render for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  It has no counterpart in the reference.  Same arguments, same pixels as rt_render,
 * but the call allocates nothing on host or device: the launch's control block (queue head, ray counters) and, for
 * recursions too deep for LDS, the recursion stack live in a file-scope device array of RT_NOALLOC_SCRATCH_BYTES
 * that every GPU the library runs on holds until the process ends (its address is resolved per device by rt_init,
 * or by the first call on a device).  Always one pass, one lane per pixel; no stats.  max_depth beyond what the
 * scratch holds for one 256-lane block (RT_NOALLOC_MAX_DEPTH) is rejected with an error.  Calls on one GPU share
 * that GPU's array: calls that overlap on one GPU (different streams) race, undetected; calls on one stream are ordered.
 * Under stream capture the device must have been visited before (rt_init(device) or an uncaptured call): resolving
 * the array's address is not a capturable operation, and a first use under capture is refused with an error. */
#define RT_NOALLOC_SCRATCH_BYTES (8u * 1024u * 1024u + 4096u)
#define RT_NOALLOC_MAX_DEPTH 513u
int rt_render_noalloc(const rt_scene* scene, const rt_camera* camera, const rt_renderer* renderer,
                      uint32_t width, uint32_t height, uint64_t init_state, uint64_t init_seq,
                      uint32_t antialiasing_level, const rt_partition* part /* NULL = whole image */,
                      float* d_out_rgb, void* stream);

/* Inverse of the row-tile partition: scatter `n_parts` gathered part buffers (each padded to
 * `part_stride_pixels` pixels, DEVICE memory) into the full image d_image_rgb (DEVICE). */
int rt_unpack_parts(const float* d_parts_rgb, uint64_t part_stride_pixels, uint32_t n_parts,
                    uint32_t tile_rows, uint32_t width, uint32_t height, float* d_image_rgb,
                    void* stream);

/* Blocking one-shot with host buffers: exactly ImageTracer::new + fire_all_rays +
 * HdrImage::set_pixels (src/imagetracer.rs:40, :105-125; src/hdrimage.rs:86-96).
 * out_rgb: width*height*3 floats, HdrImage order (row 0 = top), fully overwritten.
 * Multi-GPU fan-out is internal: the rows are dealt round-robin over the GPUs in use (RUSTRACER_DEVICES = a list of
 * device indices, or RUSTRACER_NUM_DEVICES = a count; default: one GPU per 2^21 samples of the frame, the current device
 * first -- only the current device when the process is one rank of a multi-process job, WORLD_SIZE > 1 or LOCAL_RANK set),
 * every GPU renders its part and copies it to the host over its own link, host threads place the rows in out_rgb.
 * Between calls the library keeps, per GPU, the uploaded World, a pinned staging buffer (and, for the renderers that
 * copy back, a device image), a stream and its events until rt_fire_release().  The World is reused when its tables
 * are unchanged: a 128-bit content key over shapes, materials and texels selects it, the shape and material tables
 * are then compared byte for byte; texels are trusted to the key (a colliding texel change is the one unverified case).
 * stats->kernel_ms is the longest GPU's. */
int rt_fire_all_rays(const rt_shape* shapes, uint32_t n_shapes,
                     const rt_material* materials, uint32_t n_materials,
                     const rt_texture* textures, uint32_t n_textures,
                     const rt_camera* camera, const rt_renderer* renderer,
                     uint32_t width, uint32_t height, uint64_t init_state, uint64_t init_seq,
                     uint32_t antialiasing_level, float* out_rgb, rt_stats* stats /* nullable */);

/* Frees what rt_fire_all_rays keeps between calls on every GPU it has used: the uploaded World, the device image,
 * the pinned staging buffer, the stream and its events (about 400 MB per GPU after an 8K frame).  The next
 * rt_fire_all_rays builds them again.  No counterpart in the reference (its ImageTracer owns nothing but the image). */
void rt_fire_release(void);

/* Which GPUs the last rt_fire_all_rays of this thread used and how long each one's render kernels ran (the
 * fan-out is internal to that call; no counterpart in the reference, where rayon spreads the pixels over the cores,
 * src/imagetracer.rs:112-123).  Fills up to `cap` entries of devices[] / kernel_ms[] (either may be NULL) and
 * returns the number of parts of that call (one per device in use). */
uint32_t rt_last_fire_devices(int32_t* devices, double* kernel_ms, uint32_t cap);

/* Replaces World::ray_intersection (src/world.rs:32-50) for a batch of host rays. */
int rt_world_ray_intersection(const rt_scene* scene, const rt_ray* rays, uint64_t n_rays,
                              rt_hit* hits);

/* Replaces Renderer::solve (src/render.rs:195-205) for a batch of host (ray, pcg) pairs; the
 * advanced generators are written back to `pcgs`, colors to out_rgb (3 per ray). */
int rt_renderer_solve(const rt_scene* scene, const rt_renderer* renderer, const rt_ray* rays,
                      rt_pcg* pcgs, uint64_t n_rays, float* out_rgb);

/* Replaces Pcg::new + n x Pcg::random (src/random.rs:26-43) for a batch of seeds on the GPU:
 * out[i*n_draws + k] = k-th u32 of Pcg::new(init_state[i], init_seq[i]). */
int rt_pcg_random(const uint64_t* init_state, const uint64_t* init_seq, uint64_t n_seeds,
                  uint32_t n_draws, uint32_t* out);

/* Replaces BRDF::scatter_ray (src/material.rs:265-284; Diffuse :161-186 with create_onb_from_z, src/normal.rs:143-158;
 * Specular :225-246) for a batch: rf = the two PCG floats per item (2n), incoming_dir / interaction_point / normal
 * 3n floats each (the normal is used raw, as the reference hands it over), `depth` the new rays' depth. */
int rt_brdf_scatter_ray(uint32_t brdf, const float* rf, const float* incoming_dir, const float* interaction_point,
                        const float* normal, uint32_t depth, uint64_t n, rt_ray* out);

/* Evaluate include/rt_strict_math.h on the GPU (fn: 0 sin, 1 cos, 2 atan2(y=a,x=b), 3 acos,
 * 4 a/b, 5 sqrt(a) as the render kernels compute it, 6 its short form for a == 0 or a >= 2^-96,
 * 7 rt_maxf of a and b, 8 the bits of rt_f32_to_u32_sat of a); used to prove host/device bit-equality of the shared math.
 * fn 9 / 10: the render kernels' single-precision ESTIMATES of atan2(a, b) / acos(a) that pre-select pigment cells
 * (never a pixel value: csrc/rt_device.h, hit_pigments); the tests hold them to their stated error bounds. */
int rt_strict_math_eval(int fn, const float* a, const float* b, uint64_t n, float* out);

#ifdef __cplusplus
}
#endif
#endif /* RUSTRACER_AMD_H */
