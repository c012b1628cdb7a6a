/* hybrid_rendering_amd — C ABI of the MI355X-native ray-trace + denoise back end.
 *
 * Drop-in boundary for the four hot-path pass classes of diharaw/hybrid-rendering:
 *   RayTracedShadows      (src/ray_traced_shadows.h:7-142)
 *   RayTracedAO           (src/ray_traced_ao.h:7-126)
 *   RayTracedReflections  (src/ray_traced_reflections.h:8-150)
 *   DDGI                  (src/ddgi.h:6-135)
 * Each pass keeps the reference's shape: create(backend, resolution, scale) -> render(cmd_buf[, ddgi])
 * -> output_ds().  Here the Vulkan command buffer becomes a HIP stream (void* = hipStream_t), a
 * descriptor set becomes an hr_image_view (device pointer + extent + format), and the per-frame
 * UBO (src/common.h:161-179) is passed by value inside hr_frame_inputs.
 *
 * Conventions
 *   - plain C types only; every entry point returns hr_status (0 = HR_OK); no exceptions cross the ABI.
 *   - all image pointers are DEVICE pointers to tightly packed row-major images in the reference's
 *     formats (RGBA8 / RGBA16F / R32F depth / R32_UINT masks); the caller owns inputs, the pass owns
 *     intermediates and outputs; output views stay valid until the next render()/destroy().
 *   - calls on one handle must be externally serialised; work is asynchronous on the given stream.
 *   - "band" fields describe row tiling across GPUs: a pass instance owns rows [band_y0, band_y1) of
 *     a full_height-row frame and its images hold rows [alloc_y0, alloc_y1) (band + halo).
 */
#ifndef HR_API_H
#define HR_API_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int hr_status;
enum
{
    HR_OK                = 0,
    HR_ERR_INVALID_ARG   = 1,
    HR_ERR_HIP           = 2, /* a HIP runtime call failed; see hr_last_error() */
    HR_ERR_NO_DEVICE     = 3,
    HR_ERR_OUT_OF_MEMORY = 4,
    HR_ERR_UNSUPPORTED   = 5,
    HR_ERR_TIMEOUT       = 6, /* hr_comm (loopback): a neighbour never posted its side of a collective */
    HR_ERR_COMM          = 7  /* an RCCL call failed; see hr_last_error() */
};

const char* hr_status_string(hr_status s);
const char* hr_last_error(void); /* thread-local detail of the last failure */
const char* hr_version(void);
/* Revision of this header's struct layouts and entry points.  The parameter structs are passed by pointer WITHOUT a size field, so a host
 * built against another revision must not call into this library: compare hr_api_revision() with HR_API_REVISION once at start-up
 * (hr::Context does).  Revision 3 = round 3: hr_*_denoise, hr_hybrid_frame, ticketed hr_comm, HR_ERR_TIMEOUT / HR_ERR_COMM. */
/* revision 4 (round 4): + hr_bvh_selfcheck, hr_ddgi_trace_stats, hr_reflections_trace_stats; structs unchanged */
/* revision 5 (round 5): + hr_shadows_trace_stats_timed, hr_shadows_launch_order, hr_ao_launch_order; structs unchanged; every hr_*_create
 * returns with its images zero-filled (it waits for the fills), so a first render() on any stream is ordered after them */
/* revision 6 (round 6): + hr_scene_id, hr_ddgi_grid_from_extents, hr_ddgi_set_normal_bias, hr_scene_create_instanced,
 * hr_scene_update_instances; structs unchanged */
#define HR_API_REVISION 6
int32_t hr_api_revision(void);

/* ---- formats -------------------------------------------------------------------------------- */
typedef enum
{
    HR_FORMAT_R32_UINT = 1, /* packed 8x4 visibility masks (ray_traced_shadows.cpp:165) */
    HR_FORMAT_R16F     = 2,
    HR_FORMAT_RG16F    = 3,
    HR_FORMAT_RGBA16F  = 4,
    HR_FORMAT_R32F     = 5,
    HR_FORMAT_RGBA8    = 6
} hr_format;

typedef struct
{
    void*     data; /* device pointer */
    int32_t   width, height;
    int32_t   row_pitch_bytes;
    hr_format format;
} hr_image_view;

/* ---- per-frame inputs (replaces CommonResources + GBuffer descriptor sets) ------------------ */
/* src/common.h:106-158 */
typedef struct
{
    float data0[4]; /* xyz direction TO the light (main.cpp:963), w intensity */
    float data1[4]; /* xyz position, w radius */
    float data2[4]; /* xyz colour */
    float data3[4]; /* x type (0 directional, 1 point, 2 spot), y cos(outer), z cos(inner) */
} hr_light;

/* src/common.h:161-179; column-major mat4, 416 bytes */
typedef struct
{
    float    view_inverse[16];
    float    proj_inverse[16];
    float    view_proj_inverse[16];
    float    prev_view_proj[16];
    float    view_proj[16];
    float    cam_pos[4];
    float    current_prev_jitter[4];
    hr_light light;
} hr_ubo;

/* One G-buffer level (g_buffer.frag:96-111): the pass reads the level matching its scale
 * (g_buffer_mip), the upsample stage additionally reads level 0. */
typedef struct
{
    const void*  gb1;   /* RGBA8   albedo.rgb, metallic            (may be NULL where unused) */
    const void*  gb2;   /* RGBA16F oct normal.xy, motion.xy                                 */
    const void*  gb3;   /* RGBA16F roughness, curvature, mesh id, linear z (sky: w = -1)     */
    const float* depth; /* R32F    ndc depth, sky == 1.0                                    */
    int32_t      width, height;
} hr_gbuffer_level;

typedef struct
{
    hr_gbuffer_level cur;       /* at the pass resolution (mip = scale)  — GBuffer::output_ds()  */
    hr_gbuffer_level prev;      /* previous frame, same level            — GBuffer::history_ds() */
    hr_gbuffer_level cur_full;  /* level 0 (only read by upsample when scale != 0)               */
    hr_ubo           ubo;
    uint32_t         num_frames;          /* CommonResources::num_frames                          */
    int32_t          ping_pong;           /* CommonResources::ping_pong                           */
    const uint8_t*   sobol;               /* device, [256][4] RGBA8   (blue_noise.cpp:5-19)        */
    const uint8_t*   scrambling_ranking;  /* device, [128][128][4] RGBA8                          */
    float            z_buffer_params[4];  /* main.cpp:253-254 (AO blur only)                      */
} hr_frame_inputs;

/* ---- context & scene ------------------------------------------------------------------------- */
typedef struct hr_ctx   hr_ctx;
typedef struct hr_scene hr_scene;

hr_status hr_ctx_create(int device_ordinal, hr_ctx** out);
hr_status hr_ctx_destroy(hr_ctx* ctx);
int32_t   hr_ctx_device(const hr_ctx* ctx);   /* HIP device ordinal the context was created on (-1: NULL) */

/* Replaces dw::RayTracedScene (BLAS/TLAS build, main.cpp:74 build_tlas, common.cpp:355-521
 * initialize_for_ray_tracing): flattens instances to world space, builds the compressed 8-wide BVH
 * on the host and uploads it.  All host pointers; data is copied. */
typedef struct
{
    const uint8_t* rgba8;         /* [height][width][4] */
    int32_t        width, height;
} hr_texture;

typedef struct
{
    const float*    positions;    /* [n_tris][3][3] world-space vertex positions               */
    const float*    normals;      /* [n_tris][3][3] vertex normals, or NULL (geometric)         */
    const uint32_t* tri_material; /* [n_tris] or NULL                                           */
    const uint32_t* tri_mesh_id;  /* [n_tris] or NULL                                           */
    int32_t         n_tris;
    const float*    materials;    /* [n_materials][8]: albedo rgb, metallic, roughness, emissive rgb */
    int32_t         n_materials;
    /* Textured materials (scene_descriptor_set.glsl:20-27 Material.texture_indices0/1, :168-220 fetch_*), all optional
     * (NULL / 0 = the material constants above).  Sampling is what a ray-tracing stage gets from texture(): level 0;
     * the sampler (created in the reference's un-vendored framework) is pinned to bilinear, REPEAT, UNORM8 = b / 255. */
    const float*      uvs;               /* [n_tris][3][2] texture coordinates                                            */
    const float*      tangents;          /* [n_tris][3][3] vertex tangents (normal-mapped materials only)                 */
    const int32_t*    material_textures; /* [n_materials][6]: albedo, normal, roughness, metallic texture index (-1 = none),
                                            roughness channel, metallic channel (texture_indices1.z / .w)                 */
    const hr_texture* textures;          /* [n_textures] host RGBA8 images, copied                                        */
    int32_t           n_textures;
} hr_scene_desc;

typedef struct
{
    int32_t  n_tris, n_nodes, max_depth;
    uint64_t node_bytes, tri_bytes;
    float    bounds_lo[3], bounds_hi[3];
    float    box_pad;
} hr_scene_info;

/* HR_ERR_INVALID_ARG: a tri_material entry >= n_materials or a texture index >= n_textures (both are dereferenced by the hit
 * shading); HR_ERR_UNSUPPORTED: >= 2^23 BVH nodes, >= 2^26 triangle references, or a BVH deeper than the traversal stack (64 levels — the builder caps its
 * depth, so no triangle soup reaches it); HR_ERR_OUT_OF_MEMORY: host or device allocation failed.  Never throws. */
hr_status hr_scene_create(hr_ctx* ctx, const hr_scene_desc* desc, hr_scene** out);
hr_status hr_scene_get_info(const hr_scene* scene, hr_scene_info* info);
/* dw::Scene::id(): unique per hr_scene_create within the process, never 0 (0: NULL scene) — DDGI::render re-derives its probe grid when it
 * changes (ddgi.cpp:93-95) */
uint64_t  hr_scene_id(const hr_scene* scene);
/* Host only (no device needed): the shape of the BVH hr_scene_create would build over `positions` ([n_tris][3][3] floats). */
hr_status hr_bvh_build_info(const float* positions, int32_t n_tris, hr_scene_info* info);
/* Host only: builds the same BVH and descends it from `samples_per_triangle` points of every triangle (corners, edge midpoints,
 * centroid, then pseudo-random interior points); *uncovered = the (triangle, point) pairs that reach no leaf holding the triangle.
 * 0 for a correct tree: the builder references a triangle from several leaves (spatial splits) and the pieces must cover it. */
hr_status hr_bvh_selfcheck(const float* positions, int32_t n_tris, int32_t samples_per_triangle, int64_t* uncovered);

/* ---- instanced scenes: dw::RayTracedScene as the reference holds it (scene_descriptor_set.glsl:30-34 Instance { mat4 model_matrix; uint
 * mesh_idx; }, :102-131 fetch_hit_info / fetch_triangle through instance.mesh_idx, :150-160 transform_vertex) with the per-frame
 * acceleration-structure update of main.cpp:74 (build_tlas).
 *
 * MI355X layout: ONE 8-wide BVH in world space.  Every instance owns a private copy of its mesh's subtree (the topology is built once per
 * mesh in object space; 288 GB of HBM pay for the copies) under a top level over the instance roots, so every trace kernel keeps the
 * single-level walk of hr_scene_create — no ray transform at an instance boundary, no second stack.  hr_scene_update_instances moves
 * instances on the GPU: world-space vertices = model_matrix * (x, y, z, 1) rounded per operation (((m0 x + m1 y) + m2 z) + m3), then
 * every node box is refitted bottom-up from the triangles (one launch, children before parents by arrival counters), all on `stream`.
 * A triangle is hit iff the watertight test accepts its world-space vertices, so the answer of a query is the one a flattened
 * hr_scene_create over the same world-space vertices gives (any-hit: a function of the geometry; closest hit: smallest t, ties to the
 * smallest triangle index = instance order, then mesh order).  Hit shading interpolates the OBJECT-space vertex attributes and then
 * applies the instance's matrix: position = model_matrix * p, normal / tangent = normalize(mat3(model_matrix) * normalize(n)) — the
 * operation order of interpolated_vertex + transform_vertex.  Hit records name the global triangle index: triangles of instance i
 * follow those of instance i - 1, in mesh order. */
typedef struct
{
    const float*    positions;    /* [n_tris][3][3] OBJECT-space vertex positions                                  */
    const float*    normals;      /* [n_tris][3][3] or NULL (geometric)                                            */
    const uint32_t* tri_material; /* [n_tris] or NULL: index into hr_instanced_scene_desc.materials                */
    const float*    uvs;          /* [n_tris][3][2] or NULL                                                        */
    const float*    tangents;     /* [n_tris][3][3] or NULL                                                        */
    int32_t         n_tris;
} hr_mesh_desc;

typedef struct
{
    float    model_matrix[16];    /* column-major mat4, object -> world (Instance::model_matrix)                   */
    uint32_t mesh_idx;            /* Instance::mesh_idx                                                            */
    uint32_t mesh_id;             /* the value g_buffer.frag:109 writes to GB3.z for this instance's pixels        */
} hr_instance;

typedef struct
{
    const hr_mesh_desc* meshes;
    int32_t             n_meshes;
    const hr_instance*  instances;
    int32_t             n_instances;
    const float*        materials;          /* as hr_scene_desc */
    int32_t             n_materials;
    const int32_t*      material_textures;  /* as hr_scene_desc (NULL: untextured) */
    const hr_texture*   textures;
    int32_t             n_textures;
} hr_instanced_scene_desc;

/* Builds the per-mesh topologies on the host, lays out one subtree per instance + the top level, uploads, and runs the first update with
 * the instances' matrices.  Errors as hr_scene_create; HR_ERR_INVALID_ARG also for mesh_idx >= n_meshes or a non-finite matrix. */
hr_status hr_scene_create_instanced(hr_ctx* ctx, const hr_instanced_scene_desc* desc, hr_scene** out);
/* model_matrices: HOST [n_instances][16] column-major (copied before the call returns).  Enqueues on `stream`: matrix upload, vertex
 * transform, bottom-up refit; no host synchronisation.  Passes rendered on the same stream afterwards see the new geometry; per-scene
 * caches of the passes (AO entry-node table) notice the change.  HR_ERR_INVALID_ARG for a scene from hr_scene_create. */
hr_status hr_scene_update_instances(hr_scene* scene, const float* model_matrices, void* stream);
int32_t   hr_scene_instance_count(const hr_scene* scene);   /* 0 for a scene from hr_scene_create */
/* Introspection (tests, tools): copies the device BVH to the host after synchronising the device — hr_scene_info.node_bytes of 80-byte nodes
 * and .tri_bytes of 48-byte triangle references (layouts: csrc/bvh.h).  Either pointer may be NULL. */
hr_status hr_scene_read_bvh(const hr_scene* scene, void* nodes_out, void* tris_out);
hr_status hr_scene_destroy(hr_scene* scene);

/* Raw ray queries against the scene (replace rayQueryEXT / traceRayEXT; used by tests and tools).
 * rays: device [n][8] floats = origin xyz, t_max, direction xyz, t_min.
 * any-hit: out_occluded device [n] uint8.  closest: out_tuv device [n][3] (t,u,v), out_prim device [n] int32 (-1 miss).
 * stats (nullable, device [2] uint64): accumulates nodes visited, triangles tested. */
hr_status hr_trace_any_hit(const hr_scene* scene, int64_t n, const float* rays, uint8_t* out_occluded, uint64_t* stats, void* stream);
hr_status hr_trace_closest_hit(const hr_scene* scene, int64_t n, const float* rays, float* out_tuv, int32_t* out_prim, void* stream);

/* G-buffer synthesis by primary-ray casting (tooling: stands in for the raster GBuffer pass,
 * src/g_buffer.cpp + shaders/g_buffer.frag:86-112, so bench inputs can be produced on the GPU). */
hr_status hr_gbuffer_raycast(const hr_scene* scene, const hr_ubo* ubo, int32_t width, int32_t height, void* gb1, void* gb2, void* gb3,
                             float* depth, void* stream);
/* Nearest-filtered mip `level` (1..8) of a G-buffer level: dst (width >> level, height >> level) texel (x, y) = src texel
 * (x << level, y << level) — g_buffer.cpp:240-243 (vkCmdBlitImage with VK_FILTER_NEAREST).  gb1 is optional.  The
 * half- and quarter-resolution passes read these through hr_frame_inputs.cur / .prev. */
hr_status hr_gbuffer_mip_nearest(const hr_gbuffer_level* src, const hr_gbuffer_level* dst, int32_t level, void* stream);

/* ---- common pass plumbing ---------------------------------------------------------------------- */
typedef enum
{
    HR_SCALE_FULL_RES    = 0, /* RAY_TRACE_SCALE_FULL_RES    (common.h) */
    HR_SCALE_HALF_RES    = 1,
    HR_SCALE_QUARTER_RES = 2
} hr_scale;

/* RayTracedShadows::OutputType etc. (ray_traced_shadows.h:10-16) */
typedef enum
{
    HR_OUTPUT_RAY_TRACE             = 0,
    HR_OUTPUT_TEMPORAL_ACCUMULATION = 1,
    HR_OUTPUT_ATROUS                = 2, /* AO: bilateral blur */
    HR_OUTPUT_UPSAMPLE              = 3
} hr_output_kind;

/* Row band owned by this GPU.  Zero-initialised = whole frame. */
typedef struct
{
    int32_t band_y0, band_y1; /* rows owned, in pass-resolution pixels; 0,0 = whole frame */
    int32_t halo;             /* rows recomputed redundantly on each side (multiple of 8); the a-trous chain needs 15 */
    int32_t history_halo;     /* rows on each side whose HISTORY (and G-buffer) is readable: the driver fills them by
                                 neighbour exchange after every frame (>= halo + the largest motion in rows) */
} hr_band;

#define HR_MAX_STAGES 16
typedef struct
{
    int32_t     n_stages;
    const char* name[HR_MAX_STAGES];
    float       ms[HR_MAX_STAGES];     /* hipEvent elapsed, averaged over the frames profiled since the previous call (up to 512) */
    uint64_t    bytes[HR_MAX_STAGES];  /* algorithmic bytes of the stage */
} hr_stage_times;

/* ---- RayTracedShadows (src/ray_traced_shadows.h) ----------------------------------------------- */
typedef struct hr_shadows hr_shadows;

/* member defaults: ray_traced_shadows.h:52,69-70,101-107 */
typedef struct
{
    int32_t denoise;            /* m_denoise = true                              */
    float   bias;               /* RayTrace::bias = 0.5                          */
    float   alpha;              /* TemporalAccumulation::alpha = 0.01            */
    float   moments_alpha;      /* TemporalAccumulation::moments_alpha = 0.2     */
    float   phi_visibility;     /* ATrous::phi_visibility = 10                   */
    float   phi_normal;         /* ATrous::phi_normal = 32                       */
    float   sigma_depth;        /* ATrous::sigma_depth = 1                       */
    float   power;              /* ATrous::power = 1.2                           */
    int32_t radius;             /* ATrous::radius = 1                            */
    int32_t filter_iterations;  /* ATrous::filter_iterations = 4  (1..5)         */
    int32_t feedback_iteration; /* ATrous::feedback_iteration = 1                */
    int32_t exact;              /* 1 (default): every fp32 operation individually, correctly rounded — stage images equal the oracle
                                   and the reference's shaders BIT FOR BIT (the parity mode).  0: tolerance mode for production — the
                                   denoise / resolve kernels use the hardware's rcp / rsq / sqrt / exp / log and fused multiply-adds
                                   (2-4x faster);
                                   every fp16 image is within 2 fp16 ulp of the oracle on >= 99.9 % of its texels, with relative L2 error <= 1e-3 over
                                   those texels and <= 1e-2 over ALL texels; variance channels (shadows .y, reflections .a) additionally count
                                   |diff| <= 1e-4 as equal; tile classes agree on >= 99.5 % of the tiles.  HARD CAP per texel: outside the
                                   neighbourhoods of tiles whose class differs, every texel is within 32 fp16 ulp or 2^-10 of the oracle.  Round 5:
                                   no counted exceptions for the shadows, AO, DDGI-sample and reflections-trace images — where the reference's
                                   formulas are discontinuous (a history tap's validity thresholds, reprojection.glsl:52-67; the DDGI gather's
                                   trilinear zero on a probe plane and its NaN-driven Chebyshev term where the fp16 depth moments overflow,
                                   gi_common.glsl:188-320) the tolerance-mode kernels detect the shading points at which their fast operands
                                   cannot be trusted and take the decision with the parity arithmetic.  The reflections' DENOISED images (temporal,
                                   a-trous, upsampled output) may exceed the cap on at most max(4, 2e-5 of the pixels) pixels per image (x 5 * 4^scale
                                   in a scaled pass's upsampled output), each within 512 fp16 ulp or 2^-5: the reference's luminance edge-stopping
                                   weight exp(-|dl| / (phi sqrt(1e-10 + var))) moves by e^0.6 per fp16 ulp of its input where var == 0, so a 1-ulp
                                   difference in a stored a-trous intermediate re-weights a tap of the next iteration.  The 99.9 % population bound, end to end, is likewise a property of
                                   the SEQUENCE for the reflections' a-trous and output images — var is m2 - m1^2 of two stored fp16 moments, so one
                                   ulp of a stored moment moves a small variance by more than its size and re-weights every tap around it: where
                                   var is tiny over a region (the first frames of a history), the 1-ulp colours the trace image's fast DDGI gathers are
                                   allowed reach it through the next frame's moments: 5 of 1214 fuzzed sequences measure 99.86 - 99.89 % (all 1214
                                   >= 99.8 %; the temporal kernel's own moment path keeps the parity arithmetic); stage by stage — the a-trous and upsample kernels against the oracle's
                                   stage run on the SAME input image — the bound holds with no counted exception at all
                                   (tests/test_gpu_tolerance.py compare16; fuzz logs: profiles/r5_*);
                                   visibility masks, ray counts and traversal are identical in both modes.
                                   Tolerance mode also reprojects from the pass's own copy of the previous frame's geometry (normal, mesh id,
                                   linear z — written by its temporal kernel) instead of in->prev.gb2 / gb3 whenever in->prev.gb2 / gb3 are the
                                   pointers the previous call received as in->cur.gb2 / gb3 (the reference's G-buffer ping-pong,
                                   g_buffer.cpp:208-211): same values, 5 fewer gathers per pixel.  A caller that rewrites those images between
                                   the two calls must call hr_*_reset_history (or pass other pointers). */
} hr_shadows_params;

void      hr_shadows_default_params(hr_shadows_params* p);
hr_status hr_shadows_create(hr_ctx* ctx, int32_t full_width, int32_t full_height, hr_scale scale, const hr_band* band, hr_shadows** out);
/* RayTracedShadows::render (ray_traced_shadows.cpp:100-116) */
hr_status hr_shadows_render(hr_shadows* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_shadows_params* params, void* stream);
/* RayTracedShadows::output_ds (ray_traced_shadows.cpp:135-155) */
hr_status hr_shadows_output(hr_shadows* p, hr_output_kind kind, hr_image_view* view);
/* m_first_frame = true (ray_traced_shadows.cpp:938-968) */
hr_status hr_shadows_reset_history(hr_shadows* p);
hr_status hr_shadows_destroy(hr_shadows* p);
/* Stage-level entry points (the private methods ray_trace / temporal_accumulation / a_trous_filter /
 * upsample, ray_traced_shadows.cpp:972-1255) so a multi-GPU driver can exchange halos between them.
 * Launch order: the trace kernels of the shadows, AO and reflections passes record how long each 8x8 tile's wave lived and launch
 * the next frame's tiles heaviest first (the sort runs inside the pass's tolerance-mode temporal launch, else at the start of the next
 * trace call).  Outputs do not depend on it.  Like the visibility mask — which the next trace call overwrites and the temporal stage
 * reads — this state asks for what a frame loop does anyway: a pass's next *_ray_trace call is stream-ordered after its last
 * *_temporal call. */
hr_status hr_shadows_ray_trace(hr_shadows* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_shadows_params* params, void* stream);
/* everything of render() after the trace: temporal + a-trous chain (+ upsample); hr_shadows_ray_trace + hr_shadows_denoise == hr_shadows_render.
 * In tolerance mode this (like render) launches a-trous iterations 0 and 1 as ONE kernel; the per-iteration entry point below stays. */
hr_status hr_shadows_denoise(hr_shadows* p, const hr_frame_inputs* in, const hr_shadows_params* params, void* stream);
hr_status hr_shadows_temporal(hr_shadows* p, const hr_frame_inputs* in, const hr_shadows_params* params, void* stream);
hr_status hr_shadows_atrous_iteration(hr_shadows* p, const hr_frame_inputs* in, const hr_shadows_params* params, int32_t iteration, void* stream);
hr_status hr_shadows_upsample(hr_shadows* p, const hr_frame_inputs* in, const hr_shadows_params* params, void* stream);
/* Views of intermediates for halo exchange / golden taps: 0 mask, 1 temporal out, 2/3 moments[0/1],
 * 4 prev (feedback) image, 5/6 à-trous ping/pong, 7 upsample, 8 tile classes (uint8 as R8 in an R32 view is not
 * representable: width/height are in tiles, format HR_FORMAT_R32_UINT is NOT implied — 1 byte per tile).
 * STALE IMAGES: what render() / hr_shadows_denoise launch in tolerance mode (radius 1) fuses a-trous iterations 0 and 1, so the image
 * iteration 0 would have written (6) holds an older frame — read the pass's result through hr_shadows_output, or run the iterations one by
 * one (hr_shadows_atrous_iteration) when every intermediate is wanted. */
hr_status hr_shadows_image(hr_shadows* p, int32_t which, hr_image_view* view);
/* Row bands: did a history tap of the frames rendered since the last call fall on an image row this GPU does not hold (per-frame
 * motion beyond hr_band.history_halo)?  Such taps read as disoccluded: the band stays a valid image but stops being identical to the
 * single-GPU one; widen history_halo when this fires.  Synchronises the pass's stream; clears the flag. */
hr_status hr_shadows_history_apron_exceeded(hr_shadows* p, int32_t* exceeded);
hr_status hr_shadows_set_profiling(hr_shadows* p, int32_t enable);
hr_status hr_shadows_get_stage_times(hr_shadows* p, hr_stage_times* out); /* synchronises the recorded events */
/* rays fired by the last ray_trace (lit, non-sky pixels); synchronises the stream it ran on */
hr_status hr_shadows_ray_count(hr_shadows* p, uint64_t* rays);
/* the same per 8x8 tile: out = host array [tiles_y][tiles_x] (nullable: only the extent is returned) — the cost signal
 * the multi-GPU driver balances its row bands with (tiling.balanced_bounds) */
hr_status hr_shadows_tile_ray_counts(hr_shadows* p, uint16_t* out, int32_t* tiles_x, int32_t* tiles_y);
/* Runs the instrumented build of the trace kernel on the same inputs (same masks are produced) and
 * returns out3 = { rays fired, BVH nodes visited, triangles tested } — the terms of the trace pass's
 * algorithmic-bytes figure.  Synchronises the stream. */
hr_status hr_shadows_trace_stats(hr_shadows* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_shadows_params* params, uint64_t* out3, void* stream);
/* hr_shadows_trace_stats counts the full WALK: it bypasses the occluder cache (the triangle that shadowed a pixel last frame is tested before
 * the walk), whose contents depend on the previous frames.  This variant leaves the cache ON: the counts are those of the kernel a render()
 * of `in` launches in the pass's present state (bench.py divides THESE by the timed kernel's duration).  It advances the cache exactly as
 * that trace would; masks and every other output are the same either way. */
/* The launch order of the trace kernel as its NEXT launch will read it (csrc/tile_order.h: launch slot -> 8x8 tile, last frame's heaviest tiles
 * first; the identity list from creation until the first sort has run).  out = host array of *n_tiles words (nullable: only the count is
 * returned; 0 when the launch order is switched off).  Always a permutation of 0 .. n_tiles - 1.  Synchronises the stream of the last render.
 * Introspection for tests and tools (tests/test_gpu_tile_order.py: a hipGraph captured on the FIRST frame replays with the order too). */
hr_status hr_shadows_launch_order(hr_shadows* p, uint32_t* out, int32_t* n_tiles);
hr_status hr_shadows_trace_stats_timed(hr_shadows* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_shadows_params* params, uint64_t* out3, void* stream);
/* After hr_shadows_trace_stats: sum over waves of the slowest lane's (node + triangle) steps.  SIMD lane utilisation of
 * the traversal loop = (nodes + triangles) / (64 * wave_max_steps). */
hr_status hr_shadows_trace_divergence(hr_shadows* p, uint64_t* wave_max_steps);

/* ---- RayTracedAO (src/ray_traced_ao.h) ------------------------------------------------------------ */
typedef struct hr_ao hr_ao;

/* member defaults: ray_traced_ao.h:53-54,72,92,103 */
typedef struct
{
    int32_t denoise;     /* m_denoise = true                    */
    float   ray_length;  /* RayTrace::ray_length = 7.0          */
    float   bias;        /* RayTrace::bias = 0.3                */
    float   alpha;       /* TemporalAccumulation::alpha = 0.01  */
    int32_t blur_radius; /* BilateralBlur::blur_radius = 4      */
    float   power;       /* Upsample::power = 1.2               */
    int32_t spp;         /* EXTENSION (reference = 1): samples per pixel, 1..4 (BASELINE.json configs[2]) */
    int32_t exact;       /* 1 (default) = bit-for-bit parity arithmetic, 0 = tolerance mode (see hr_shadows_params.exact) */
} hr_ao_params;

void      hr_ao_default_params(hr_ao_params* p);
hr_status hr_ao_create(hr_ctx* ctx, int32_t full_width, int32_t full_height, hr_scale scale, const hr_band* band, hr_ao** out);
/* RayTracedAO::render (ray_traced_ao.cpp:98-112) */
hr_status hr_ao_render(hr_ao* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_ao_params* params, void* stream);
/* RayTracedAO::output_ds (ray_traced_ao.cpp:128-148); HR_OUTPUT_ATROUS = OUTPUT_BILATERAL_BLUR */
hr_status hr_ao_output(hr_ao* p, hr_output_kind kind, hr_image_view* view);
hr_status hr_ao_reset_history(hr_ao* p);
hr_status hr_ao_destroy(hr_ao* p);
/* stage-level entry points: ray_trace (:863-903), temporal_accumulation (:983-1028),
 * bilateral_blur pass 0 = direction (1,0), pass 1 = (0,1) (:1032-1137), upsample (:918-955) */
hr_status hr_ao_ray_trace(hr_ao* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_ao_params* params, void* stream);
/* temporal + blur X + blur Y (+ upsample); in tolerance mode (radius 4) the two blur passes are ONE kernel and IMG 5 (blur X) is not written */
hr_status hr_ao_denoise(hr_ao* p, const hr_frame_inputs* in, const hr_ao_params* params, void* stream);
hr_status hr_ao_temporal(hr_ao* p, const hr_frame_inputs* in, const hr_ao_params* params, void* stream);
hr_status hr_ao_blur(hr_ao* p, const hr_frame_inputs* in, const hr_ao_params* params, int32_t pass, void* stream);
hr_status hr_ao_upsample(hr_ao* p, const hr_frame_inputs* in, const hr_ao_params* params, void* stream);
/* 0 mask planes, 1/2 AO[0/1], 3/4 history length[0/1], 5/6 blur[0/1], 7 upsample, 8 tile classes (1 byte per tile).
 * STALE IMAGE: in tolerance mode (blur radius 4) render() / hr_ao_denoise blur X and Y in one kernel and image 5 (blur X) is not written. */
hr_status hr_ao_image(hr_ao* p, int32_t which, hr_image_view* view);
hr_status hr_ao_history_apron_exceeded(hr_ao* p, int32_t* exceeded);   /* see hr_shadows_history_apron_exceeded */
hr_status hr_ao_set_profiling(hr_ao* p, int32_t enable);
hr_status hr_ao_get_stage_times(hr_ao* p, hr_stage_times* out);
hr_status hr_ao_ray_count(hr_ao* p, uint64_t* rays);
hr_status hr_ao_trace_stats(hr_ao* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_ao_params* params, uint64_t* out3, void* stream);
/* as hr_shadows_launch_order */
hr_status hr_ao_launch_order(hr_ao* p, uint32_t* out, int32_t* n_tiles);

/* ---- environment inputs (replace CommonResources::current_skybox_ds / IBL images) ------------------- */
/* Cubemaps are [6][size][size] RGBA16F, faces +X -X +Y -Y +Z -Z, fetched NEAREST.
 * prefiltered: `prefiltered_levels` mips of the specular-prefiltered environment, level l has size
 * prefiltered_size >> l and starts right after level l-1.  brdf_lut: [lut_size][lut_size] RG16F. */
typedef struct
{
    const void* sky;
    int32_t     sky_size;
    const void* prefiltered;
    int32_t     prefiltered_size, prefiltered_levels;
    const void* brdf_lut;
    int32_t     brdf_lut_size;
} hr_environment;

/* ---- DDGI (src/ddgi.h) ---------------------------------------------------------------------------- */
typedef struct hr_ddgi hr_ddgi;

/* DDGIUniforms, scalar layout, 88 bytes (ddgi.cpp:14-32 == shaders/gi/gi_common.glsl:10-28) */
typedef struct
{
    float   grid_start_position[3];
    float   grid_step[3];
    int32_t probe_counts[3];
    float   max_distance, depth_sharpness, hysteresis, normal_bias, energy_preservation;
    int32_t irradiance_probe_side_length, irradiance_texture_width, irradiance_texture_height;
    int32_t depth_probe_side_length, depth_texture_width, depth_texture_height;
    int32_t rays_per_probe, visibility_test;
} hr_ddgi_uniforms;

/* member defaults: ddgi.h:54-56,102 */
typedef struct
{
    int32_t infinite_bounces;          /* RayTrace::infinite_bounces = true            */
    float   infinite_bounce_intensity; /* RayTrace::infinite_bounce_intensity = 1.7    */
    float   gi_intensity;              /* SampleProbeGrid::gi_intensity = 1.0          */
    float   random_orientation[9];     /* column-major 3x3 probe-ray rotation of this frame: the reference draws
                                          it from std::mt19937 seeded by std::random_device (ddgi.cpp:73,788);
                                          here the caller supplies it so frames are reproducible */
    int32_t exact;                     /* 1 (default) = bit-for-bit parity arithmetic; 0 = tolerance mode for the per-pixel probe-grid
                                          sample (see hr_shadows_params.exact; the probe trace and atlas updates have one mode) */
} hr_ddgi_params;

void      hr_ddgi_default_params(hr_ddgi_params* p);
/* DDGI::initialize_probe_grid (ddgi.cpp:150-169) + the atlas sizing of create_images (:197-201) + the constants update_properties_ubo uploads
 * (:738-763, member defaults ddgi.h:54-56,71-75,92-95): probe_counts = ivec3((max - min) / probe_distance) + 2, grid_start_position = min_extents,
 * grid_step = probe_distance, max_distance = 1.5 * probe_distance, hysteresis 0.98, depth_sharpness 50, normal_bias 0.25, energy preservation
 * 0.85, octahedral sides 8 / 16, visibility test on.  Host only (no device).  probe_distance > 0, rays_per_probe > 0 (the reference: 256). */
hr_status hr_ddgi_grid_from_extents(const float min_extents[3], const float max_extents[3], float probe_distance, int32_t rays_per_probe, hr_ddgi_uniforms* out);
/* DDGI(backend, common, g_buffer, scale) + initialize_probe_grid/recreate_probe_grid_resources
 * (ddgi.cpp:61-76,150-237): the grid description is passed in (probe counts, atlas sizes). */
hr_status hr_ddgi_create(hr_ctx* ctx, int32_t full_width, int32_t full_height, hr_scale scale, const hr_ddgi_uniforms* grid, hr_ddgi** out);
/* DDGI::render (ddgi.cpp:89-104): ray_trace -> probe_update (irradiance, depth, borders) -> sample_probe_grid */
hr_status hr_ddgi_render(hr_ddgi* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_environment* env, const hr_ddgi_params* params, void* stream);
/* DDGI::output_ds (ddgi.cpp:128-131): the per-pixel sampled irradiance, RGBA16F */
hr_status hr_ddgi_output(hr_ddgi* p, hr_image_view* view);
/* DDGI::current_read_ds (ddgi.cpp:135-138): irradiance + depth atlases written by the last render() */
hr_status hr_ddgi_current_read(hr_ddgi* p, hr_image_view* irradiance, hr_image_view* depth);
/* Multi-GPU sharding (the reference is single-GPU): this instance traces and updates only the probes
 * of grid z-slabs [probe_z0, probe_z1) — their atlas rows [1 + z0*(side+2), 1 + z1*(side+2)) are contiguous
 * (ddgi.cpp:197-201) — and samples image rows [row_y0, row_y1) (row_y0 a multiple of 8).  The caller all-gathers the
 * slab rows of hr_ddgi_current_write() between hr_ddgi_probe_update and hr_ddgi_sample_probe_grid. */
hr_status hr_ddgi_set_shard(hr_ddgi* p, int32_t probe_z0, int32_t probe_z1, int32_t row_y0, int32_t row_y1);
hr_status hr_ddgi_current_write(hr_ddgi* p, hr_image_view* irradiance, hr_image_view* depth);
/* DDGI::set_normal_bias (ddgi.h:29): read by the next render's uniform upload (ddgi.cpp:747); hr_ddgi_get_uniforms reads it back */
hr_status hr_ddgi_set_normal_bias(hr_ddgi* p, float normal_bias);
/* DDGI::restart_accumulation (ddgi.h:33) */
hr_status hr_ddgi_restart_accumulation(hr_ddgi* p);
hr_status hr_ddgi_destroy(hr_ddgi* p);
/* stage-level entry points (ddgi.cpp:767-986); probe range [probe0, probe1) lets a multi-GPU driver
 * split G1-G4 by z-slab and all-gather the atlas rows */
hr_status hr_ddgi_ray_trace(hr_ddgi* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_environment* env, const hr_ddgi_params* params, void* stream);
/* Instrumented ray trace (the counter build of the same kernel; same rays and results): out3 = rays traced (probe rays + the light / sky
 * rays of the hit points), BVH node steps, triangle tests — the BVH term of the trace pass's algorithmic bytes.  Synchronises. */
hr_status hr_ddgi_trace_stats(hr_ddgi* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_environment* env, const hr_ddgi_params* params, uint64_t* out3, void* stream);
hr_status hr_ddgi_probe_update(hr_ddgi* p, void* stream);
hr_status hr_ddgi_sample_probe_grid(hr_ddgi* p, const hr_frame_inputs* in, const hr_ddgi_params* params, void* stream);
hr_status hr_ddgi_end_frame(hr_ddgi* p); /* m_first_frame = false; m_ping_pong = !m_ping_pong (ddgi.cpp:101-103) */
/* 0 radiance, 1 direction+distance ([probes][rays] RGBA16F), 2/3 irradiance atlas[0/1], 4/5 depth atlas[0/1], 6 sample image */
hr_status hr_ddgi_image(hr_ddgi* p, int32_t which, hr_image_view* view);
hr_status hr_ddgi_get_uniforms(hr_ddgi* p, hr_ddgi_uniforms* out);
hr_status hr_ddgi_set_profiling(hr_ddgi* p, int32_t enable);
hr_status hr_ddgi_get_stage_times(hr_ddgi* p, hr_stage_times* out);
hr_status hr_ddgi_ray_count(hr_ddgi* p, uint64_t* rays);

/* ---- RayTracedReflections (src/ray_traced_reflections.h) ---------------------------------------------- */
typedef struct hr_reflections hr_reflections;

/* member defaults: ray_traced_reflections.h:53-59,77-79,110-116 */
typedef struct
{
    int32_t denoise;                         /* m_denoise = true                                  */
    int32_t sample_gi;                       /* RayTrace::sample_gi = true                        */
    int32_t approximate_with_ddgi;           /* RayTrace::approximate_with_ddgi = true            */
    float   gi_intensity;                    /* 0.5                                               */
    float   rough_ddgi_intensity;            /* 0.5                                               */
    float   ibl_indirect_specular_intensity; /* 0.05                                              */
    float   bias;                            /* 0.5                                               */
    float   trim;                            /* 0.8                                               */
    float   alpha;                           /* TemporalAccumulation::alpha = 0.01                */
    float   moments_alpha;                   /* 0.2                                               */
    int32_t blur_as_input;                   /* false                                             */
    float   phi_color;                       /* ATrous::phi_color = 10                            */
    float   phi_normal;                      /* 32                                                */
    float   sigma_depth;                     /* 1                                                 */
    int32_t radius;                          /* 1                                                 */
    int32_t filter_iterations;               /* 4                                                 */
    int32_t feedback_iteration;              /* 1                                                 */
    float   camera_delta[3];                 /* CommonResources::camera_delta (main.cpp:1077-1079) */
    float   frame_time;                      /* CommonResources::frame_time (pushed, unused by the shader) */
    int32_t exact;                           /* 1 (default) = bit-for-bit parity arithmetic, 0 = tolerance mode (see hr_shadows_params.exact); in tolerance mode the
                                                ray-trace stage keeps rays, hits, ray counts and the ray-length channel bit-exact and computes the hit shading's DDGI
                                                irradiance gathers with the fast arithmetic: the trace image's colour is within the image rule */
} hr_reflections_params;

void      hr_reflections_default_params(hr_reflections_params* p);
hr_status hr_reflections_create(hr_ctx* ctx, int32_t full_width, int32_t full_height, hr_scale scale, const hr_band* band, hr_reflections** out);
/* RayTracedReflections::render(cmd_buf, ddgi) (ray_traced_reflections.cpp:107-123); reads ddgi->current_read_ds() */
hr_status hr_reflections_render(hr_reflections* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_environment* env, hr_ddgi* ddgi,
                                const hr_reflections_params* params, void* stream);
/* RayTracedReflections::output_ds (ray_traced_reflections.cpp:149-169) */
hr_status hr_reflections_output(hr_reflections* p, hr_output_kind kind, hr_image_view* view);
hr_status hr_reflections_reset_history(hr_reflections* p);
hr_status hr_reflections_destroy(hr_reflections* p);
/* stage-level entry points: ray_trace (:997-1057), temporal_accumulation (:1087-1139), a_trous_filter iteration (:1143-1256), upsample (:1260-1296) */
hr_status hr_reflections_ray_trace(hr_reflections* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_environment* env, hr_ddgi* ddgi,
                                   const hr_reflections_params* params, void* stream);
/* Instrumented ray trace, as hr_ddgi_trace_stats: out3 = rays (reflection rays + light rays of the hit points), node steps, triangle tests. */
hr_status hr_reflections_trace_stats(hr_reflections* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_environment* env, hr_ddgi* ddgi,
                                     const hr_reflections_params* params, uint64_t* out3, void* stream);
/* temporal + a-trous chain (+ upsample); tolerance mode: iterations 0 and 1 as ONE kernel */
hr_status hr_reflections_denoise(hr_reflections* p, const hr_frame_inputs* in, const hr_reflections_params* params, void* stream);
hr_status hr_reflections_temporal(hr_reflections* p, const hr_frame_inputs* in, const hr_reflections_params* params, void* stream);
hr_status hr_reflections_atrous_iteration(hr_reflections* p, const hr_frame_inputs* in, const hr_reflections_params* params, int32_t iteration, void* stream);
hr_status hr_reflections_upsample(hr_reflections* p, const hr_frame_inputs* in, const hr_reflections_params* params, void* stream);
/* 0 ray-trace output, 1/2 temporal colour[0/1], 3/4 moments[0/1], 5 prev (feedback) image, 6/7 a-trous ping/pong, 8 upsample, 9 tile classes.
 * STALE IMAGE: in tolerance mode (radius 1) render() / hr_reflections_denoise fuse a-trous iterations 0 and 1; the image iteration 0 would
 * have written (7) holds an older frame (see hr_shadows_image). */
/* which = 10: the colour history the NEXT frame's temporal stage will read (feedback image with blur_as_input, else this frame's temporal
 * output) — the image a row-tiled host exchanges with its neighbours (hr_reflections_exchange_history does) */
hr_status hr_reflections_image(hr_reflections* p, int32_t which, hr_image_view* view);
hr_status hr_reflections_history_apron_exceeded(hr_reflections* p, int32_t* exceeded);   /* see hr_shadows_history_apron_exceeded */
hr_status hr_reflections_set_profiling(hr_reflections* p, int32_t enable);
hr_status hr_reflections_get_stage_times(hr_reflections* p, hr_stage_times* out);
hr_status hr_reflections_ray_count(hr_reflections* p, uint64_t* rays);

/* ---- the frame (src/main.cpp:80-83) ------------------------------------------------------------------------------- */
/* The reference records shadows, AO, DDGI and reflections into ONE command buffer with per-resource barriers, so the GPU overlaps the
 * independent chains.  hr_hybrid_frame gives a HIP host the same: the four render() calls of a frame enqueued as the dependency graph
 * they form (shadows | AO | DDGI probe trace + updates -> reflections | DDGI per-pixel sample), every output bit-identical to the serial
 * order.  The passes are NOT owned; call order at the call site is the reference's, mode picks how the launches reach the GPU. */
typedef struct hr_hybrid_frame hr_hybrid_frame;
typedef enum
{
    HR_FRAME_SERIAL  = 0, /* one stream, the reference's order (= calling the four render() yourself) */
    HR_FRAME_STREAMS = 1, /* fork / join over three internal streams + `stream` */
    HR_FRAME_GRAPH   = 2  /* the forked frame captured into one hipGraph per frame; the instantiated graph is updated in place.
                             Stage profiling (hr_*_set_profiling) must be off: timing events cannot be read back from a captured launch */
} hr_frame_mode;
typedef struct
{
    const hr_environment*        environment;          /* DDGI + reflections */
    const hr_frame_inputs*       shadows_inputs;       /* each pass reads the G-buffer level of its own RayTraceScale */
    const hr_shadows_params*     shadows_params;
    const hr_frame_inputs*       ao_inputs;
    const hr_ao_params*          ao_params;
    const hr_frame_inputs*       ddgi_inputs;
    const hr_ddgi_params*        ddgi_params;
    const hr_frame_inputs*       reflections_inputs;
    const hr_reflections_params* reflections_params;
} hr_hybrid_frame_desc;
/* any of the passes may be NULL (reflections need ddgi); they must outlive the frame object */
hr_status hr_hybrid_frame_create(hr_ctx* ctx, hr_shadows* shadows, hr_ao* ao, hr_ddgi* ddgi, hr_reflections* reflections, hr_hybrid_frame** out);
hr_status hr_hybrid_frame_render(hr_hybrid_frame* f, const hr_scene* scene, const hr_hybrid_frame_desc* desc, hr_frame_mode mode, void* stream);
/* Fork / join for a host that enqueues the chains itself (hr::TiledHybridFrame: the row-tiled passes post their neighbour exchanges from
 * inside render()).  fork: side_streams[0..2] (owned by the frame object) wait for everything enqueued on `stream` so far; join: `stream`
 * waits for everything enqueued on them since. */
hr_status hr_hybrid_frame_fork(hr_hybrid_frame* f, void* stream, void** side_streams);
hr_status hr_hybrid_frame_join(hr_hybrid_frame* f, void* stream);
/* HR_FRAME_GRAPH bookkeeping: graphs instantiated (1 in steady state) and in-place updates (one per later frame) */
hr_status hr_hybrid_frame_graph_stats(hr_hybrid_frame* f, int32_t* instantiations, int32_t* updates);
hr_status hr_hybrid_frame_destroy(hr_hybrid_frame* f);

/* ---- DeferredShading composite (src/deferred_shading.h; SURVEY.md §8f "next" row 1) ------------------- */
/* The consumer of the four passes: shaders/deferred.frag:177-205 as a per-pixel kernel.  Inputs are full-resolution views
 * (the passes' OUTPUT_UPSAMPLE outputs).  Like the reference every pixel is shaded; render_skybox then covers the sky texels. */
typedef struct hr_deferred hr_deferred;

typedef struct
{
    int32_t use_ray_traced_shadows;     /* Shading::use_ray_traced_shadows = true     */
    int32_t use_ray_traced_ao;          /* true                                        */
    int32_t use_ray_traced_reflections; /* true                                        */
    int32_t use_ddgi;                   /* true                                        */
    float   irradiance_sh9[9][4];       /* s_IrradianceSH (9x1 texels, rgb used) — dw::CubemapSHProjection output */
    int32_t draw_skybox;                /* 1 (default): render_skybox (deferred_shading.cpp:734-789) — texels the G-buffer left at
                                           depth 1 take hr_environment.sky along the ray through the pixel centre; 0: shading only */
} hr_deferred_params;

void      hr_deferred_default_params(hr_deferred_params* p);
hr_status hr_deferred_create(hr_ctx* ctx, int32_t width, int32_t height, hr_deferred** out);
/* DeferredShading::render(cmd_buf, ao, shadows, reflections, ddgi) -> render_shading (deferred_shading.cpp:715-723):
 * shadow / ao: R16F or RG16F view (channel 0 is read); reflections / gi: RGBA16F; any of them may be NULL when its flag is 0.
 * in->cur_full supplies GB1/GB2/GB3/depth; env supplies the prefiltered cubemap + BRDF LUT. */
hr_status hr_deferred_render(hr_deferred* p, const hr_frame_inputs* in, const hr_environment* env, const hr_image_view* shadow,
                             const hr_image_view* ao, const hr_image_view* reflections, const hr_image_view* gi,
                             const hr_deferred_params* params, void* stream);
/* DeferredShading::output_ds: RGBA16F HDR colour */
hr_status hr_deferred_output(hr_deferred* p, hr_image_view* view);
hr_status hr_deferred_destroy(hr_deferred* p);

/* ---- GroundTruthPathTracer (src/ground_truth_path_tracer.h:7-44) — SURVEY.md §8f row 3 ---------------------- */
typedef struct hr_ground_truth hr_ground_truth;
typedef struct
{
    int32_t max_ray_bounces;      /* PathTrace::max_ray_bounces = 2 (ground_truth_path_tracer.h:30); only read when trace_indirect != 0 (< 32) */
    float   roughness_multiplier; /* CommonResources::roughness_multiplier */
    int32_t trace_indirect;       /* 0 (default) = the reference as shipped: the recursive traceRayEXT of rchit:95-105 is commented out.
                                     1 = that call re-enabled (rchit:67-108 verbatim): a multi-bounce on-device reference */
} hr_ground_truth_params;

void      hr_ground_truth_default_params(hr_ground_truth_params* p);
/* band: optional rows [band_y0, band_y1) of the image (pixels are independent: no halo, no exchange) */
hr_status hr_ground_truth_create(hr_ctx* ctx, int32_t width, int32_t height, const hr_band* band, hr_ground_truth** out);
/* GroundTruthPathTracer::render (ground_truth_path_tracer.cpp:44-111): one jittered primary sample per pixel, direct
 * light + sky light at the first hit, running mean over the frames since restart_accumulation(). */
hr_status hr_ground_truth_render(hr_ground_truth* p, const hr_scene* scene, const hr_ubo* ubo, const hr_environment* env,
                                 const hr_ground_truth_params* params, void* stream);
/* GroundTruthPathTracer::output_ds (:122-125): RGBA16F running mean */
hr_status hr_ground_truth_output(hr_ground_truth* p, hr_image_view* view);
hr_status hr_ground_truth_restart_accumulation(hr_ground_truth* p); /* ground_truth_path_tracer.h:18 */
hr_status hr_ground_truth_ray_count(hr_ground_truth* p, uint64_t* rays);
hr_status hr_ground_truth_set_profiling(hr_ground_truth* p, int32_t enable);
hr_status hr_ground_truth_get_stage_times(hr_ground_truth* p, hr_stage_times* out);
hr_status hr_ground_truth_destroy(hr_ground_truth* p);

/* ---- TemporalAA (src/temporal_aa.h:17-62) — SURVEY.md §8f row 4 ------------------------------------------------ */
typedef struct hr_taa hr_taa;
typedef struct
{
    int32_t enabled;      /* m_enabled = true */
    int32_t sharpen;      /* m_sharpen = true */
    int32_t reset;        /* m_reset = true and never cleared upstream (temporal_aa.cpp:112,184): history re-seeded every frame; 0 = keep history */
    float   feedback_min; /* 0.88 */
    float   feedback_max; /* 0.97 */
} hr_taa_params;

void      hr_taa_default_params(hr_taa_params* p);
hr_status hr_taa_create(hr_ctx* ctx, int32_t width, int32_t height, hr_taa** out);
/* TemporalAA::update (temporal_aa.cpp:64-81): advances the Halton(2,3) jitter; writes (current.xy, prev.xy) — the value
 * the application puts into hr_ubo.current_prev_jitter and into its projection matrix (main.cpp:941-957).  Nullable out. */
hr_status hr_taa_update(hr_taa* p, uint32_t num_frames, const hr_taa_params* params, float* current_prev_jitter);
/* TemporalAA::render (:84-172): colour = the image being anti-aliased (DeferredShading::output_ds), g = full-resolution
 * G-buffer level (GB2.zw motion vectors, depth), ping_pong = CommonResources::ping_pong. */
hr_status hr_taa_render(hr_taa* p, const hr_image_view* color, const hr_gbuffer_level* g, int32_t ping_pong, const hr_taa_params* params, void* stream);
/* TemporalAA::output_ds (:196-199) */
hr_status hr_taa_output(hr_taa* p, int32_t ping_pong, hr_image_view* view);
hr_status hr_taa_set_profiling(hr_taa* p, int32_t enable);
hr_status hr_taa_get_stage_times(hr_taa* p, hr_stage_times* out);
hr_status hr_taa_destroy(hr_taa* p);

/* ToneMap::render (src/tone_map.cpp:98-143, shaders/tone_map.frag:50-68): exposure, ACES film curve, pow(1/2.2) over an
 * RGBA16F colour image read through the bilinear sampler at the pixel centres; single_channel = 1 shows .rrr (the
 * shadows / AO visualisations, tone_map.cpp:131).  Stateless.  out_rgba32f (device [h][w][4] float, nullable) receives
 * FS_OUT_Color; out_rgba8 (device [h][w][4] uint8, nullable) its UNORM8 conversion floor(c * 255 + 0.5). */
hr_status hr_tone_map(hr_ctx* ctx, const hr_image_view* color, int32_t single_channel, float exposure, float* out_rgba32f, uint8_t* out_rgba8,
                      void* stream);

/* ---- self test ------------------------------------------------------------------------------------ */
/* Evaluates the device-side arithmetic of the numerical contract on arrays so tests can
 * compare it bit for bit with a CPU replay.  which: 0 sincos(x)->(s,c)  1 exp(x)  2 log(x)  3 pow(x,y)
 * 4 fp32->fp16 bits (as float of the uint16)  5 oct_decode(x,y)->(nx,ny,nz)  6 oct_encode(x,y,z)->(ex,ey).
 * in: device [n][3] floats, out: device [n][3] floats. */
hr_status hr_selftest_math(int32_t which, int64_t n, const float* in, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HR_API_H */
