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
 *   - the two arithmetic modes (`exact`) and what the tolerance mode promises: docs/TOLERANCE.md.
 * This file is the reference's PUBLIC class surface; hr_api_stages.h (stage-level entry points, introspection, the frame object) and
 * hr_api_post.h (deferred composite, ground truth, TAA) are included at the end.
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
/* Revision of the struct layouts and entry points.  The parameter structs are passed by pointer WITHOUT a size field, so a host built against
 * another revision must not call in: compare hr_api_revision() with HR_API_REVISION once at start-up (hr::Context does).  History: docs/API_HISTORY.md. */
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

/* ---- instanced scenes (scene_descriptor_set.glsl:30-34 Instance { mat4 model_matrix; uint mesh_idx; }, :102-160; main.cpp:74 build_tlas) ----
 * The reference's scene model — meshes + instances, the acceleration structure updated every frame — on ONE world-space 8-wide BVH: every
 * instance owns a private copy of its mesh's subtree under a top level over the instance roots, so the trace kernels keep the single-level
 * walk of hr_scene_create.  hr_scene_update_instances moves instances on the GPU (vertices = model_matrix * (p, 1), one rounding per
 * operation; node boxes refitted level by level; only instances whose matrix changed are touched; the top level re-built when it degrades).  Answers equal those of a flattened
 * hr_scene_create over the same world-space vertices; hit records name the global triangle index (instance order, then mesh order); hit
 * shading interpolates the OBJECT-space attributes, then applies interpolated_vertex + transform_vertex's operations.  DESIGN.md section 2. */
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
/* The top level over the instance roots (host SAH over the instances' boxes, 8-wide) is re-built by hr_scene_update_instances on its own when the
 * instances have moved far enough for its boxes to overlap (half-area sum > 1.5x the sum at the last build, and a fresh one at least 10 % better);
 * this forces one (re-places the roots, refits every level once) / counts them.  Answers never depend on it. */
hr_status hr_scene_rebuild_top_level(hr_scene* scene, void* stream);
int32_t   hr_scene_top_level_rebuilds(const hr_scene* scene);
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
    int32_t exact;              /* 1 (default): parity mode — every stage image equals the oracle and the reference's shaders bit for bit.
                                   0: tolerance mode (production; what bench.py times): hardware rcp / rsq / sqrt / exp / log + FMAs in the
                                   denoise kernels; masks, ray counts, DDGI atlases and the reflections' trace image stay bit-exact, every other
                                   fp16 image is within 2 fp16 ulp (or 2^-20) of the oracle on >= 99.9 % of its texels.  THE CONTRACT, clause by clause with
                                   the test behind each: docs/TOLERANCE.md. */
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
/* DDGI::set_normal_bias (ddgi.h:29): read by the next render's uniform upload (ddgi.cpp:747); hr_ddgi_get_uniforms reads it back */
hr_status hr_ddgi_set_normal_bias(hr_ddgi* p, float normal_bias);
/* DDGI::restart_accumulation (ddgi.h:33) */
hr_status hr_ddgi_restart_accumulation(hr_ddgi* p);
hr_status hr_ddgi_destroy(hr_ddgi* p);
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
#ifdef __cplusplus
}
#endif
/* the rest of the ABI: stage-level entry points, introspection and the frame object (multi-GPU drivers, tests, tools) ... */
#include "hr_api_stages.h"
/* ... and the SURVEY 8(f) rows: deferred composite, ground-truth accumulator, TAA, tone map, self test */
#include "hr_api_post.h"
#endif /* HR_API_H */
