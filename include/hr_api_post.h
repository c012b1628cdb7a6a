/* hr_api_post.h — part of the C ABI of hybrid_rendering_amd (included by hr_api.h; do not include on its own).
 *
 * The SURVEY.md 8(f) rows downstream of the hot path — deferred composite, ground-truth accumulator, TAA, tone map — and the self test. */
#ifndef HR_API_POST_H
#define HR_API_POST_H
#ifndef HR_API_H
#error "include hr_api.h"
#endif
#ifdef __cplusplus
extern "C" {
#endif

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
#endif /* HR_API_POST_H */
