/* hr_api_stages.h — part of the C ABI of hybrid_rendering_amd (included by hr_api.h; do not include on its own).
 *
 * What hr_api.h leaves out because the reference's classes keep it PRIVATE: the stage-level entry points of the four passes (the
 * methods ray_trace / temporal_accumulation / a_trous_filter / upsample ... that render() calls), their intermediate images, profiling
 * and introspection — what a multi-GPU driver (halo exchange between stages: include/hr/tiled.hpp), the tests and the tools need — and
 * the frame object that enqueues the four passes as the dependency graph they form. */
#ifndef HR_API_STAGES_H
#define HR_API_STAGES_H
#ifndef HR_API_H
#error "include hr_api.h"
#endif
#ifdef __cplusplus
extern "C" {
#endif

/* ---- profiler ranges --------------------------------------------------------------------------------------------- */
/* Profiler ranges under the reference's DW_SCOPED_SAMPLE names ("Ray Traced Shadows" > "Ray Trace", "Temporal Accumulation", "Iteration 0" ...;
 * "Ambient Occlusion", "Ray Traced Reflections", "DDGI" > "Probe Update" / "Sample Probe Grid", ...; stages fused into one launch carry the joined names) around every pass and
 * stage, on the calling host thread.  mode 0: off (default; HR_MARKERS=1|2 in the environment sets the initial mode), 1: roctx — shows up in
 * `rocprofv3 --marker-trace` (librocprofiler-sdk-roctx.so is dlopen'ed on first use; silently nothing when it is absent), 2: an in-process log
 * that hr_markers_log() returns as "+name" / "-" lines (tests).  Process-wide. */
hr_status hr_set_markers(int32_t mode);
int32_t   hr_markers_log(char* out, int32_t capacity);   /* returns the length of the whole log; out may be NULL */

/* ---- scene introspection ---------------------------------------------------------------------------------------- */
/* Introspection (tests, tools): copies the device BVH to the host after synchronising the device — hr_scene_info.node_bytes of 80-byte nodes
 * and .tri_bytes of 48-byte triangle references (layouts: csrc/bvh.h).  Either pointer may be NULL. */
hr_status hr_scene_read_bvh(const hr_scene* scene, void* nodes_out, void* tris_out);

/* ---- RayTracedShadows: stages + introspection (ray_traced_shadows.cpp:972-1255) ------------------------------------ */
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

/* ---- RayTracedAO: stages + introspection --------------------------------------------------------------------------- */
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

/* ---- DDGI: sharding, stages + introspection ------------------------------------------------------------------------ */
/* Multi-GPU sharding (the reference is single-GPU): this instance traces and updates only the probes
 * of grid z-slabs [probe_z0, probe_z1) — their atlas rows [1 + z0*(side+2), 1 + z1*(side+2)) are contiguous
 * (ddgi.cpp:197-201) — and samples image rows [row_y0, row_y1) (row_y0 a multiple of 8).  The caller all-gathers the
 * slab rows of hr_ddgi_current_write() between hr_ddgi_probe_update and hr_ddgi_sample_probe_grid. */
hr_status hr_ddgi_set_shard(hr_ddgi* p, int32_t probe_z0, int32_t probe_z1, int32_t row_y0, int32_t row_y1);
hr_status hr_ddgi_current_write(hr_ddgi* p, hr_image_view* irradiance, hr_image_view* depth);

/* ---- DDGI: stages -------------------------------------------------------------------------------------------------- */
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

/* ---- RayTracedReflections: stages + introspection ------------------------------------------------------------------ */
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

#ifdef __cplusplus
}
#endif
#endif /* HR_API_STAGES_H */
