// C++ host shims over the C ABI (hr_api.h) that keep the reference's class / method shapes, so the
// application code of diharaw/hybrid-rendering (src/main.cpp:80-83, src/deferred_shading.cpp:715-723)
// can call the MI355X back end with the same call sites:
//
//   reference                                                     this header
//   ------------------------------------------------------------  ---------------------------------------------
//   RayTracedShadows(backend, common, g_buffer, scale)            hr::RayTracedShadows(ctx, common, g_buffer, scale)
//   void render(dw::vk::CommandBuffer::Ptr cmd_buf)               void render(hr::Stream cmd_buf)          ray_traced_shadows.h:26
//   void render(cmd_buf, DDGI* ddgi)                              void render(hr::Stream cmd_buf, DDGI*)   ray_traced_reflections.h:27
//   dw::vk::DescriptorSet::Ptr output_ds()                        hr::ImageView output_ds()
//   width()/height()/scale()/current_output()/set_current_output  same names
//   DDGI::current_read_ds(), restart_accumulation(), setters      same names
//
// dw::vk::CommandBuffer::Ptr -> hr::Stream (a hipStream_t), dw::vk::DescriptorSet::Ptr -> hr::ImageView (device pointer + extent +
// format).  As in the reference (ray_traced_shadows.h:127-129) a pass keeps NON-OWNING pointers to the application's
// hr::CommonResources (per-frame UBO, frame counters, blue-noise tables, scene, environment) and hr::GBuffer (current / history mip
// chain), given at construction; the application updates both every frame (main.cpp:123-128, :951-966) and then calls render(cmd_buf)
// in the order of main.cpp:80-83 — the call sites do not change.  A second form, render(cmd_buf, const hr::Frame&), takes the inputs
// explicitly (tests, tools).  render() returns void like the reference; failures throw hr::Error on THIS side of the ABI only.
#pragma once
#include "../hr_api.h"
#include <stdexcept>
#include <string>

namespace hr {

struct Error : std::runtime_error
{
    hr_status status;
    Error(hr_status s, const char* what) : std::runtime_error(std::string(what) + ": " + hr_status_string(s) + " — " + hr_last_error()), status(s) {}
};
inline void check(hr_status s, const char* what)
{
    if (s != HR_OK) throw Error(s, what);
}

using Stream    = void*;         // hipStream_t
using ImageView = hr_image_view; // replaces dw::vk::DescriptorSet::Ptr of an output

enum RayTraceScale { RAY_TRACE_SCALE_FULL_RES = HR_SCALE_FULL_RES, RAY_TRACE_SCALE_HALF_RES = HR_SCALE_HALF_RES, RAY_TRACE_SCALE_QUARTER_RES = HR_SCALE_QUARTER_RES };

class Context
{
public:
    explicit Context(int device = 0)
    {
        // the parameter structs carry no size field: a host compiled against another header revision must not call in
        if (hr_api_revision() != HR_API_REVISION) throw std::runtime_error("hr::Context: libhybrid_rendering_amd was built from another revision of hr_api.h (" + std::to_string(hr_api_revision()) + " vs " + std::to_string(HR_API_REVISION) + ")");
        check(hr_ctx_create(device, &m_ctx), "hr_ctx_create");
    }
    ~Context() { hr_ctx_destroy(m_ctx); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    hr_ctx* handle() const { return m_ctx; }
private:
    hr_ctx* m_ctx = nullptr;
};

// dw::RayTracedScene
class Scene
{
public:
    Scene(Context& ctx, const hr_scene_desc& desc) { check(hr_scene_create(ctx.handle(), &desc, &m_scene), "hr_scene_create"); }
    // meshes + instances, as the reference's scene holds them (scene_descriptor_set.glsl:30-34); update_instances() every frame an instance
    // moved replaces main.cpp:74 build_tlas(cmd_buf)
    Scene(Context& ctx, const hr_instanced_scene_desc& desc) { check(hr_scene_create_instanced(ctx.handle(), &desc, &m_scene), "hr_scene_create_instanced"); }
    void     update_instances(const float* model_matrices, Stream cmd_buf) { check(hr_scene_update_instances(m_scene, model_matrices, cmd_buf), "hr_scene_update_instances"); }
    int      instance_count() const { return hr_scene_instance_count(m_scene); }
    uint64_t id() const { return hr_scene_id(m_scene); }   // dw::Scene::id()
    ~Scene() { hr_scene_destroy(m_scene); }
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;
    hr_scene* handle() const { return m_scene; }
private:
    hr_scene* m_scene = nullptr;
};

// What CommonResources + GBuffer hand to every pass each frame, spelled out (the explicit-inputs form of render()).
struct Frame
{
    const Scene*          scene = nullptr;
    hr_frame_inputs       inputs {};
    const hr_environment* environment = nullptr; // reflections / DDGI
};

// src/common.h CommonResources — the per-frame state the application owns and every pass reads: the per-frame UBO contents
// (main.cpp:951-966), frame counters toggled by the app (main.cpp:123-128), the blue-noise tables (blue_noise.cpp:5-19), the
// ray-traced scene and the current environment maps.  Passes hold a pointer to it; nothing here is owned by a pass.
struct CommonResources
{
    const Scene*          scene = nullptr;                 // dw::RayTracedScene
    hr_ubo                ubo {};                          // per_frame_ubo
    uint32_t              num_frames = 0;                  // CommonResources::num_frames
    bool                  ping_pong = false;               // CommonResources::ping_pong
    const uint8_t*        sobol = nullptr;                 // device, blue_noise_image_1 / 2
    const uint8_t*        scrambling_ranking = nullptr;
    const hr_environment* environment = nullptr;           // current_skybox + prefiltered chain + BRDF LUT
    float                 z_buffer_params[4] = { 0, 0, 0, 0 }; // main.cpp:253-254
    float                 camera_delta[3] = { 0, 0, 0 };   // main.cpp:1077-1079
    float                 frame_time = 0.0f;
};

// src/g_buffer.h GBuffer — the images every pass samples: mip chain (g_buffer.cpp:240-243) of the current frame (output_ds()) and of
// the previous one (history_ds()); level `scale` is read by a pass created with that RayTraceScale, level 0 by its upsample stage.
struct GBuffer
{
    hr_gbuffer_level current[3] {};   // mip 0 (full resolution), 1 (half), 2 (quarter)
    hr_gbuffer_level history[3] {};
    uint32_t width() const { return (uint32_t)current[0].width; }
    uint32_t height() const { return (uint32_t)current[0].height; }
};

inline Frame make_frame(const CommonResources& c, const GBuffer& g, int scale)
{
    Frame f;
    f.scene = c.scene; f.environment = c.environment;
    f.inputs.cur = g.current[scale]; f.inputs.prev = g.history[scale].depth ? g.history[scale] : g.current[scale]; f.inputs.cur_full = g.current[0];
    f.inputs.ubo = c.ubo; f.inputs.num_frames = c.num_frames; f.inputs.ping_pong = c.ping_pong ? 1 : 0;
    f.inputs.sobol = c.sobol; f.inputs.scrambling_ranking = c.scrambling_ranking;
    for (int i = 0; i < 4; i++) f.inputs.z_buffer_params[i] = c.z_buffer_params[i];
    return f;
}

class RayTracedShadows
{
public:
    enum OutputType { OUTPUT_RAY_TRACE, OUTPUT_TEMPORAL_ACCUMULATION, OUTPUT_ATROUS, OUTPUT_UPSAMPLE }; // ray_traced_shadows.h:10-16

    RayTracedShadows(Context& ctx, uint32_t width, uint32_t height, RayTraceScale scale = RAY_TRACE_SCALE_FULL_RES, const hr_band* band = nullptr) :
        m_scale(scale), m_width(width >> scale), m_height(height >> scale)
    {
        hr_shadows_default_params(&params);
        check(hr_shadows_create(ctx.handle(), (int32_t)width, (int32_t)height, (hr_scale)scale, band, &m_pass), "hr_shadows_create");
    }
    // the reference's constructor shape: RayTracedShadows(backend, common_resources, g_buffer, scale) — non-owning pointers, read at render()
    RayTracedShadows(Context& ctx, CommonResources* common_resources, GBuffer* g_buffer, RayTraceScale scale = RAY_TRACE_SCALE_FULL_RES, const hr_band* band = nullptr) :
        RayTracedShadows(ctx, g_buffer->width(), g_buffer->height(), scale, band)
    {
        m_common_resources = common_resources; m_g_buffer = g_buffer;
    }
    ~RayTracedShadows() { hr_shadows_destroy(m_pass); }
    RayTracedShadows(const RayTracedShadows&) = delete;
    RayTracedShadows& operator=(const RayTracedShadows&) = delete;

    void render(Stream cmd_buf) { render(cmd_buf, make_frame(*providers(), *m_g_buffer, (int)m_scale)); }   // ray_traced_shadows.h:26
    void render(Stream cmd_buf, const Frame& frame) { check(hr_shadows_render(m_pass, frame.scene->handle(), &frame.inputs, &params, cmd_buf), "RayTracedShadows::render"); }
    ImageView output_ds()
    {
        ImageView v;
        check(hr_shadows_output(m_pass, (hr_output_kind)m_current_output, &v), "RayTracedShadows::output_ds");
        return v;
    }
    uint32_t      width() const { return m_width; }
    uint32_t      height() const { return m_height; }
    RayTraceScale scale() const { return m_scale; }
    OutputType    current_output() const { return m_current_output; }
    void          set_current_output(OutputType o) { m_current_output = o; }
    void          reset_history() { check(hr_shadows_reset_history(m_pass), "reset_history"); }
    hr_shadows*   handle() const { return m_pass; }

    hr_shadows_params params; // the members the reference exposes through gui() (ray_traced_shadows.cpp:120-131)

private:
    const CommonResources* providers() const
    {
        if (!m_common_resources || !m_g_buffer) throw Error(HR_ERR_INVALID_ARG, "RayTracedShadows::render(cmd_buf): constructed without CommonResources / GBuffer");
        return m_common_resources;
    }
    CommonResources* m_common_resources = nullptr;   // non-owning, like the reference's raw pointers
    GBuffer*         m_g_buffer = nullptr;
    hr_shadows*   m_pass = nullptr;
    RayTraceScale m_scale;
    OutputType    m_current_output = OUTPUT_UPSAMPLE;
    uint32_t      m_width, m_height;
};

class RayTracedAO
{
public:
    enum OutputType { OUTPUT_RAY_TRACE, OUTPUT_TEMPORAL_ACCUMULATION, OUTPUT_BILATERAL_BLUR, OUTPUT_UPSAMPLE }; // ray_traced_ao.h:10-16

    RayTracedAO(Context& ctx, uint32_t width, uint32_t height, RayTraceScale scale = RAY_TRACE_SCALE_HALF_RES, const hr_band* band = nullptr) :
        m_scale(scale), m_width(width >> scale), m_height(height >> scale)
    {
        hr_ao_default_params(&params);
        check(hr_ao_create(ctx.handle(), (int32_t)width, (int32_t)height, (hr_scale)scale, band, &m_pass), "hr_ao_create");
    }
    // the reference's constructor shape: RayTracedAO(backend, common_resources, g_buffer, scale) — non-owning pointers, read at render()
    RayTracedAO(Context& ctx, CommonResources* common_resources, GBuffer* g_buffer, RayTraceScale scale = RAY_TRACE_SCALE_HALF_RES, const hr_band* band = nullptr) :
        RayTracedAO(ctx, g_buffer->width(), g_buffer->height(), scale, band)
    {
        m_common_resources = common_resources; m_g_buffer = g_buffer;
    }
    ~RayTracedAO() { hr_ao_destroy(m_pass); }
    RayTracedAO(const RayTracedAO&) = delete;
    RayTracedAO& operator=(const RayTracedAO&) = delete;

    void render(Stream cmd_buf) { render(cmd_buf, make_frame(*providers(), *m_g_buffer, (int)m_scale)); }   // ray_traced_ao.h:26
    void render(Stream cmd_buf, const Frame& frame) { check(hr_ao_render(m_pass, frame.scene->handle(), &frame.inputs, &params, cmd_buf), "RayTracedAO::render"); }
    ImageView output_ds()
    {
        ImageView v;
        check(hr_ao_output(m_pass, (hr_output_kind)m_current_output, &v), "RayTracedAO::output_ds");
        return v;
    }
    uint32_t      width() const { return m_width; }
    uint32_t      height() const { return m_height; }
    RayTraceScale scale() const { return m_scale; }
    OutputType    current_output() const { return m_current_output; }
    void          set_current_output(OutputType o) { m_current_output = o; }
    hr_ao*        handle() const { return m_pass; }

    hr_ao_params params;

private:
    const CommonResources* providers() const
    {
        if (!m_common_resources || !m_g_buffer) throw Error(HR_ERR_INVALID_ARG, "RayTracedAO::render(cmd_buf): constructed without CommonResources / GBuffer");
        return m_common_resources;
    }
    CommonResources* m_common_resources = nullptr;   // non-owning, like the reference's raw pointers
    GBuffer*         m_g_buffer = nullptr;
    hr_ao*        m_pass = nullptr;
    RayTraceScale m_scale;
    OutputType    m_current_output = OUTPUT_UPSAMPLE;
    uint32_t      m_width, m_height;
};

class DDGI
{
public:
    // explicit grid (tests, tools, BASELINE configs[4]'s fixed 16 x 8 x 16): the probe grid never re-derives itself
    DDGI(Context& ctx, uint32_t width, uint32_t height, const hr_ddgi_uniforms& grid, RayTraceScale scale = RAY_TRACE_SCALE_FULL_RES) :
        m_ctx(&ctx), m_scale(scale), m_full_width(width), m_full_height(height), m_width(width >> scale), m_height(height >> scale), m_grid(grid), m_explicit_grid(true)
    {
        hr_ddgi_default_params(&params);
        m_probe_distance = grid.grid_step[0]; m_normal_bias = grid.normal_bias;
        check(hr_ddgi_create(ctx.handle(), (int32_t)width, (int32_t)height, (hr_scale)scale, &grid, &m_pass), "hr_ddgi_create");
    }
    DDGI(Context& ctx, CommonResources* common_resources, GBuffer* g_buffer, const hr_ddgi_uniforms& grid, RayTraceScale scale = RAY_TRACE_SCALE_FULL_RES) :
        DDGI(ctx, g_buffer->width(), g_buffer->height(), grid, scale)
    {
        m_common_resources = common_resources; m_g_buffer = g_buffer;
    }
    // The reference's constructor, argument for argument (ddgi.h:12, ddgi.cpp:61-76): DDGI(backend, common_resources, g_buffer, scale).  The probe
    // grid is derived from the extents of common_resources->scene by the first render() and again whenever the scene's id changes
    // (ddgi.cpp:93-95 -> initialize_probe_grid :150-169 -> recreate_probe_grid_resources :723-734), with the probe distance / normal bias the
    // setters hold at that moment (main.cpp:1094-1136 sets both just before it swaps the scene).
    DDGI(Context& ctx, CommonResources* common_resources, GBuffer* g_buffer, RayTraceScale scale = RAY_TRACE_SCALE_FULL_RES) :
        m_ctx(&ctx), m_common_resources(common_resources), m_g_buffer(g_buffer), m_scale(scale), m_full_width(g_buffer->width()), m_full_height(g_buffer->height()),
        m_width(g_buffer->width() >> scale), m_height(g_buffer->height() >> scale)
    {
        hr_ddgi_default_params(&params);
    }
    ~DDGI() { hr_ddgi_destroy(m_pass); }
    DDGI(const DDGI&) = delete;
    DDGI& operator=(const DDGI&) = delete;

    void render(Stream cmd_buf) { render(cmd_buf, make_frame(*providers(), *m_g_buffer, (int)m_scale)); }   // ddgi.h:15
    // "If the scene has changed re-initialize the probe grid" (ddgi.cpp:93-95); render() calls it, and so does whoever needs handle() before the
    // first render (hr::HybridFrame)
    void prepare(const Scene& scene)
    {
        if (!m_explicit_grid && m_last_scene_id != hr_scene_id(scene.handle())) initialize_probe_grid(scene);
    }
    void render(Stream cmd_buf, const Frame& frame)
    {
        if (frame.scene) prepare(*frame.scene);
        if (!m_pass) throw Error(HR_ERR_INVALID_ARG, "DDGI::render: no scene to derive the probe grid from");
        check(hr_ddgi_render(m_pass, frame.scene->handle(), &frame.inputs, frame.environment, &params, cmd_buf), "DDGI::render");
    }
    ImageView output_ds()
    {
        ImageView v;
        check(hr_ddgi_output(m_pass, &v), "DDGI::output_ds");
        return v;
    }
    // DDGI::current_read_ds(): the irradiance + depth atlases written by the last render()
    void current_read_ds(ImageView& irradiance, ImageView& depth) { check(hr_ddgi_current_read(m_pass, &irradiance, &depth), "DDGI::current_read_ds"); }
    uint32_t      width() const { return m_width; }
    uint32_t      height() const { return m_height; }
    RayTraceScale scale() const { return m_scale; }
    // ddgi.h:24-33
    const int32_t* probe_counts() const { return m_grid.probe_counts; }   // glm::ivec3 in the reference; {0, 0, 0} before the first render()
    float         normal_bias() const { return m_normal_bias; }
    float         probe_distance() const { return m_probe_distance; }
    float         infinite_bounce_intensity() const { return params.infinite_bounce_intensity; }
    float         gi_intensity() const { return params.gi_intensity; }
    void          set_normal_bias(float v)   // takes effect with the next render (update_properties_ubo, ddgi.cpp:747)
    {
        m_normal_bias = v; m_grid.normal_bias = v;
        if (m_pass) check(hr_ddgi_set_normal_bias(m_pass, v), "DDGI::set_normal_bias");
    }
    void          set_probe_distance(float v) { m_probe_distance = v; }   // takes effect when the scene changes, as in the reference
    void          set_infinite_bounce_intensity(float v) { params.infinite_bounce_intensity = v; }
    void          set_gi_intensity(float v) { params.gi_intensity = v; }
    void          set_rays_per_probe(int32_t n) { m_rays_per_probe = n; }  // RayTrace::rays_per_probe = 256 (ddgi.h:56); read when the grid is (re)derived
    void          restart_accumulation() { if (m_pass) check(hr_ddgi_restart_accumulation(m_pass), "DDGI::restart_accumulation"); }
    const hr_ddgi_uniforms& uniforms() const { return m_grid; }   // what update_properties_ubo uploads
    hr_ddgi*      handle() const { return m_pass; }               // NULL until the first render() of a scene-derived grid

    hr_ddgi_params params; // incl. this frame's probe-ray rotation (std::mt19937 in the reference, ddgi.cpp:788)

private:
    const CommonResources* providers() const
    {
        if (!m_common_resources || !m_g_buffer) throw Error(HR_ERR_INVALID_ARG, "DDGI::render(cmd_buf): constructed without CommonResources / GBuffer");
        return m_common_resources;
    }
    void initialize_probe_grid(const Scene& scene)   // ddgi.cpp:150-169 + recreate_probe_grid_resources :723-734 (waits for the device, restarts the accumulation)
    {
        hr_scene_info info;
        check(hr_scene_get_info(scene.handle(), &info), "hr_scene_get_info");
        hr_ddgi_uniforms g;
        check(hr_ddgi_grid_from_extents(info.bounds_lo, info.bounds_hi, m_probe_distance, m_rays_per_probe, &g), "hr_ddgi_grid_from_extents");
        g.normal_bias = m_normal_bias;
        hr_ddgi* fresh = nullptr;
        check(hr_ddgi_create(m_ctx->handle(), (int32_t)m_full_width, (int32_t)m_full_height, (hr_scale)m_scale, &g, &fresh), "hr_ddgi_create");
        hr_ddgi_destroy(m_pass);   // synchronises the device (backend->wait_idle())
        m_pass = fresh; m_grid = g; m_last_scene_id = hr_scene_id(scene.handle());
    }
    Context*         m_ctx = nullptr;
    CommonResources* m_common_resources = nullptr;   // non-owning, like the reference's raw pointers
    GBuffer*         m_g_buffer = nullptr;
    hr_ddgi*      m_pass = nullptr;
    RayTraceScale m_scale;
    uint32_t      m_full_width, m_full_height, m_width, m_height;
    hr_ddgi_uniforms m_grid {};
    bool          m_explicit_grid = false;
    uint64_t      m_last_scene_id = ~0ull;             // m_last_scene_id = UINT32_MAX (ddgi.h:117)
    float         m_probe_distance = 1.0f;             // ProbeGrid::probe_distance = 1.0 (ddgi.h:73)
    float         m_normal_bias = 0.25f;               // ProbeUpdate::normal_bias = 0.25 (ddgi.h:95)
    int32_t       m_rays_per_probe = 256;              // RayTrace::rays_per_probe = 256 (ddgi.h:56)
};

class RayTracedReflections
{
public:
    enum OutputType { OUTPUT_RAY_TRACE, OUTPUT_TEMPORAL_ACCUMULATION, OUTPUT_ATROUS, OUTPUT_UPSAMPLE }; // ray_traced_reflections.h:11-17

    RayTracedReflections(Context& ctx, uint32_t width, uint32_t height, RayTraceScale scale = RAY_TRACE_SCALE_HALF_RES, const hr_band* band = nullptr) :
        m_scale(scale), m_width(width >> scale), m_height(height >> scale)
    {
        hr_reflections_default_params(&params);
        check(hr_reflections_create(ctx.handle(), (int32_t)width, (int32_t)height, (hr_scale)scale, band, &m_pass), "hr_reflections_create");
    }
    // the reference's constructor shape: RayTracedReflections(backend, common_resources, g_buffer, scale) — non-owning pointers, read at render()
    RayTracedReflections(Context& ctx, CommonResources* common_resources, GBuffer* g_buffer, RayTraceScale scale = RAY_TRACE_SCALE_HALF_RES, const hr_band* band = nullptr) :
        RayTracedReflections(ctx, g_buffer->width(), g_buffer->height(), scale, band)
    {
        m_common_resources = common_resources; m_g_buffer = g_buffer;
    }
    ~RayTracedReflections() { hr_reflections_destroy(m_pass); }
    RayTracedReflections(const RayTracedReflections&) = delete;
    RayTracedReflections& operator=(const RayTracedReflections&) = delete;

    // RayTracedReflections::render(cmd_buf, ddgi) — ray_traced_reflections.h:27
    void render(Stream cmd_buf, DDGI* ddgi)
    {
        const CommonResources* c = providers();
        for (int i = 0; i < 3; i++) params.camera_delta[i] = c->camera_delta[i];   // pushed with the temporal pass (ray_traced_reflections.cpp:1124-1135)
        params.frame_time = c->frame_time;
        render(cmd_buf, make_frame(*c, *m_g_buffer, (int)m_scale), ddgi);
    }
    void render(Stream cmd_buf, const Frame& frame, DDGI* ddgi)
    {
        check(hr_reflections_render(m_pass, frame.scene->handle(), &frame.inputs, frame.environment, ddgi->handle(), &params, cmd_buf), "RayTracedReflections::render");
    }
    ImageView output_ds()
    {
        ImageView v;
        check(hr_reflections_output(m_pass, (hr_output_kind)m_current_output, &v), "RayTracedReflections::output_ds");
        return v;
    }
    uint32_t        width() const { return m_width; }
    uint32_t        height() const { return m_height; }
    RayTraceScale   scale() const { return m_scale; }
    OutputType      current_output() const { return m_current_output; }
    void            set_current_output(OutputType o) { m_current_output = o; }
    hr_reflections* handle() const { return m_pass; }

    hr_reflections_params params;

private:
    const CommonResources* providers() const
    {
        if (!m_common_resources || !m_g_buffer) throw Error(HR_ERR_INVALID_ARG, "RayTracedReflections::render(cmd_buf): constructed without CommonResources / GBuffer");
        return m_common_resources;
    }
    CommonResources* m_common_resources = nullptr;   // non-owning, like the reference's raw pointers
    GBuffer*         m_g_buffer = nullptr;
    hr_reflections* m_pass = nullptr;
    RayTraceScale   m_scale;
    OutputType      m_current_output = OUTPUT_UPSAMPLE;
    uint32_t        m_width, m_height;
};

// The frame of main.cpp:80-83 as ONE call: shadows, AO, DDGI, reflections enqueued as the dependency graph they form (hr_api.h
// hr_hybrid_frame: shadows | AO | DDGI probe trace + updates -> reflections | DDGI per-pixel sample).  The reference gets this overlap
// from recording the four render() calls into one Vulkan command buffer; a HIP host gets it here — as forked streams or as one hipGraph
// per frame (captured, then updated in place) — with every pass output bit-identical to the four serial calls.
//
//   hr::HybridFrame frame(ctx, &common, &g_buffer, &shadows, &ao, &ddgi, &reflections);
//   frame.render(cmd_buf);                                   // instead of shadows.render(cmd_buf); ao.render(cmd_buf); ddgi.render(cmd_buf); reflections.render(cmd_buf, &ddgi);
class HybridFrame
{
public:
    enum Mode { SERIAL = HR_FRAME_SERIAL, STREAMS = HR_FRAME_STREAMS, GRAPH = HR_FRAME_GRAPH };

    HybridFrame(Context& ctx, CommonResources* common_resources, GBuffer* g_buffer, RayTracedShadows* shadows, RayTracedAO* ao, DDGI* ddgi, RayTracedReflections* reflections) :
        m_ctx(&ctx), m_common_resources(common_resources), m_g_buffer(g_buffer), m_shadows(shadows), m_ao(ao), m_ddgi(ddgi), m_reflections(reflections)
    {
        if (m_ddgi && m_common_resources->scene) m_ddgi->prepare(*m_common_resources->scene);
        create();
    }
    ~HybridFrame() { hr_hybrid_frame_destroy(m_frame); }
    HybridFrame(const HybridFrame&) = delete;
    HybridFrame& operator=(const HybridFrame&) = delete;

    void render(Stream cmd_buf, Mode mode = STREAMS)
    {
        const CommonResources& c = *m_common_resources;
        if (m_ddgi)   // a scene-derived probe grid re-creates its pass when the scene changes (ddgi.cpp:93-95): the frame object follows
        {
            m_ddgi->prepare(*c.scene);
            if (m_ddgi->handle() != m_ddgi_handle) { hr_hybrid_frame_destroy(m_frame); m_frame = nullptr; create(); }
        }
        Frame fs, fa, fg, fr;
        hr_hybrid_frame_desc d {};
        d.environment = c.environment;
        if (m_shadows) { fs = make_frame(c, *m_g_buffer, (int)m_shadows->scale()); d.shadows_inputs = &fs.inputs; d.shadows_params = &m_shadows->params; }
        if (m_ao) { fa = make_frame(c, *m_g_buffer, (int)m_ao->scale()); d.ao_inputs = &fa.inputs; d.ao_params = &m_ao->params; }
        if (m_ddgi) { fg = make_frame(c, *m_g_buffer, (int)m_ddgi->scale()); d.ddgi_inputs = &fg.inputs; d.ddgi_params = &m_ddgi->params; }
        if (m_reflections)
        {
            for (int i = 0; i < 3; i++) m_reflections->params.camera_delta[i] = c.camera_delta[i];   // as RayTracedReflections::render(cmd_buf, ddgi)
            m_reflections->params.frame_time = c.frame_time;
            fr = make_frame(c, *m_g_buffer, (int)m_reflections->scale()); d.reflections_inputs = &fr.inputs; d.reflections_params = &m_reflections->params;
        }
        check(hr_hybrid_frame_render(m_frame, c.scene->handle(), &d, (hr_frame_mode)mode, cmd_buf), "HybridFrame::render");
    }
    // GRAPH mode: graphs instantiated so far (1 in steady state) and in-place updates
    void graph_stats(int& instantiations, int& updates) { int32_t a = 0, b = 0; check(hr_hybrid_frame_graph_stats(m_frame, &a, &b), "graph_stats"); instantiations = a; updates = b; }
    hr_hybrid_frame* handle() const { return m_frame; }

private:
    void create()
    {
        m_ddgi_handle = m_ddgi ? m_ddgi->handle() : nullptr;
        check(hr_hybrid_frame_create(m_ctx->handle(), m_shadows ? m_shadows->handle() : nullptr, m_ao ? m_ao->handle() : nullptr, m_ddgi_handle,
                                     m_reflections ? m_reflections->handle() : nullptr, &m_frame), "hr_hybrid_frame_create");
    }
    Context*              m_ctx;
    hr_ddgi*              m_ddgi_handle = nullptr;
    CommonResources*      m_common_resources;
    GBuffer*              m_g_buffer;
    RayTracedShadows*     m_shadows;
    RayTracedAO*          m_ao;
    DDGI*                 m_ddgi;
    RayTracedReflections* m_reflections;
    hr_hybrid_frame*      m_frame = nullptr;
};

// src/deferred_shading.h:15-60 — the shading (composite) part
class DeferredShading
{
public:
    DeferredShading(Context& ctx, uint32_t width, uint32_t height)
    {
        hr_deferred_default_params(&params);
        check(hr_deferred_create(ctx.handle(), (int32_t)width, (int32_t)height, &m_pass), "hr_deferred_create");
    }
    DeferredShading(Context& ctx, CommonResources* common_resources, GBuffer* g_buffer) : DeferredShading(ctx, g_buffer->width(), g_buffer->height())
    {
        m_common_resources = common_resources; m_g_buffer = g_buffer;
    }
    ~DeferredShading() { hr_deferred_destroy(m_pass); }
    DeferredShading(const DeferredShading&) = delete;
    DeferredShading& operator=(const DeferredShading&) = delete;

    // DeferredShading::render(cmd_buf, ao, shadows, reflections, ddgi) (deferred_shading.cpp:715-723), literally: the passes' output_ds()
    void render(Stream cmd_buf, RayTracedAO* ao, RayTracedShadows* shadows, RayTracedReflections* reflections, DDGI* ddgi)
    {
        if (!m_common_resources || !m_g_buffer) throw Error(HR_ERR_INVALID_ARG, "DeferredShading::render(cmd_buf, ...): constructed without CommonResources / GBuffer");
        const ImageView a = ao->output_ds(), s = shadows->output_ds(), r = reflections->output_ds(), g = ddgi->output_ds();
        render(cmd_buf, make_frame(*m_common_resources, *m_g_buffer, 0), &a, &s, &r, &g);
    }
    // explicit-inputs form: the pass outputs are the views returned by their output_ds()
    void render(Stream cmd_buf, const Frame& frame, const ImageView* ao, const ImageView* shadows, const ImageView* reflections, const ImageView* gi)
    {
        check(hr_deferred_render(m_pass, &frame.inputs, frame.environment, shadows, ao, reflections, gi, &params, cmd_buf), "DeferredShading::render");
    }
    ImageView output_ds()
    {
        ImageView v;
        check(hr_deferred_output(m_pass, &v), "DeferredShading::output_ds");
        return v;
    }
    hr_deferred* handle() const { return m_pass; }

    hr_deferred_params params;

private:
    CommonResources* m_common_resources = nullptr;
    GBuffer*         m_g_buffer = nullptr;
    hr_deferred* m_pass = nullptr;
};

// src/ground_truth_path_tracer.h:7-44
class GroundTruthPathTracer
{
public:
    GroundTruthPathTracer(Context& ctx, uint32_t width, uint32_t height) : m_width(width), m_height(height)
    {
        hr_ground_truth_default_params(&params);
        check(hr_ground_truth_create(ctx.handle(), (int32_t)width, (int32_t)height, nullptr, &m_pass), "hr_ground_truth_create");
    }
    GroundTruthPathTracer(Context& ctx, CommonResources* common_resources, GBuffer* g_buffer) : GroundTruthPathTracer(ctx, g_buffer->width(), g_buffer->height())
    {
        m_common_resources = common_resources; m_g_buffer = g_buffer;
    }
    ~GroundTruthPathTracer() { hr_ground_truth_destroy(m_pass); }
    GroundTruthPathTracer(const GroundTruthPathTracer&) = delete;
    GroundTruthPathTracer& operator=(const GroundTruthPathTracer&) = delete;

    // the reference reads the camera / light UBO and the sky cubemap from CommonResources (ground_truth_path_tracer.h:16)
    void render(Stream cmd_buf)
    {
        if (!m_common_resources || !m_g_buffer) throw Error(HR_ERR_INVALID_ARG, "GroundTruthPathTracer::render(cmd_buf): constructed without CommonResources / GBuffer");
        render(cmd_buf, make_frame(*m_common_resources, *m_g_buffer, 0));
    }
    void render(Stream cmd_buf, const Frame& frame)
    {
        check(hr_ground_truth_render(m_pass, frame.scene->handle(), &frame.inputs.ubo, frame.environment, &params, cmd_buf), "GroundTruthPathTracer::render");
    }
    ImageView output_ds()
    {
        ImageView v;
        check(hr_ground_truth_output(m_pass, &v), "GroundTruthPathTracer::output_ds");
        return v;
    }
    void             restart_accumulation() { check(hr_ground_truth_restart_accumulation(m_pass), "GroundTruthPathTracer::restart_accumulation"); }
    uint32_t         width() const { return m_width; }
    uint32_t         height() const { return m_height; }
    hr_ground_truth* handle() const { return m_pass; }

    hr_ground_truth_params params;

private:
    CommonResources* m_common_resources = nullptr;
    GBuffer*         m_g_buffer = nullptr;
    hr_ground_truth* m_pass = nullptr;
    uint32_t         m_width, m_height;
};

// src/temporal_aa.h:17-62
class TemporalAA
{
public:
    TemporalAA(Context& ctx, uint32_t width, uint32_t height)
    {
        hr_taa_default_params(&params);
        check(hr_taa_create(ctx.handle(), (int32_t)width, (int32_t)height, &m_pass), "hr_taa_create");
    }
    ~TemporalAA() { hr_taa_destroy(m_pass); }
    TemporalAA(const TemporalAA&) = delete;
    TemporalAA& operator=(const TemporalAA&) = delete;

    // TemporalAA::update(): CommonResources::num_frames selects the Halton sample
    void update(uint32_t num_frames) { check(hr_taa_update(m_pass, num_frames, &params, m_jitter), "TemporalAA::update"); }
    // colour = DeferredShading::output_ds() (or any pass output being visualised); g_buffer = GBuffer::output_ds() level 0
    void render(Stream cmd_buf, const ImageView& color, const hr_gbuffer_level& g_buffer, bool ping_pong)
    {
        m_ping_pong = ping_pong;
        check(hr_taa_render(m_pass, &color, &g_buffer, ping_pong ? 1 : 0, &params, cmd_buf), "TemporalAA::render");
    }
    ImageView output_ds()
    {
        ImageView v;
        check(hr_taa_output(m_pass, m_ping_pong ? 1 : 0, &v), "TemporalAA::output_ds");
        return v;
    }
    bool         enabled() const { return params.enabled != 0; }
    const float* current_jitter() const { return m_jitter; }     // vec2
    const float* prev_jitter() const { return m_jitter + 2; }    // vec2
    hr_taa*      handle() const { return m_pass; }

    hr_taa_params params;

private:
    hr_taa* m_pass = nullptr;
    float   m_jitter[4] = { 0, 0, 0, 0 };
    bool    m_ping_pong = false;
};

// src/tone_map.h:16-40.  render() reads the (TAA) colour image and writes the displayable image: out_rgba8 (device, nullable)
// is what the reference renders into the swap chain, out_rgba32f (device, nullable) the unquantised FS_OUT_Color.
class ToneMap
{
public:
    explicit ToneMap(Context& ctx) : m_ctx(ctx) {}
    void render(Stream cmd_buf, const ImageView& color, uint8_t* out_rgba8, float* out_rgba32f = nullptr, bool single_channel = false)
    {
        check(hr_tone_map(m_ctx.handle(), &color, single_channel ? 1 : 0, exposure, out_rgba32f, out_rgba8, cmd_buf), "ToneMap::render");
    }
    float exposure = 1.0f;   // tone_map.h:38 m_exposure

private:
    Context& m_ctx;
};

} // namespace hr
