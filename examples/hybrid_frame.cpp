// C++ host example: one hybrid frame loop through include/hr/passes.hpp in the order of the reference's render loop
// (src/main.cpp:74-99): shadows, AO, DDGI, reflections, deferred composite, TAA — plus the ground-truth accumulator —
// on a Cornell-style box, three frames with a moving camera.  No Python, no torch: hipMalloc + the C ABI.
//
//   hipcc -std=c++17 -I include examples/hybrid_frame.cpp -L hybrid_rendering_amd -lhybrid_rendering_amd \
//         -Wl,-rpath,$PWD/hybrid_rendering_amd -o /tmp/hybrid_frame && /tmp/hybrid_frame
#include <hr/passes.hpp>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

struct V3 { float x, y, z; };

void quad(std::vector<float>& v, std::vector<uint32_t>& mat, uint32_t m, V3 a, V3 b, V3 c, V3 d)
{
    const V3 t[6] = { a, b, c, a, c, d };
    for (const V3& p : t) { v.push_back(p.x); v.push_back(p.y); v.push_back(p.z); }
    mat.push_back(m); mat.push_back(m);
}

void mul(const float* A, const float* B, float* C)
{
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++)
        {
            float s = 0;
            for (int k = 0; k < 4; k++) s += A[k * 4 + r] * B[c * 4 + k];
            C[c * 4 + r] = s;
        }
}
bool invert(const float* m, float* inv)
{
    double a[4][8];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) { a[r][c] = m[c * 4 + r]; a[r][4 + c] = r == c; }
    for (int i = 0; i < 4; i++)
    {
        int p = i;
        for (int r = i + 1; r < 4; r++) if (std::fabs(a[r][i]) > std::fabs(a[p][i])) p = r;
        if (std::fabs(a[p][i]) < 1e-12) return false;
        for (int c = 0; c < 8; c++) std::swap(a[i][c], a[p][c]);
        const double d = a[i][i];
        for (int c = 0; c < 8; c++) a[i][c] /= d;
        for (int r = 0; r < 4; r++)
            if (r != i) { const double f = a[r][i]; for (int c = 0; c < 8; c++) a[r][c] -= f * a[i][c]; }
    }
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) inv[c * 4 + r] = (float)a[r][4 + c];
    return true;
}

uint16_t f2h(float f) // round to nearest even, finite inputs of moderate size only
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    int32_t  e = (int32_t)((u >> 23) & 0xff) - 127 + 15;
    uint32_t m = u & 0x7fffffu;
    if (e <= 0) return (uint16_t)sign;
    if (e >= 31) return (uint16_t)(sign | 0x7bffu);
    uint32_t h = (uint32_t)(e << 10) | (m >> 13);
    const uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}
float h2f(uint16_t h)
{
    const uint32_t e = (h >> 10) & 31u, m = h & 0x3ffu;
    float v = e == 0 ? std::ldexp((float)m, -24) : std::ldexp((float)(m | 0x400u), (int)e - 25);
    return (h & 0x8000u) ? -v : v;
}

#define HIP_OK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); return 2; } } while (0)

template <typename T>
T* upload(const std::vector<T>& v)
{
    void* d = nullptr;
    if (hipMalloc(&d, v.size() * sizeof(T)) != hipSuccess) return nullptr;
    (void)hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    return (T*)d;
}

// mean of channel `c` (of `nch` half channels) over an image view
double mean_of(const hr::ImageView& v, int nch, int c)
{
    std::vector<uint16_t> h((size_t)v.width * v.height * nch);
    (void)hipMemcpy(h.data(), v.data, h.size() * 2, hipMemcpyDeviceToHost);
    double s = 0;
    for (size_t i = 0; i < (size_t)v.width * v.height; i++) s += h2f(h[i * nch + c]);
    return s / ((double)v.width * v.height);
}

// two views hold the same bytes
bool same_image(const hr::ImageView& a, const hr::ImageView& b)
{
    if (a.width != b.width || a.height != b.height || a.row_pitch_bytes != b.row_pitch_bytes) return false;
    const size_t n = (size_t)a.row_pitch_bytes * a.height;
    std::vector<uint8_t> x(n), y(n);
    (void)hipMemcpy(x.data(), a.data, n, hipMemcpyDeviceToHost);
    (void)hipMemcpy(y.data(), b.data, n, hipMemcpyDeviceToHost);
    return std::memcmp(x.data(), y.data(), n) == 0;
}

} // namespace

int main(int argc, char** argv)
{
    const int W = 320, H = 192;
    // ---- scene: Cornell-style room with two boxes; materials: white, red, green, polished grey ----------------------
    std::vector<float>    v;
    std::vector<uint32_t> mat;
    const float S = 100.0f;
    quad(v, mat, 0, { 0, 0, 0 }, { 0, 0, S }, { S, 0, S }, { S, 0, 0 });
    quad(v, mat, 0, { 0, S, 0 }, { S, S, 0 }, { S, S, S }, { 0, S, S });
    quad(v, mat, 0, { 0, 0, 0 }, { S, 0, 0 }, { S, S, 0 }, { 0, S, 0 });
    quad(v, mat, 1, { 0, 0, 0 }, { 0, S, 0 }, { 0, S, S }, { 0, 0, S });
    quad(v, mat, 2, { S, 0, 0 }, { S, 0, S }, { S, S, S }, { S, S, 0 });
    auto box = [&](uint32_t m, V3 lo, V3 hi) {
        quad(v, mat, m, { lo.x, lo.y, lo.z }, { lo.x, lo.y, hi.z }, { lo.x, hi.y, hi.z }, { lo.x, hi.y, lo.z });
        quad(v, mat, m, { hi.x, lo.y, lo.z }, { hi.x, hi.y, lo.z }, { hi.x, hi.y, hi.z }, { hi.x, lo.y, hi.z });
        quad(v, mat, m, { lo.x, hi.y, lo.z }, { lo.x, hi.y, hi.z }, { hi.x, hi.y, hi.z }, { hi.x, hi.y, lo.z });
        quad(v, mat, m, { lo.x, lo.y, lo.z }, { lo.x, hi.y, lo.z }, { hi.x, hi.y, lo.z }, { hi.x, lo.y, lo.z });
        quad(v, mat, m, { lo.x, lo.y, hi.z }, { hi.x, lo.y, hi.z }, { hi.x, hi.y, hi.z }, { lo.x, hi.y, hi.z });
    };
    box(0, { 15, 0, 15 }, { 45, 60, 45 });
    box(3, { 55, 0, 50 }, { 85, 30, 80 });
    const float materials[4][8] = { { 0.75f, 0.75f, 0.75f, 0, 0.6f, 0, 0, 0 }, { 0.7f, 0.1f, 0.1f, 0, 0.6f, 0, 0, 0 },
                                    { 0.1f, 0.7f, 0.1f, 0, 0.6f, 0, 0, 0 }, { 0.6f, 0.6f, 0.65f, 1.0f, 0.15f, 0, 0, 0 } };
    const int n_tris = (int)v.size() / 9;

    try
    {
        hr::Context ctx(0);
        hr_scene_desc desc {};
        desc.positions = v.data(); desc.n_tris = n_tris; desc.tri_material = mat.data();
        desc.materials = &materials[0][0]; desc.n_materials = 4;
        hr::Scene scene(ctx, desc);

        // ---- environment: a 16^2 sky cubemap (blue above, dark below), its mip chain, a flat BRDF LUT -------------
        const int SK = 16, LV = 4;
        std::vector<uint16_t> sky((size_t)6 * SK * SK * 4), pre, lut((size_t)16 * 16 * 2);
        for (int f = 0; f < 6; f++)
            for (int i = 0; i < SK * SK; i++)
            {
                const bool up = f == 2 || (f != 3 && (i / SK) < SK / 2); // +Y face and the upper half of the side faces
                const float c[3] = { up ? 0.35f : 0.08f, up ? 0.55f : 0.07f, up ? 0.9f : 0.06f };
                for (int k = 0; k < 3; k++) sky[((size_t)f * SK * SK + i) * 4 + k] = f2h(c[k]);
                sky[((size_t)f * SK * SK + i) * 4 + 3] = f2h(1.0f);
            }
        for (int l = 0, s = SK; l < LV; l++, s >>= 1)   // nearest-decimated chain is enough for the example
            for (int f = 0; f < 6; f++)
                for (int y = 0; y < s; y++)
                    for (int x = 0; x < s; x++)
                        for (int k = 0; k < 4; k++) pre.push_back(sky[(((size_t)f * SK + (y << l)) * SK + (x << l)) * 4 + k]);
        for (size_t i = 0; i < lut.size(); i += 2) { lut[i] = f2h(0.9f); lut[i + 1] = f2h(0.05f); }
        hr_environment env {};
        env.sky = upload(sky); env.sky_size = SK;
        env.prefiltered = upload(pre); env.prefiltered_size = SK; env.prefiltered_levels = LV;
        env.brdf_lut = upload(lut); env.brdf_lut_size = 16;

        // ---- DDGI: the probe grid is derived from the scene's extents by the pass itself (DDGI::initialize_probe_grid, ddgi.cpp:150-169):
        // probe distance S / 3 over the S-sized room -> ivec3(3) + 2 = 5 x 5 x 5 probes, the presets' normal bias 1.0 (main.cpp:1094-1095) ----
        auto configure_ddgi = [&](hr::DDGI& d) { d.set_normal_bias(1.0f); d.set_probe_distance(S / 3.0f); d.set_rays_per_probe(64); };

        // ---- passes, created like main.cpp:1150-1159: every pass keeps non-owning pointers to the application's CommonResources
        // and GBuffer (ray_traced_shadows.h:127-129) and reads them at render() --------------------------------------------------
        hr::CommonResources common;
        hr::GBuffer         g_buffer;
        common.scene = &scene; common.environment = &env;
        g_buffer.current[0].width = W; g_buffer.current[0].height = H;   // the extent is read at construction, the images every frame
        hr::RayTracedShadows      shadows(ctx, &common, &g_buffer, hr::RAY_TRACE_SCALE_FULL_RES);
        hr::RayTracedAO           ao(ctx, &common, &g_buffer, hr::RAY_TRACE_SCALE_FULL_RES);
        hr::DDGI                  ddgi(ctx, &common, &g_buffer);   // the reference's four constructor arguments (ddgi.h:12)
        configure_ddgi(ddgi);
        hr::RayTracedReflections  reflections(ctx, &common, &g_buffer, hr::RAY_TRACE_SCALE_FULL_RES);
        hr::DeferredShading       deferred(ctx, &common, &g_buffer);
        hr::TemporalAA            taa(ctx, W, H);
        hr::GroundTruthPathTracer ground_truth(ctx, &common, &g_buffer);
        // hr::HybridFrame: the same four passes enqueued as the dependency graph they form — forked streams, and one hipGraph per frame.
        // Two more sets of passes (each keeps its own temporal history) so that their outputs can be compared with the serial calls.
        hr::RayTracedShadows      shadows_s(ctx, &common, &g_buffer, hr::RAY_TRACE_SCALE_FULL_RES), shadows_g(ctx, &common, &g_buffer, hr::RAY_TRACE_SCALE_FULL_RES);
        hr::RayTracedAO           ao_s(ctx, &common, &g_buffer, hr::RAY_TRACE_SCALE_FULL_RES), ao_g(ctx, &common, &g_buffer, hr::RAY_TRACE_SCALE_FULL_RES);
        hr::DDGI                  ddgi_s(ctx, &common, &g_buffer), ddgi_g(ctx, &common, &g_buffer);
        configure_ddgi(ddgi_s); configure_ddgi(ddgi_g);
        hr::RayTracedReflections  reflections_s(ctx, &common, &g_buffer, hr::RAY_TRACE_SCALE_FULL_RES), reflections_g(ctx, &common, &g_buffer, hr::RAY_TRACE_SCALE_FULL_RES);
        hr::HybridFrame           frame_streams(ctx, &common, &g_buffer, &shadows_s, &ao_s, &ddgi_s, &reflections_s);
        hr::HybridFrame           frame_graph(ctx, &common, &g_buffer, &shadows_g, &ao_g, &ddgi_g, &reflections_g);
        hipStream_t               app_stream = nullptr;
        HIP_OK(hipStreamCreateWithFlags(&app_stream, hipStreamNonBlocking));
        int forked_mismatches = 0;

        // ---- G-buffers (two, ping-pong) + blue-noise tables ---------------------------------------------------------------
        void *gb1[2], *gb2[2], *gb3[2], *depth[2];
        for (int i = 0; i < 2; i++)
        {
            HIP_OK(hipMalloc(&gb1[i], (size_t)W * H * 4)); HIP_OK(hipMalloc(&gb2[i], (size_t)W * H * 8));
            HIP_OK(hipMalloc(&gb3[i], (size_t)W * H * 8)); HIP_OK(hipMalloc(&depth[i], (size_t)W * H * 4));
        }
        std::vector<uint8_t> sob(256 * 4), sr(128 * 128 * 4);
        uint32_t lcg = 12345u;
        for (auto& b : sob) { lcg = lcg * 1664525u + 1013904223u; b = (uint8_t)(lcg >> 24); }
        for (auto& b : sr) { lcg = lcg * 1664525u + 1013904223u; b = (uint8_t)(lcg >> 24); }
        const uint8_t *sob_d = upload(sob), *sr_d = upload(sr);

        const float fy = 1.0f / std::tan(40.0f * 3.14159265f / 360.0f), fx = fy * (float)H / (float)W, n = 1.0f, fa = 1000.0f;
        float prev_vp[16] = { 0 };
        double m_shadow = 0, m_ao = 0, m_gi = 0, m_refl = 0, m_final = 0, m_taa = 0, m_gt = 0;
        for (uint32_t f = 0; f < 3; f++)
        {
            const int pp = (int)(f & 1);
            hr_ubo& u = common.ubo;            // main.cpp:951-966 fills the per-frame UBO of CommonResources
            std::memset(&u, 0, sizeof(u));
            // TemporalAA::update() first: the jitter goes into the projection and the UBO (main.cpp:941-957, :1025)
            taa.update(f);
            const float eye[3] = { 50.0f + 1.5f * f, 50.0f, 235.0f - 2.0f * f };
            float view[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, -eye[0], -eye[1], -eye[2], 1 };
            float proj[16] = { fx, 0, 0, 0, 0, -fy, 0, 0, 0, 0, fa / (n - fa), -1, 0, 0, -(fa * n) / (fa - n), 0 };
            float jit[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, taa.current_jitter()[0], taa.current_jitter()[1], 0, 1 }, jproj[16];
            mul(jit, proj, jproj);
            mul(jproj, view, u.view_proj);
            std::memcpy(u.prev_view_proj, f ? prev_vp : u.view_proj, sizeof(prev_vp));
            std::memcpy(prev_vp, u.view_proj, sizeof(prev_vp));
            if (!invert(u.view_proj, u.view_proj_inverse) || !invert(view, u.view_inverse) || !invert(jproj, u.proj_inverse)) return 3;
            u.cam_pos[0] = eye[0]; u.cam_pos[1] = eye[1]; u.cam_pos[2] = eye[2]; u.cam_pos[3] = 1.0f;
            std::memcpy(u.current_prev_jitter, taa.current_jitter(), 8);
            std::memcpy(u.current_prev_jitter + 2, taa.prev_jitter(), 8);
            u.light.data0[1] = -1.0f; u.light.data0[3] = 9000.0f;
            u.light.data1[0] = 50; u.light.data1[1] = 92; u.light.data1[2] = 50; u.light.data1[3] = 3.0f;   // soft point light under the ceiling
            u.light.data2[0] = u.light.data2[1] = u.light.data2[2] = 1.0f;
            u.light.data3[0] = 1.0f;

            hr::check(hr_gbuffer_raycast(scene.handle(), &u, W, H, gb1[pp], gb2[pp], gb3[pp], (float*)depth[pp], nullptr), "hr_gbuffer_raycast");
            hr_gbuffer_level cur { gb1[pp], gb2[pp], gb3[pp], (const float*)depth[pp], W, H };
            hr_gbuffer_level prv = f ? hr_gbuffer_level { gb1[!pp], gb2[!pp], gb3[!pp], (const float*)depth[!pp], W, H } : cur;
            g_buffer.current[0] = cur; g_buffer.history[0] = prv;           // GBuffer::output_ds() / history_ds()
            common.num_frames = f; common.ping_pong = pp != 0;              // main.cpp:123-128
            common.sobol = sob_d; common.scrambling_ranking = sr_d;
            common.z_buffer_params[0] = 1.0f - fa / n; common.z_buffer_params[1] = fa / n;
            common.z_buffer_params[2] = common.z_buffer_params[0] / fa; common.z_buffer_params[3] = common.z_buffer_params[1] / fa;

            // main.cpp:80-99, call for call
            hr::Stream cmd_buf = nullptr;
            shadows.render(cmd_buf);
            ao.render(cmd_buf);
            ddgi.render(cmd_buf);
            reflections.render(cmd_buf, &ddgi);
            deferred.render(cmd_buf, &ao, &shadows, &reflections, &ddgi);
            hr::ImageView s_v = shadows.output_ds(), a_v = ao.output_ds(), r_v = reflections.output_ds(), g_v = ddgi.output_ds();
            hr::ImageView color = deferred.output_ds();
            taa.render(cmd_buf, color, cur, pp != 0);
            ground_truth.render(cmd_buf);
            // the same frame through hr::HybridFrame (one call instead of four): forked streams, then as a hipGraph
            frame_streams.render(app_stream, hr::HybridFrame::STREAMS);
            frame_graph.render(app_stream, hr::HybridFrame::GRAPH);
            HIP_OK(hipDeviceSynchronize());
            const bool eq_s = same_image(shadows_s.output_ds(), shadows.output_ds()) && same_image(ao_s.output_ds(), ao.output_ds()) &&
                              same_image(ddgi_s.output_ds(), ddgi.output_ds()) && same_image(reflections_s.output_ds(), reflections.output_ds());
            const bool eq_g = same_image(shadows_g.output_ds(), shadows.output_ds()) && same_image(ao_g.output_ds(), ao.output_ds()) &&
                              same_image(ddgi_g.output_ds(), ddgi.output_ds()) && same_image(reflections_g.output_ds(), reflections.output_ds());
            std::printf("frame %u: hr::HybridFrame STREAMS %s serial, GRAPH %s serial\n", f, eq_s ? "==" : "DIFFERS FROM", eq_g ? "==" : "DIFFERS FROM");
            forked_mismatches += (eq_s ? 0 : 1) + (eq_g ? 0 : 1);
            m_shadow = mean_of(s_v, s_v.format == HR_FORMAT_R16F ? 1 : 2, 0);
            m_ao = mean_of(a_v, 1, 0); m_gi = mean_of(g_v, 4, 1); m_refl = mean_of(r_v, 4, 1);
            m_final = mean_of(color, 4, 1); m_taa = mean_of(taa.output_ds(), 4, 1); m_gt = mean_of(ground_truth.output_ds(), 4, 1);
            std::printf("frame %u: shadow %.4f  ao %.4f  gi %.4f  reflections %.4f  composite %.4f  taa %.4f  ground truth %.4f\n", f, m_shadow, m_ao,
                        m_gi, m_refl, m_final, m_taa, m_gt);
        }
        // ToneMap::render (main.cpp:99): the anti-aliased HDR frame -> the displayable 8-bit image, written as a PPM
        hr::ToneMap tone_map(ctx);
        uint8_t*    ldr = nullptr;
        HIP_OK(hipMalloc(&ldr, (size_t)W * H * 4));
        tone_map.render(nullptr, taa.output_ds(), ldr);
        std::vector<uint8_t> host((size_t)W * H * 4);
        HIP_OK(hipMemcpy(host.data(), ldr, host.size(), hipMemcpyDeviceToHost));
        HIP_OK(hipFree(ldr));
        double m_ldr = 0.0;
        if (FILE* fp = std::fopen(argc > 1 ? argv[1] : "/tmp/hybrid_frame.ppm", "wb"))
        {
            std::fprintf(fp, "P6\n%d %d\n255\n", W, H);
            for (size_t i = 0; i < (size_t)W * H; i++) { std::fwrite(&host[i * 4], 1, 3, fp); m_ldr += host[i * 4] + host[i * 4 + 1] + host[i * 4 + 2]; }
            std::fclose(fp);
        }
        m_ldr /= 3.0 * 255.0 * W * H;
        std::printf("tone-mapped frame: mean %.4f\n", m_ldr);
        int inst = 0, upd = 0;
        frame_graph.graph_stats(inst, upd);
        std::printf("hr::HybridFrame GRAPH: %d graph instantiated, %d in-place updates; %d mismatching frames\n", inst, upd, forked_mismatches);
        std::printf("hr::DDGI(ctx, common, g_buffer): probe grid %d x %d x %d derived from the scene's extents at probe distance %.3f, normal bias %.2f\n",
                    ddgi.probe_counts()[0], ddgi.probe_counts()[1], ddgi.probe_counts()[2], ddgi.probe_distance(), ddgi.normal_bias());
        std::printf("hybrid_frame: %d triangles, %dx%d, all passes ran\n", n_tris, W, H);
        const bool ok = forked_mismatches == 0 && inst >= 1 && m_shadow > 0.05 && m_shadow < 1.0 && m_ao > 0.2 && m_ao <= 1.0 && m_gi > 0.0 && m_final > 0.0 && m_taa > 0.0 && m_gt > 0.0 &&
                        std::isfinite(m_refl) && std::isfinite(m_final) && m_ldr > 0.02 && m_ldr < 0.98;
        return ok ? 0 : 1;
    }
    catch (const hr::Error& e)
    {
        std::fprintf(stderr, "hr error: %s\n", e.what());
        return 4;
    }
}
