// C++ host example: BASELINE.json configs[0] (256x256 Cornell box, 32 triangles, hard shadows, 1 spp) driven
// through include/hr/passes.hpp — the same call shapes as the reference's frame loop (src/main.cpp:49-129):
// build the scene, fill the per-frame UBO, render(), read output_ds().
//
//   hipcc -std=c++17 -I include examples/cornell_shadows.cpp -L hybrid_rendering_amd -lhybrid_rendering_amd \
//         -Wl,-rpath,$PWD/hybrid_rendering_amd -o /tmp/cornell_shadows && /tmp/cornell_shadows
#include <hr/passes.hpp>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

struct V3 { float x, y, z; };

void quad(std::vector<float>& v, V3 a, V3 b, V3 c, V3 d)
{
    const V3 t[6] = { a, b, c, a, c, d };
    for (const V3& p : t) { v.push_back(p.x); v.push_back(p.y); v.push_back(p.z); }
}

// column-major 4x4 helpers (glm conventions)
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

#define HIP_OK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); return 2; } } while (0)

} // namespace

int main()
{
    const int W = 256, H = 256;
    // ---- scene: 5 walls + 2 boxes + light quad = 32 triangles --------------------------------------------------
    std::vector<float> v;
    const float S = 100.0f;
    quad(v, { 0, 0, 0 }, { 0, 0, S }, { S, 0, S }, { S, 0, 0 });           // floor
    quad(v, { 0, S, 0 }, { S, S, 0 }, { S, S, S }, { 0, S, S });           // ceiling
    quad(v, { 0, 0, 0 }, { S, 0, 0 }, { S, S, 0 }, { 0, S, 0 });           // back
    quad(v, { 0, 0, 0 }, { 0, S, 0 }, { 0, S, S }, { 0, 0, S });           // left
    quad(v, { S, 0, 0 }, { S, 0, S }, { S, S, S }, { S, S, 0 });           // right
    auto box = [&](V3 lo, V3 hi) {
        quad(v, { lo.x, lo.y, lo.z }, { lo.x, lo.y, hi.z }, { lo.x, hi.y, hi.z }, { lo.x, hi.y, lo.z });
        quad(v, { hi.x, lo.y, lo.z }, { hi.x, hi.y, lo.z }, { hi.x, hi.y, hi.z }, { hi.x, lo.y, hi.z });
        quad(v, { lo.x, hi.y, lo.z }, { lo.x, hi.y, hi.z }, { hi.x, hi.y, hi.z }, { hi.x, hi.y, lo.z });
        quad(v, { lo.x, lo.y, lo.z }, { lo.x, hi.y, lo.z }, { hi.x, hi.y, lo.z }, { hi.x, lo.y, lo.z });
        quad(v, { lo.x, lo.y, hi.z }, { hi.x, lo.y, hi.z }, { hi.x, hi.y, hi.z }, { lo.x, hi.y, hi.z });
    };
    box({ 15, 0, 15 }, { 45, 60, 45 });
    box({ 55, 0, 50 }, { 85, 30, 80 });
    quad(v, { 35, 99.5f, 35 }, { 65, 99.5f, 35 }, { 65, 99.5f, 65 }, { 35, 99.5f, 65 });
    const int n_tris = (int)v.size() / 9;

    try
    {
        hr::Context ctx(0);
        hr_scene_desc desc {};
        desc.positions = v.data();
        desc.n_tris    = n_tris;
        hr::Scene scene(ctx, desc);

        // ---- per-frame UBO (main.cpp:937-972): camera in front of the open side, point light under the ceiling --
        const float eye[3] = { 50, 50, 235 }, f = 1.0f / std::tan(40.0f * 3.14159265f / 360.0f), n = 1.0f, fa = 1000.0f;
        float view[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, -eye[0], -eye[1], -eye[2], 1 };
        float proj[16] = { f, 0, 0, 0, 0, -f, 0, 0, 0, 0, fa / (n - fa), -1, 0, 0, -(fa * n) / (fa - n), 0 };
        hr::Frame frame;
        frame.scene = &scene;
        hr_ubo& u = frame.inputs.ubo;
        std::memset(&frame.inputs, 0, sizeof(frame.inputs));
        mul(proj, view, u.view_proj);
        std::memcpy(u.prev_view_proj, u.view_proj, sizeof(u.view_proj));
        if (!invert(u.view_proj, u.view_proj_inverse)) return 3;
        invert(view, u.view_inverse);
        invert(proj, u.proj_inverse);
        u.cam_pos[0] = eye[0]; u.cam_pos[1] = eye[1]; u.cam_pos[2] = eye[2]; u.cam_pos[3] = 1.0f;
        u.light.data0[1] = -1.0f; u.light.data0[3] = 5000.0f;                       // direction (unused by a point light), intensity
        u.light.data1[0] = 50; u.light.data1[1] = 95; u.light.data1[2] = 50; u.light.data1[3] = 0.0f; // position, radius 0 = hard
        u.light.data2[0] = u.light.data2[1] = u.light.data2[2] = 1.0f;
        u.light.data3[0] = 1.0f;                                                     // LIGHT_TYPE_POINT

        // ---- G-buffer + blue-noise tables in HBM -----------------------------------------------------------------
        void *gb1, *gb2, *gb3, *depth, *sobol, *sr;
        HIP_OK(hipMalloc(&gb1, (size_t)W * H * 4)); HIP_OK(hipMalloc(&gb2, (size_t)W * H * 8));
        HIP_OK(hipMalloc(&gb3, (size_t)W * H * 8)); HIP_OK(hipMalloc(&depth, (size_t)W * H * 4));
        HIP_OK(hipMalloc(&sobol, 256 * 4)); HIP_OK(hipMalloc(&sr, 128 * 128 * 4));
        HIP_OK(hipMemset(sobol, 0, 256 * 4)); HIP_OK(hipMemset(sr, 0, 128 * 128 * 4)); // hard light: the sample is irrelevant
        hr::check(hr_gbuffer_raycast(scene.handle(), &u, W, H, gb1, gb2, gb3, (float*)depth, nullptr), "hr_gbuffer_raycast");
        hr_gbuffer_level lvl { gb1, gb2, gb3, (const float*)depth, W, H };
        frame.inputs.cur = frame.inputs.prev = frame.inputs.cur_full = lvl;
        frame.inputs.sobol = (const uint8_t*)sobol;
        frame.inputs.scrambling_ranking = (const uint8_t*)sr;

        // ---- the pass, exactly as main.cpp:80 uses it ------------------------------------------------------------
        hr::RayTracedShadows shadows(ctx, W, H, hr::RAY_TRACE_SCALE_FULL_RES);
        for (uint32_t i = 0; i < 4; i++)
        {
            frame.inputs.num_frames = i;
            frame.inputs.ping_pong  = i & 1;
            shadows.render(nullptr, frame);
        }
        HIP_OK(hipDeviceSynchronize());
        shadows.set_current_output(hr::RayTracedShadows::OUTPUT_RAY_TRACE);
        hr::ImageView mask = shadows.output_ds();
        std::vector<uint32_t> words((size_t)mask.width * mask.height);
        HIP_OK(hipMemcpy(words.data(), mask.data, words.size() * 4, hipMemcpyDeviceToHost));
        size_t lit = 0;
        for (uint32_t wv : words) lit += (size_t)__builtin_popcount(wv);
        uint64_t rays = 0;
        hr::check(hr_shadows_ray_count(shadows.handle(), &rays), "hr_shadows_ray_count");
        std::printf("cornell32: %d triangles, %dx%d, %llu shadow rays, lit fraction %.4f\n", n_tris, W, H, (unsigned long long)rays, (double)lit / (W * H));
        return (n_tris == 32 && rays > 10000 && lit > 1000 && lit < (size_t)W * H) ? 0 : 1;
    }
    catch (const hr::Error& e)
    {
        std::fprintf(stderr, "hr error: %s\n", e.what());
        return 4;
    }
}
