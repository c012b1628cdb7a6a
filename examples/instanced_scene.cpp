// C++ host example: a scene as the reference holds it — meshes + instances { model_matrix, mesh_idx } (scene_descriptor_set.glsl:30-34) — with one
// instance moving every frame: hr::Scene(ctx, hr_instanced_scene_desc) + scene.update_instances(matrices, cmd_buf) in the place of
// dw::RayTracedScene::build_tlas(cmd_buf) (main.cpp:74), then RayTracedShadows::render(cmd_buf) as ever.  Every frame's visibility mask is
// compared with the one a flattened hr_scene_create over the same world-space triangles gives: identical.
//
//   hipcc -std=c++17 -I include examples/instanced_scene.cpp -L hybrid_rendering_amd -lhybrid_rendering_amd \
//         -Wl,-rpath,$PWD/hybrid_rendering_amd -o /tmp/instanced_scene && /tmp/instanced_scene
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
    const float S = 100.0f;
    // ---- meshes (object space): the room, a unit cube ------------------------------------------------------------
    std::vector<float> room, cube;
    quad(room, { 0, 0, 0 }, { 0, 0, S }, { S, 0, S }, { S, 0, 0 });
    quad(room, { 0, S, 0 }, { S, S, 0 }, { S, S, S }, { 0, S, S });
    quad(room, { 0, 0, 0 }, { S, 0, 0 }, { S, S, 0 }, { 0, S, 0 });
    quad(room, { 0, 0, 0 }, { 0, S, 0 }, { 0, S, S }, { 0, 0, S });
    quad(room, { S, 0, 0 }, { S, 0, S }, { S, S, S }, { S, S, 0 });
    {
        const V3 lo { -0.5f, -0.5f, -0.5f }, hi { 0.5f, 0.5f, 0.5f };
        quad(cube, { lo.x, lo.y, lo.z }, { lo.x, lo.y, hi.z }, { lo.x, hi.y, hi.z }, { lo.x, hi.y, lo.z });
        quad(cube, { hi.x, lo.y, lo.z }, { hi.x, hi.y, lo.z }, { hi.x, hi.y, hi.z }, { hi.x, lo.y, hi.z });
        quad(cube, { lo.x, hi.y, lo.z }, { lo.x, hi.y, hi.z }, { hi.x, hi.y, hi.z }, { hi.x, hi.y, lo.z });
        quad(cube, { lo.x, lo.y, lo.z }, { hi.x, lo.y, lo.z }, { hi.x, lo.y, hi.z }, { lo.x, lo.y, hi.z });
        quad(cube, { lo.x, lo.y, lo.z }, { lo.x, hi.y, lo.z }, { hi.x, hi.y, lo.z }, { hi.x, lo.y, lo.z });
        quad(cube, { lo.x, lo.y, hi.z }, { hi.x, lo.y, hi.z }, { hi.x, hi.y, hi.z }, { lo.x, hi.y, hi.z });
    }
    hr_mesh_desc meshes[2] {};
    meshes[0].positions = room.data(); meshes[0].n_tris = (int)room.size() / 9;
    meshes[1].positions = cube.data(); meshes[1].n_tris = (int)cube.size() / 9;
    // ---- instances: the room (identity), three cubes: translate * rotate_y * scale, column-major ---------------------------
    auto model = [](float tx, float ty, float tz, float angle, float sx, float sy, float sz, float* m) {
        const float c = std::cos(angle), s = std::sin(angle);
        const float r[16] = { c * sx, 0, -s * sx, 0,  0, sy, 0, 0,  s * sz, 0, c * sz, 0,  tx, ty, tz, 1 };
        std::memcpy(m, r, sizeof(r));
    };
    const int I = 4;
    hr_instance inst[I] {};
    auto place = [&](int frame, float* mats) {
        model(0, 0, 0, 0, 1, 1, 1, mats);
        model(30, 30, 30, 0.4f, 30, 60, 30, mats + 16);
        model(70, 15, 65, -0.3f, 30, 30, 30, mats + 32);
        model(20.0f + 12.0f * frame, 70, 50, 0.25f * frame, 18, 10, 18, mats + 48);     // the one that flies under the light
    };
    float mats[I * 16];
    place(0, mats);
    for (int i = 0; i < I; i++) { std::memcpy(inst[i].model_matrix, mats + 16 * i, 64); inst[i].mesh_idx = i ? 1u : 0u; inst[i].mesh_id = 1u + i; }
    hr_instanced_scene_desc d {};
    d.meshes = meshes; d.n_meshes = 2; d.instances = inst; d.n_instances = I;

    try
    {
        hr::Context ctx(0);
        hr::Scene scene(ctx, d);                                   // per-mesh topologies built once, one subtree per instance
        if (scene.instance_count() != I) return 5;

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
        u.light.data0[1] = -1.0f; u.light.data0[3] = 5000.0f;
        u.light.data1[0] = 50; u.light.data1[1] = 95; u.light.data1[2] = 50; u.light.data1[3] = 0.0f;
        u.light.data2[0] = u.light.data2[1] = u.light.data2[2] = 1.0f;
        u.light.data3[0] = 1.0f;

        void *gb1, *gb2, *gb3, *depth, *sobol, *sr;
        HIP_OK(hipMalloc(&gb1, (size_t)W * H * 4)); HIP_OK(hipMalloc(&gb2, (size_t)W * H * 8));
        HIP_OK(hipMalloc(&gb3, (size_t)W * H * 8)); HIP_OK(hipMalloc(&depth, (size_t)W * H * 4));
        HIP_OK(hipMalloc(&sobol, 256 * 4)); HIP_OK(hipMalloc(&sr, 128 * 128 * 4));
        HIP_OK(hipMemset(sobol, 0, 256 * 4)); HIP_OK(hipMemset(sr, 0, 128 * 128 * 4));
        hr_gbuffer_level lvl { gb1, gb2, gb3, (const float*)depth, W, H };
        frame.inputs.cur = frame.inputs.prev = frame.inputs.cur_full = lvl;
        frame.inputs.sobol = (const uint8_t*)sobol;
        frame.inputs.scrambling_ranking = (const uint8_t*)sr;

        hr::RayTracedShadows shadows(ctx, W, H, hr::RAY_TRACE_SCALE_FULL_RES), shadows_flat(ctx, W, H, hr::RAY_TRACE_SCALE_FULL_RES);
        shadows.set_current_output(hr::RayTracedShadows::OUTPUT_RAY_TRACE);
        shadows_flat.set_current_output(hr::RayTracedShadows::OUTPUT_RAY_TRACE);
        int    same = 0;
        size_t lit_first = 0, lit_last = 0;
        for (uint32_t i = 0; i < 4; i++)
        {
            place((int)i, mats);
            scene.update_instances(mats, nullptr);                  // main.cpp:74 scene->build_tlas(cmd_buf)
            hr::check(hr_gbuffer_raycast(scene.handle(), &u, W, H, gb1, gb2, gb3, (float*)depth, nullptr), "hr_gbuffer_raycast");
            frame.inputs.num_frames = i;
            frame.inputs.ping_pong  = i & 1;
            shadows.render(nullptr, frame);                         // main.cpp:80
            // the same world-space triangles through the flattened constructor
            std::vector<float> flat;
            for (int k = 0; k < I; k++)
            {
                const std::vector<float>& src = k ? cube : room;
                const float* m = mats + 16 * k;
                for (size_t v = 0; v + 2 < src.size(); v += 3)
                    for (int r = 0; r < 3; r++) flat.push_back(((m[r] * src[v] + m[4 + r] * src[v + 1]) + m[8 + r] * src[v + 2]) + m[12 + r] * 1.0f);
            }
            hr_scene_desc fd {};
            fd.positions = flat.data(); fd.n_tris = (int)flat.size() / 9;
            hr::Scene flat_scene(ctx, fd);
            hr::Frame ff = frame;
            ff.scene = &flat_scene;
            shadows_flat.render(nullptr, ff);
            HIP_OK(hipDeviceSynchronize());
            hr::ImageView ma = shadows.output_ds(), mb = shadows_flat.output_ds();
            std::vector<uint32_t> wa((size_t)ma.width * ma.height), wb(wa.size());
            HIP_OK(hipMemcpy(wa.data(), ma.data, wa.size() * 4, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(wb.data(), mb.data, wb.size() * 4, hipMemcpyDeviceToHost));
            same += wa == wb ? 1 : 0;
            size_t lit = 0;
            for (uint32_t wv : wa) lit += (size_t)__builtin_popcount(wv);
            if (i == 0) lit_first = lit;
            lit_last = lit;
        }
        std::printf("instanced scene: %d instances of %d meshes, %dx%d, 4 frames with a moving instance: %d of 4 masks equal the flattened scene's; lit pixels %zu -> %zu\n",
                    I, 2, W, H, same, lit_first, lit_last);
        return (same == 4 && lit_first != lit_last && lit_last > 1000) ? 0 : 1;
    }
    catch (const hr::Error& e)
    {
        std::fprintf(stderr, "hr error: %s\n", e.what());
        return 4;
    }
}
