// C++ host example of the native multi-GPU path (include/hr_comm.h, include/hr/tiled.hpp): the whole hybrid frame — RayTracedShadows,
// RayTracedAO, DDGI (probes sharded by z-slab, atlas rows all-gathered) and RayTracedReflections (fed by the DDGI atlases) — row-tiled
// over TWO ranks, one host thread per rank, four frames with a moving camera; every band row of every pass is compared with an
// un-tiled render of the same frame.
// With two or more GPUs visible the ranks run on GPU 0 and 1 and talk RCCL over xGMI; on a one-GPU box both ranks share the device
// and the in-process loopback back end carries the rows (RCCL refuses two ranks on one device).  No Python, no torch.
//
//   hipcc -std=c++17 -I include examples/tiled_frame.cpp -L hybrid_rendering_amd -lhr_comm -lhybrid_rendering_amd -lpthread \
//         -Wl,-rpath,$PWD/hybrid_rendering_amd -o /tmp/tiled_frame && /tmp/tiled_frame
#include <hr/tiled.hpp>
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

namespace {

bool g_free_running = false;   // --free-running (see the frame loop)

struct V3 { float x, y, z; };
void quad(std::vector<float>& v, V3 a, V3 b, V3 c, V3 d)
{
    const V3 t[6] = { a, b, c, a, c, d };
    for (const V3& p : t) { v.push_back(p.x); v.push_back(p.y); v.push_back(p.z); }
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
    const int32_t  e = (int32_t)((u >> 23) & 0xff) - 127 + 15;
    const uint32_t m = u & 0x7fffffu;
    if (e <= 0) return (uint16_t)sign;
    if (e >= 31) return (uint16_t)(sign | 0x7bffu);
    uint32_t h = (uint32_t)(e << 10) | (m >> 13);
    const uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}
template <typename T>
T* upload(const std::vector<T>& v)
{
    void* d = nullptr;
    if (hipMalloc(&d, v.size() * sizeof(T)) != hipSuccess) return nullptr;
    (void)hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    return (T*)d;
}

constexpr int W = 256, H = 192, kFrames = 4, kWorld = 2;
std::atomic<int> g_failures { 0 };

void fill_ubo(hr_ubo& u, uint32_t f, float* prev_vp)
{
    std::memset(&u, 0, sizeof(u));
    const float fy = 1.0f / std::tan(40.0f * 3.14159265f / 360.0f), fx = fy * (float)H / (float)W, n = 1.0f, fa = 1000.0f;
    const float eye[3] = { 50.0f + 2.5f * f, 50.0f + 1.0f * f, 235.0f - 3.0f * f };
    float view[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, -eye[0], -eye[1], -eye[2], 1 };
    float proj[16] = { fx, 0, 0, 0, 0, -fy, 0, 0, 0, 0, fa / (n - fa), -1, 0, 0, -(fa * n) / (fa - n), 0 };
    mul(proj, view, u.view_proj);
    std::memcpy(u.prev_view_proj, f ? prev_vp : u.view_proj, 64);
    std::memcpy(prev_vp, u.view_proj, 64);
    invert(u.view_proj, u.view_proj_inverse); invert(view, u.view_inverse); invert(proj, u.proj_inverse);
    u.cam_pos[0] = eye[0]; u.cam_pos[1] = eye[1]; u.cam_pos[2] = eye[2]; u.cam_pos[3] = 1.0f;
    u.light.data0[1] = -1.0f; u.light.data0[3] = 9000.0f;
    u.light.data1[0] = 50; u.light.data1[1] = 92; u.light.data1[2] = 50; u.light.data1[3] = 3.0f;   // soft point light under the ceiling
    u.light.data2[0] = u.light.data2[1] = u.light.data2[2] = 1.0f;
    u.light.data3[0] = 1.0f;
}

std::vector<uint8_t> download(const hr::ImageView& v)
{
    std::vector<uint8_t> h((size_t)v.row_pitch_bytes * v.height);
    (void)hipMemcpy(h.data(), v.data, h.size(), hipMemcpyDeviceToHost);
    return h;
}

void rank_main(int rank, int device, bool use_rccl, const uint8_t* id, const std::vector<float>* verts)
{
    try
    {
        (void)hipSetDevice(device);
        hr::Context ctx(device);
        hr_scene_desc desc {};
        desc.positions = verts->data(); desc.n_tris = (int32_t)(verts->size() / 9);
        hr::Scene scene(ctx, desc);
        hr::Comm comm = use_rccl ? hr::Comm(ctx, kWorld, rank, id) : hr::Comm::loopback(ctx, kWorld, rank, "tiled_frame");
        hipStream_t stream;
        (void)hipStreamCreate(&stream);
        // environment: a 16^2 sky cubemap (blue above, dark below), a nearest-decimated chain, a flat BRDF LUT
        const int SK = 16, LV = 4;
        std::vector<uint16_t> sky((size_t)6 * SK * SK * 4), pre, lut((size_t)16 * 16 * 2);
        for (int f = 0; f < 6; f++)
            for (int i = 0; i < SK * SK; i++)
            {
                const bool  up = f == 2 || (f != 3 && (i / SK) < SK / 2);
                const float c[3] = { up ? 0.35f : 0.08f, up ? 0.55f : 0.07f, up ? 0.9f : 0.06f };
                for (int k = 0; k < 3; k++) sky[((size_t)f * SK * SK + i) * 4 + k] = f2h(c[k]);
                sky[((size_t)f * SK * SK + i) * 4 + 3] = f2h(1.0f);
            }
        for (int l = 0, sz = SK; l < LV; l++, sz >>= 1)
            for (int f = 0; f < 6; f++)
                for (int y = 0; y < sz; y++)
                    for (int x = 0; x < sz; x++)
                        for (int k = 0; k < 4; k++) pre.push_back(sky[(((size_t)f * SK + (y << l)) * SK + (x << l)) * 4 + k]);
        for (size_t i = 0; i < lut.size(); i += 2) { lut[i] = f2h(0.9f); lut[i + 1] = f2h(0.05f); }
        hr_environment env {};
        env.sky = upload(sky); env.sky_size = SK;
        env.prefiltered = upload(pre); env.prefiltered_size = SK; env.prefiltered_levels = LV;
        env.brdf_lut = upload(lut); env.brdf_lut_size = 16;
        // DDGI grid 5 x 5 x 5 over the room (ddgi.cpp:150-169, :197-201): rank 0 traces and updates z-slabs [0, 2), rank 1 [2, 5)
        hr_ddgi_uniforms g {};
        for (int a = 0; a < 3; a++) { g.grid_start_position[a] = 0.0f; g.grid_step[a] = 100.0f / 3.0f; g.probe_counts[a] = 5; }
        g.max_distance = g.grid_step[0] * 1.5f; g.depth_sharpness = 50.0f; g.hysteresis = 0.98f; g.normal_bias = 1.0f; g.energy_preservation = 0.85f;
        g.irradiance_probe_side_length = 8; g.depth_probe_side_length = 16; g.rays_per_probe = 64; g.visibility_test = 1;
        g.irradiance_texture_width = 10 * 25 + 2; g.irradiance_texture_height = 10 * 5 + 2;
        g.depth_texture_width = 18 * 25 + 2; g.depth_texture_height = 18 * 5 + 2;
        hr::CommonResources common;
        hr::GBuffer         g_buffer;
        common.scene = &scene; common.environment = &env;
        g_buffer.current[0].width = W; g_buffer.current[0].height = H;
        const std::vector<int32_t> bounds = hr::uniform_bounds(H, kWorld);
        hr::TiledShadows     shadows(ctx, comm, &common, &g_buffer, bounds);
        hr::TiledAO          ao(ctx, comm, &common, &g_buffer, bounds, hr::RAY_TRACE_SCALE_FULL_RES);
        hr::ShardedDDGI      ddgi(ctx, comm, &common, &g_buffer, g, bounds);
        hr::TiledReflections reflections(ctx, comm, &common, &g_buffer, bounds, hr::RAY_TRACE_SCALE_FULL_RES);
        hr::TiledHybridFrame tiled(ctx, &shadows, &ao, &ddgi, &reflections);
        hr::RayTracedShadows     whole_shadows(ctx, &common, &g_buffer);          // the un-tiled reference, rendered by every rank
        hr::RayTracedAO          whole_ao(ctx, &common, &g_buffer, hr::RAY_TRACE_SCALE_FULL_RES);
        hr::DDGI                 whole_ddgi(ctx, &common, &g_buffer, g);
        hr::RayTracedReflections whole_reflections(ctx, &common, &g_buffer, hr::RAY_TRACE_SCALE_FULL_RES);
        void *gb1[2], *gb2[2], *gb3[2], *depth[2];
        for (int i = 0; i < 2; i++)
        {
            (void)hipMalloc(&gb1[i], (size_t)W * H * 4); (void)hipMalloc(&gb2[i], (size_t)W * H * 8);
            (void)hipMalloc(&gb3[i], (size_t)W * H * 8); (void)hipMalloc(&depth[i], (size_t)W * H * 4);
        }
        std::vector<uint8_t> sob(256 * 4), sr(128 * 128 * 4);
        uint32_t lcg = 12345u;
        for (auto& b : sob) { lcg = lcg * 1664525u + 1013904223u; b = (uint8_t)(lcg >> 24); }
        for (auto& b : sr) { lcg = lcg * 1664525u + 1013904223u; b = (uint8_t)(lcg >> 24); }
        void *sob_d, *sr_d;
        (void)hipMalloc(&sob_d, sob.size()); (void)hipMemcpy(sob_d, sob.data(), sob.size(), hipMemcpyHostToDevice);
        (void)hipMalloc(&sr_d, sr.size()); (void)hipMemcpy(sr_d, sr.data(), sr.size(), hipMemcpyHostToDevice);
        common.sobol = (const uint8_t*)sob_d; common.scrambling_ranking = (const uint8_t*)sr_d;
        common.z_buffer_params[0] = 1.0f - 1000.0f; common.z_buffer_params[1] = 1000.0f;
        common.z_buffer_params[2] = common.z_buffer_params[0] / 1000.0f; common.z_buffer_params[3] = 1.0f;
        float prev_vp[16] = { 0 };
        for (uint32_t f = 0; f < (uint32_t)kFrames; f++)
        {
            const int pp = (int)(f & 1);
            fill_ubo(common.ubo, f, prev_vp);
            hr::check(hr_gbuffer_raycast(scene.handle(), &common.ubo, W, H, gb1[pp], gb2[pp], gb3[pp], (float*)depth[pp], stream), "hr_gbuffer_raycast");
            hr_gbuffer_level cur { gb1[pp], gb2[pp], gb3[pp], (const float*)depth[pp], W, H };
            g_buffer.current[0] = cur;
            g_buffer.history[0] = f ? hr_gbuffer_level { gb1[!pp], gb2[!pp], gb3[!pp], (const float*)depth[!pp], W, H } : cur;
            common.num_frames = f; common.ping_pong = pp != 0;
            // main.cpp:80-81 for this rank's band: hr::TiledHybridFrame forks the two passes over its streams; each posts its neighbour
            // exchange from inside render() (one communicator, per-pass tickets) — odd frames take the plain serial calls for comparison
            // --free-running: every frame forked and NO host synchronisation between frames — the order between a frame's temporal
            // kernels and the neighbour's apron rows of the previous frame then rests on the communicator's tickets alone
            if (!g_free_running && (f & 1)) tiled.render(stream, /*forked=*/false);
            else tiled.render(stream);
            whole_shadows.render(stream);
            whole_ao.render(stream);
            whole_ddgi.render(stream);
            whole_reflections.render(stream, &whole_ddgi);
            if (g_free_running && f + 1 < (uint32_t)kFrames) continue;   // only the last frame is looked at
            (void)hipStreamSynchronize(stream);
            // band rows of this rank == the same rows of the un-tiled render, bit for bit
            whole_shadows.set_current_output(hr::RayTracedShadows::OUTPUT_ATROUS);
            shadows.pass().set_current_output(hr::RayTracedShadows::OUTPUT_ATROUS);
            const hr::ImageView vs = shadows.pass().output_ds(), ws = whole_shadows.output_ds(), va = ao.pass().output_ds(), wa = whole_ao.output_ds();
            const std::vector<uint8_t> a = download(vs), b = download(ws), c = download(va), d = download(wa);
            const size_t s0 = (size_t)bounds[rank] * vs.row_pitch_bytes, s1 = (size_t)bounds[rank + 1] * vs.row_pitch_bytes;
            const size_t a0 = (size_t)bounds[rank] * va.row_pitch_bytes, a1 = (size_t)bounds[rank + 1] * va.row_pitch_bytes;
            const bool ok_s = std::memcmp(a.data() + s0, b.data() + s0, s1 - s0) == 0, ok_a = std::memcmp(c.data() + a0, d.data() + a0, a1 - a0) == 0;
            // DDGI: this rank's rows of the per-pixel sample, and BOTH whole atlases (own slabs traced here, the others all-gathered)
            const hr::ImageView vd = ddgi.pass().output_ds(), wd = whole_ddgi.output_ds();
            const std::vector<uint8_t> dd = download(vd), dw = download(wd);
            const size_t d0 = (size_t)bounds[rank] * vd.row_pitch_bytes, d1 = (size_t)bounds[rank + 1] * vd.row_pitch_bytes;
            bool ok_d = std::memcmp(dd.data() + d0, dw.data() + d0, d1 - d0) == 0;
            hr::ImageView ti, td, wi, wdep;
            ddgi.pass().current_read_ds(ti, td);
            whole_ddgi.current_read_ds(wi, wdep);
            ok_d = ok_d && download(ti) == download(wi) && download(td) == download(wdep);
            // reflections: a-trous output rows of the band (full resolution here, so no upsample stage)
            whole_reflections.set_current_output(hr::RayTracedReflections::OUTPUT_ATROUS);
            reflections.pass().set_current_output(hr::RayTracedReflections::OUTPUT_ATROUS);
            const hr::ImageView vr = reflections.pass().output_ds(), wr = whole_reflections.output_ds();
            const std::vector<uint8_t> rr = download(vr), rw = download(wr);
            const size_t r0 = (size_t)bounds[rank] * vr.row_pitch_bytes, r1 = (size_t)bounds[rank + 1] * vr.row_pitch_bytes;
            const bool ok_r = std::memcmp(rr.data() + r0, rw.data() + r0, r1 - r0) == 0;
            std::printf("rank %d frame %u (%s): rows %d-%d  shadows %s  ao %s  ddgi %s  reflections %s\n", rank, f, (!g_free_running && (f & 1)) ? "serial" : "forked", bounds[rank], bounds[rank + 1],
                        ok_s ? "==" : "DIFFER", ok_a ? "==" : "DIFFER", ok_d ? "==" : "DIFFER", ok_r ? "==" : "DIFFER");
            if (!ok_s || !ok_a || !ok_d || !ok_r) g_failures++;
            if (shadows.history_apron_exceeded() || ao.history_apron_exceeded() || reflections.history_apron_exceeded()) { std::printf("rank %d frame %u: motion beyond the history apron\n", rank, f); g_failures++; }
        }
        comm.wait(stream);
        (void)hipStreamSynchronize(stream);
    }
    catch (const hr::Error& e)
    {
        std::fprintf(stderr, "rank %d: hr error: %s\n", rank, e.what());
        g_failures++;
    }
}

} // namespace

int main(int argc, char** argv)
{
    for (int i = 1; i < argc; i++) if (!std::strcmp(argv[i], "--free-running")) g_free_running = true;
    // Cornell-style room with two boxes
    const float S = 100.0f;
    std::vector<float> v;
    quad(v, { 0, 0, 0 }, { S, 0, 0 }, { S, 0, S }, { 0, 0, S });
    quad(v, { 0, S, 0 }, { 0, S, S }, { S, S, S }, { S, S, 0 });
    quad(v, { 0, 0, 0 }, { 0, S, 0 }, { S, S, 0 }, { S, 0, 0 });
    quad(v, { 0, 0, 0 }, { 0, 0, S }, { 0, S, S }, { 0, S, 0 });
    quad(v, { S, 0, 0 }, { S, S, 0 }, { S, S, S }, { S, 0, S });
    for (int b = 0; b < 2; b++)
    {
        const float x0 = b ? 58.0f : 18.0f, z0 = b ? 20.0f : 50.0f, w = 26.0f, h = b ? 55.0f : 28.0f;
        quad(v, { x0, h, z0 }, { x0, h, z0 + w }, { x0 + w, h, z0 + w }, { x0 + w, h, z0 });
        quad(v, { x0, 0, z0 }, { x0, h, z0 }, { x0 + w, h, z0 }, { x0 + w, 0, z0 });
        quad(v, { x0, 0, z0 + w }, { x0 + w, 0, z0 + w }, { x0 + w, h, z0 + w }, { x0, h, z0 + w });
        quad(v, { x0, 0, z0 }, { x0, 0, z0 + w }, { x0, h, z0 + w }, { x0, h, z0 });
        quad(v, { x0 + w, 0, z0 }, { x0 + w, h, z0 }, { x0 + w, h, z0 + w }, { x0 + w, 0, z0 + w });
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { std::fprintf(stderr, "no HIP device\n"); return 2; }
    const bool use_rccl = ndev >= kWorld;
    uint8_t id[HR_COMM_ID_BYTES] = { 0 };
    if (use_rccl && hr_comm_get_unique_id(id) != HR_OK) { std::fprintf(stderr, "RCCL unavailable: %s\n", hr_last_error()); return 3; }
    std::printf("tiled_frame: %d ranks, transport %s\n", kWorld, use_rccl ? "RCCL" : "in-process loopback (one GPU)");
    std::vector<std::thread> th;
    for (int r = 0; r < kWorld; r++) th.emplace_back(rank_main, r, use_rccl ? r : 0, use_rccl, id, &v);
    for (auto& t : th) t.join();
    std::printf("tiled_frame: %s\n", g_failures == 0 ? "every band row equals the un-tiled render" : "MISMATCH");
    return g_failures == 0 ? 0 : 1;
}
