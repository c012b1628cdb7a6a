// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).  Pinned against the reference's own shaders: tests/test_ref_shaders.py.
// Restatement of the ray-traced ambient-occlusion pass:
//   A1 ao/ao_ray_trace.comp:90-126 (+ brdf.glsl:8-32 sample_cosine_lobe / make_rotation_matrix,
//      ray_query.glsl:6-30 query_visibility)
//   A3 ao/ao_denoise_reprojection.comp:191-260
//   A4 ao/ao_denoise_bilateral_blur.comp:75-139 (two passes: direction (1,0) then (0,1),
//      ray_traced_ao.cpp:1042,1066,1091,1115)
//   A5 ao/ao_upsample.comp:63-112 -> orc_upsample(sky_value = 1, power)
// Extension (not in the reference, BASELINE.json configs[2] asks for 4 spp): `spp` samples per pixel,
// sample index = spp*num_frames + s, one mask plane per sample; spp = 1 is the reference.
#include "orc_api.h"
#include "orc_bvh.h"
#include "orc_reproject.h"

using namespace orc;

namespace orc {

// brdf.glsl:8-16
void make_rotation_matrix(vec3 z, vec3* x, vec3* y)
{
    const vec3 ref = std::fabs(dot(z, v3(0, 1, 0))) > 0.99f ? v3(0, 0, 1) : v3(0, 1, 0);
    *x = normalize(cross(ref, z));
    *y = cross(z, *x);
}

// brdf.glsl:20-32
vec3 sample_cosine_lobe(vec3 n, float rx, float ry)
{
    rx = fmax2(0.00001f, rx);
    ry = fmax2(0.00001f, ry);
    const float phi       = 2.0f * ORC_M_PI * ry;
    const float cos_theta = std::sqrt(rx);
    const float sin_theta = std::sqrt(1.0f - rx);
    float s, c;
    det_sincos(phi, &s, &c);
    const vec3 t = v3(sin_theta * c, sin_theta * s, cos_theta);
    vec3 x, y;
    make_rotation_matrix(n, &x, &y);
    // mat3(x, y, z) * t
    vec3 r = v3((x.x * t.x + y.x * t.y) + n.x * t.z, (x.y * t.x + y.y * t.y) + n.y * t.z, (x.z * t.x + y.z * t.y) + n.z * t.z);
    return normalize(r);
}

} // namespace orc

extern "C" {

// A1.  mask: [spp][ceil(h/4)][ceil(w/8)]; pixels outside the image contribute 0.
void orc_ao_ray_trace(const void* scene_, const void* ubo_, int w, int h, const float* depth, const uint16_t* gb2, const uint8_t* sobol,
                      const uint8_t* scrambling_ranking, float bias, float ray_length, uint32_t num_frames, int spp, uint32_t* mask, uint64_t* rays_out)
{
    const Scene& scene = *(const Scene*)scene_;
    const UBO&   ubo   = *(const UBO*)ubo_;
    BlueNoise    bn { sobol, scrambling_ranking };
    ImgH<4>      g2 { gb2, w, h };
    const int    mw = ceil_div(w, 8), mh = ceil_div(h, 4);
    uint64_t     rays = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : rays)
    for (int my = 0; my < mh; my++)
        for (int mx = 0; mx < mw; mx++)
            for (int s = 0; s < spp; s++)
            {
                uint32_t bits = 0;
                for (int ly = 0; ly < 4; ly++)
                    for (int lx = 0; lx < 8; lx++)
                    {
                        int x = mx * 8 + lx, y = my * 4 + ly;
                        // no bounds check in ao_ray_trace.comp:90-126: edge threads read depth 0 / normal (0,0) and trace
                        float    d      = (x < w && y < h) ? depth[(size_t)y * w + x] : 0.0f;
                        uint32_t result = 0;
                        if (d != 1.0f)
                        {
                            float tu = ((float)x + 0.5f) / (float)w, tv = ((float)y + 0.5f) / (float)h;
                            vec3  world_pos  = world_position_from_depth(tu, tv, d, ubo.view_proj_inverse);
                            vec3  normal     = octohedral_to_direction(g2.fetch(x, y, 0), g2.fetch(x, y, 1));
                            vec3  ray_origin = world_pos + normal * bias;
                            int   idx        = (int)num_frames * spp + s;
                            float r0 = sample_blue_noise(x, y, idx, 0, bn), r1 = sample_blue_noise(x, y, idx, 1, bn);
                            vec3  dir = sample_cosine_lobe(normal, r0, r1);
                            rays++;
                            result = scene.any_hit(ray_origin, dir, 0.01f, ray_length) ? 0u : 1u;
                        }
                        bits |= result << (ly * 8 + lx);
                    }
                mask[((size_t)s * mh + my) * mw + mx] = bits;
            }
    if (rays_out) *rays_out = rays;
}

static inline float unpack_bit(const ImgU& m, int x, int y, uint32_t oob)
{
    int mx = x >= 0 ? x >> 3 : -((-x + 7) >> 3);
    int my = y >= 0 ? y >> 2 : -((-y + 3) >> 2);
    uint32_t word = m.fetch(mx, my, oob);
    int bx = x - mx * 8, by = y - my * 4;
    return (float)((word >> (by * 8 + bx)) & 1u);
}

// A3.  out_ao R16F, out_len R16F; hist_ao / hist_len = previous frame's outputs.  tile_class: 1 = blur.
void orc_ao_temporal(const void* ubo_, int w, int h, int spp, const uint32_t* mask, const float* depth, const uint16_t* gb2, const uint16_t* gb3,
                     const float* prev_depth, const uint16_t* prev_gb2, const uint16_t* prev_gb3, const uint16_t* hist_ao,
                     const uint16_t* hist_len, float alpha, uint16_t* out_ao, uint16_t* out_len, uint8_t* tile_class)
{
    const UBO& ubo = *(const UBO*)ubo_;
    const int  mw = ceil_div(w, 8), mh = ceil_div(h, 4);
    ImgH<4>    g2 { gb2, w, h }, g3 { gb3, w, h }, pg2 { prev_gb2, w, h }, pg3 { prev_gb3, w, h };
    ImgF       pd { prev_depth, w, h };
    ImgH<1>    ha { hist_ao, w, h }, hl { hist_len, w, h };
    ImgHW<1>   oa { out_ao, w, h }, ol { out_len, w, h };
    const int  tw = ceil_div(w, 8), th = ceil_div(h, 8);
#pragma omp parallel for schedule(dynamic, 2)
    for (int ty = 0; ty < th; ty++)
        for (int tx = 0; tx < tw; tx++)
        {
            bool should_denoise = false;
            for (int ly = 0; ly < 8; ly++)
                for (int lx = 0; lx < 8; lx++)
                {
                    int x = tx * 8 + lx, y = ty * 8 + ly;
                    // no bounds check in the shader (:191-260): edge threads of ragged groups read depth 0 / G-buffer 0, run
                    // the body (stores dropped) and vote in g_should_denoise
                    // neighbourhood mean (:157-185); out-of-image MASK TEXELS read all-ones (:111-112)
                    float sum = 0.0f;
                    for (int s = 0; s < spp; s++)
                    {
                        ImgU m { mask + (size_t)s * mh * mw, mw, mh };
                        for (int yy = -8; yy <= 8; yy++)
                            for (int xx = -8; xx <= 8; xx++) sum += unpack_bit(m, x + xx, y + yy, 0xFFFFFFFFu);
                    }
                    const float mean = sum / (289.0f * (float)spp);
                    const float d = (x < w && y < h) ? depth[(size_t)y * w + x] : 0.0f;
                    float out = 1.0f, history_length = 0.0f;
                    if (d != 1.0f)
                    {
                        float ao = 0.0f;
                        for (int s = 0; s < spp; s++)
                        {
                            ImgU m { mask + (size_t)s * mh * mw, mw, mh };
                            ao += unpack_bit(m, x, y, 0xFFFFFFFFu);
                        }
                        ao = ao / (float)spp;
                        float history_ao;
                        ReprojectIn in;
                        in.x = x; in.y = y; in.depth = d;
                        in.view_proj_inverse = &ubo.view_proj_inverse;
                        in.gb2 = g2; in.gb3 = g3; in.pgb2 = pg2; in.pgb3 = pg3; in.pdepth = pd;
                        in.w = w; in.h = h;
                        bool success = reproject<true, false, false, 1>(in, ha, nullptr, &hl, &history_ao, nullptr, &history_length);
                        history_length = fmin2(32.0f, success ? history_length + 1.0f : 1.0f);
                        if (success)
                        {
                            float spatial_variance = mean;
                            spatial_variance       = fmax2(spatial_variance - mean * mean, 0.0f);
                            const float sd = std::sqrt(spatial_variance);
                            history_ao     = clampf(history_ao, mean - 0.5f * sd, mean + 0.5f * sd);
                        }
                        const float a = success ? fmax2(alpha, 1.0f / history_length) : 1.0f;
                        out = mixf(history_ao, ao, a);
                    }
                    oa.store(x, y, 0, out);
                    ol.store(x, y, 0, history_length);
                    if (out < 1.0f) should_denoise = true;
                }
            tile_class[(size_t)ty * tw + tx] = should_denoise ? 1 : 0;
        }
}

// A4: one separable pass.  Tiles not flagged keep the cleared value 1.0 (ray_traced_ao.cpp:1048-1055).
void orc_ao_blur(int w, int h, const uint16_t* in_ao, const float* depth, const uint16_t* gb2, const uint8_t* tile_class, const float* zbp,
                 int dir_x, int dir_y, int radius, uint16_t* out_ao)
{
    ImgH<1>   in { in_ao, w, h };
    ImgH<4>   g2 { gb2, w, h };
    ImgF      dp { depth, w, h };
    ImgHW<1>  out { out_ao, w, h };
    const int tw = ceil_div(w, 8), th = ceil_div(h, 8);
#pragma omp parallel for schedule(dynamic, 2)
    for (int ty = 0; ty < th; ty++)
        for (int tx = 0; tx < tw; tx++)
        {
            const bool denoise = tile_class[(size_t)ty * tw + tx] != 0;
            for (int ly = 0; ly < 8; ly++)
                for (int lx = 0; lx < 8; lx++)
                {
                    int x = tx * 8 + lx, y = ty * 8 + ly;
                    if (x >= w || y >= h) continue;
                    if (!denoise) { out.store(x, y, 0, 1.0f); continue; }
                    const float d = dp.fetch(x, y);
                    if (d == 1.0f) { out.store(x, y, 0, 1.0f); continue; }
                    const float deviation = (float)radius / 1.5f;
                    float total_ao = in.fetch(x, y, 0), total_weight = 1.0f;
                    const float center_depth  = linear_eye_depth(d, zbp);
                    const vec3  center_normal = octohedral_to_direction(g2.fetch(x, y, 0), g2.fetch(x, y, 1));
                    for (int i = -radius; i <= radius; i++)
                    {
                        if (i == 0) continue;
                        const int   sx = x + dir_x * i, sy = y + dir_y * i;
                        const float sample_depth  = linear_eye_depth(dp.fetch(sx, sy), zbp);
                        const float sample_ao     = in.fetch(sx, sy, 0);
                        const vec3  sample_normal = octohedral_to_direction(g2.fetch(sx, sy, 0), g2.fetch(sx, sy, 1));
                        float weight = gaussian_weight((float)i, deviation);
                        const float wZ = det_exp(-std::fabs(center_depth - sample_depth) / 1.0f);
                        const float wN = det_pow_auto(clampf(dot(center_normal, sample_normal), 0.0f, 1.0f), 32.0f);
                        weight = weight * (det_exp((0.0f - 1.0f) - fmax2(wZ, 0.0f)) * wN);
                        total_ao += weight * sample_ao;
                        total_weight += weight;
                    }
                    out.store(x, y, 0, total_ao / fmax2(total_weight, 0.0001f));
                }
        }
}

} // extern "C"
