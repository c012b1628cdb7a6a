// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).  Pinned against the reference's own shaders: tests/test_ref_shaders.py.
// Restatement of the ray-traced soft-shadow pass:
//   S1 shadows/shadows_ray_trace.comp:89-132  (+ lighting.glsl:6-111 SOFT_SHADOWS|SHADOW_RAY_ONLY,
//      ray_query.glsl:34-59 query_distance)
//   S3 shadows/shadows_denoise_reprojection.comp:196-293
//   S4 shadows/shadows_denoise_copy_shadow_tiles.comp:32-36  (folded into S5 via the tile class)
//   S5 shadows/shadows_denoise_atrous.comp:94-174
//   S6 shadows/shadows_upsample.comp:62-109
#include "orc_api.h"
#include "orc_bvh.h"
#include "orc_reproject.h"

using namespace orc;

namespace orc {

// lighting.glsl:6-111, variant SOFT_SHADOWS + SHADOW_RAY_ONLY + RAY_TRACING
void fetch_light_properties_shadow(const Light& light, vec3 P, vec3 N, float rx, float ry, vec3* Wi, float* t_max, float* attenuation)
{
    const int  type      = (int)light.data3[0];
    const vec3 ldir      = v3(light.data0[0], light.data0[1], light.data0[2]);
    const vec3 lpos      = v3(light.data1[0], light.data1[1], light.data1[2]);
    const float lradius  = light.data1[3];
    vec3  light_dir;
    float radius;
    if (type == 0)
    {
        light_dir    = ldir;
        radius       = lradius;
        *t_max       = 10000.0f;
        *attenuation = 1.0f;
    }
    else
    {
        vec3  to_light       = lpos - P;
        light_dir            = normalize(to_light);
        float light_distance = length(to_light);
        radius               = lradius / light_distance;
        *t_max               = light_distance;
        *attenuation         = 1.0f / (light_distance * light_distance);
    }
    vec3  light_tangent   = normalize(cross(light_dir, v3(0.0f, 1.0f, 0.0f)));
    vec3  light_bitangent = normalize(cross(light_tangent, light_dir));
    float point_radius    = radius * std::sqrt(rx);
    float point_angle     = ry * 2.0f * ORC_M_PI;
    float s, c;
    det_sincos(point_angle, &s, &c);
    float dx = point_radius * c, dy = point_radius * s;
    *Wi      = normalize((light_dir + dx * light_tangent) + dy * light_bitangent);
    if (type == 2)
    {
        float angle_attenuation = dot(*Wi, ldir);
        angle_attenuation       = smoothstepf(light.data3[1], light.data3[2], angle_attenuation);
        float light_distance    = *t_max;
        *attenuation            = angle_attenuation / (light_distance * light_distance);
    }
    *attenuation = *attenuation * clampf(dot(N, *Wi), 0.0f, 1.0f);
}

} // namespace orc

extern "C" {

// S1.  mask: [ceil(h/4)][ceil(w/8)] uint32, bit (y&3)*8 + (x&7).  The shader has no bounds check
// (shadows_ray_trace.comp:89-132): a thread of a ragged edge group fetches depth 0 and normal (0,0) (pinned rule: texel
// fetches outside the image return 0), fires its ray and contributes its bit (verified against the
// reference shader at ragged sizes by tests/test_ref_shaders.py).  rays_out (optional): number of rays fired.
void orc_shadows_ray_trace(const void* scene_, const void* ubo_, int w, int h, const float* depth, const uint16_t* gb2,
                           const uint8_t* sobol, const uint8_t* scrambling_ranking, float bias, uint32_t num_frames,
                           uint32_t* mask, uint64_t* rays_out)
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
        {
            uint32_t bits = 0;
            for (int ly = 0; ly < 4; ly++)
                for (int lx = 0; lx < 8; lx++)
                {
                    int x = mx * 8 + lx, y = my * 4 + ly;
                    float d = (x < w && y < h) ? depth[(size_t)y * w + x] : 0.0f;
                    uint32_t result = 0;
                    if (d != 1.0f)
                    {
                        float tu = ((float)x + 0.5f) / (float)w, tv = ((float)y + 0.5f) / (float)h;
                        vec3  world_pos  = world_position_from_depth(tu, tv, d, ubo.view_proj_inverse);
                        vec3  normal     = octohedral_to_direction(g2.fetch(x, y, 0), g2.fetch(x, y, 1));
                        vec3  ray_origin = world_pos + normal * bias;
                        float r0 = sample_blue_noise(x, y, (int)num_frames, 0, bn);
                        float r1 = sample_blue_noise(x, y, (int)num_frames, 1, bn);
                        vec3  Wi;
                        float t_max, attenuation;
                        fetch_light_properties_shadow(ubo.light, world_pos, normal, r0, r1, &Wi, &t_max, &attenuation);
                        if (attenuation > 0.0f)
                        {
                            rays++;
                            result = scene.any_hit(ray_origin, Wi, 0.01f, t_max) ? 0u : 1u;
                        }
                    }
                    bits |= result << (ly * 8 + lx);
                }
            mask[(size_t)my * mw + mx] = bits;
        }
    if (rays_out) *rays_out = rays;
}

// Same ray generation, but emits the rays instead of tracing them (for CPU replay / debugging).
// rays: [w*h][8] = origin xyz, t_max, dir xyz, valid(1/0)
void orc_shadows_gen_rays(const void* ubo_, int w, int h, const float* depth, const uint16_t* gb2, const uint8_t* sobol,
                          const uint8_t* scrambling_ranking, float bias, uint32_t num_frames, float* rays)
{
    const UBO& ubo = *(const UBO*)ubo_;
    BlueNoise  bn { sobol, scrambling_ranking };
    ImgH<4>    g2 { gb2, w, h };
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            float* r = rays + ((size_t)y * w + x) * 8;
            for (int i = 0; i < 8; i++) r[i] = 0.0f;
            float d = depth[(size_t)y * w + x];
            if (d == 1.0f) continue;
            float tu = ((float)x + 0.5f) / (float)w, tv = ((float)y + 0.5f) / (float)h;
            vec3  world_pos  = world_position_from_depth(tu, tv, d, ubo.view_proj_inverse);
            vec3  normal     = octohedral_to_direction(g2.fetch(x, y, 0), g2.fetch(x, y, 1));
            vec3  ray_origin = world_pos + normal * bias;
            float r0 = sample_blue_noise(x, y, (int)num_frames, 0, bn);
            float r1 = sample_blue_noise(x, y, (int)num_frames, 1, bn);
            vec3  Wi;
            float t_max, attenuation;
            fetch_light_properties_shadow(ubo.light, world_pos, normal, r0, r1, &Wi, &t_max, &attenuation);
            r[0] = ray_origin.x; r[1] = ray_origin.y; r[2] = ray_origin.z; r[3] = t_max;
            r[4] = Wi.x; r[5] = Wi.y; r[6] = Wi.z; r[7] = attenuation > 0.0f ? 1.0f : 0.0f;
        }
}

static inline float unpack_hit(const ImgU& m, int x, int y, uint32_t oob)
{
    // populate_cache + unpack_shadow_hit_value (shadows_denoise_reprojection.comp:114-153):
    // mask texel (x>>3, y>>2) via floor division, bit (y&3)*8+(x&7).
    int mx = x >= 0 ? x >> 3 : -((-x + 7) >> 3);
    int my = y >= 0 ? y >> 2 : -((-y + 3) >> 2);
    uint32_t word = m.fetch(mx, my, oob);
    int bx = x - mx * 8, by = y - my * 4;
    return (float)((word >> (by * 8 + bx)) & 1u);
}

// S3.  Images at pass resolution w x h:
//   out_vis_var RG16F, out_moments RGBA16F (m1, m2, history length, 0)
//   hist_vis_var RG16F (previous à-trous feedback image), hist_moments RGBA16F
//   tile_class: [ceil(h/8)][ceil(w/8)] uint8: 1 = needs à-trous, 0 = fully shadowed / sky tile
void orc_shadows_temporal(const void* ubo_, int w, int h, const uint32_t* mask, const float* depth, const uint16_t* gb2,
                          const uint16_t* gb3, const float* prev_depth, const uint16_t* prev_gb2, const uint16_t* prev_gb3,
                          const uint16_t* hist_vis_var, const uint16_t* hist_moments, float alpha, float moments_alpha,
                          uint16_t* out_vis_var, uint16_t* out_moments, uint8_t* tile_class)
{
    const UBO& ubo = *(const UBO*)ubo_;
    ImgU       m { mask, ceil_div(w, 8), ceil_div(h, 4) };
    ImgH<4>    g2 { gb2, w, h }, g3 { gb3, w, h }, pg2 { prev_gb2, w, h }, pg3 { prev_gb3, w, h };
    ImgF       pd { prev_depth, w, h };
    ImgH<2>    hv { hist_vis_var, w, h };
    ImgH<4>    hm { hist_moments, w, h };
    ImgHW<2>   ov { out_vis_var, w, h };
    ImgHW<4>   om { out_moments, w, h };
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
                    // 17x17 box mean of the visibility bits (:157-190).  Sum of 0/1 values: exact.
                    float sum = 0.0f;
                    for (int yy = -8; yy <= 8; yy++)
                        for (int xx = -8; xx <= 8; xx++) sum += unpack_hit(m, x + xx, y + yy, 0u);
                    float mean = sum / 289.0f;

                    // no bounds check in the shader (:196-293): a thread of a ragged edge group reads depth 0 / G-buffer 0
                    // (pinned out-of-image fetch), runs the whole body — its image stores are dropped, but its result
                    // still votes in g_should_denoise (verified against the reference shader at ragged sizes)
                    float d = (x < w && y < h) ? depth[(size_t)y * w + x] : 0.0f;
                    float visibility = 0.0f, out_v = 0.0f, out_var = 0.0f, mom0 = 0.0f, mom1 = 0.0f, history_length = 0.0f;
                    if (d != 1.0f)
                    {
                        visibility = unpack_hit(m, x, y, 0u);
                        float history_visibility, history_moments[2];
                        ReprojectIn in;
                        in.x = x; in.y = y; in.depth = d;
                        in.view_proj_inverse = &ubo.view_proj_inverse;
                        in.gb2 = g2; in.gb3 = g3; in.pgb2 = pg2; in.pgb3 = pg3; in.pdepth = pd;
                        in.w = w; in.h = h;
                        bool success = reproject<true, true, false, 2>(in, hv, &hm, nullptr, &history_visibility, history_moments, &history_length);
                        history_length = fmin2(32.0f, success ? history_length + 1.0f : 1.0f);
                        if (success)
                        {
                            float spatial_variance = mean;
                            spatial_variance       = fmax2(spatial_variance - mean * mean, 0.0f);
                            const float sd   = std::sqrt(spatial_variance);
                            const float nmin = mean - 0.5f * sd;
                            const float nmax = mean + 0.5f * sd;
                            history_visibility = clampf(history_visibility, nmin, nmax);
                        }
                        const float a  = success ? fmax2(alpha, 1.0f / history_length) : 1.0f;
                        const float am = success ? fmax2(moments_alpha, 1.0f / history_length) : 1.0f;
                        mom0 = visibility;
                        mom1 = mom0 * mom0;
                        mom0 = mixf(history_moments[0], mom0, am);
                        mom1 = mixf(history_moments[1], mom1, am);
                        out_var = fmax2(0.0f, mom1 - mom0 * mom0);
                        out_v   = mixf(history_visibility, visibility, a);
                    }
                    om.store(x, y, 0, mom0); om.store(x, y, 1, mom1); om.store(x, y, 2, history_length); om.store(x, y, 3, 0.0f);
                    ov.store(x, y, 0, out_v); ov.store(x, y, 1, out_var);
                    if (d != 1.0f && out_v > 0.0f) should_denoise = true;
                }
            tile_class[(size_t)ty * tw + tx] = should_denoise ? 1 : 0;
        }
}

// S4+S5: one à-trous iteration.  in/out RG16F at pass res.
void orc_shadows_atrous(int w, int h, const uint16_t* in_vis_var, const uint16_t* gb2, const uint16_t* gb3,
                        const uint8_t* tile_class, int radius, int step_size, float phi_visibility, float phi_normal,
                        float sigma_depth, float power, uint16_t* out_vis_var)
{
    ImgH<2>   in { in_vis_var, w, h };
    ImgH<4>   g2 { gb2, w, h }, g3 { gb3, w, h };
    ImgHW<2>  out { out_vis_var, w, h };
    const int tw = ceil_div(w, 8), th = ceil_div(h, 8);
    const float kernel_weights[3] = { 1.0f, 2.0f / 3.0f, 1.0f / 6.0f };
#pragma omp parallel for schedule(dynamic, 2)
    for (int ty = 0; ty < th; ty++)
        for (int tx = 0; tx < tw; tx++)
        {
            const bool denoise = tile_class[(size_t)ty * tw + tx] != 0;
            for (int ly = 0; ly < 8; ly++)
                for (int lx = 0; lx < 8; lx++)
                {
                    int x = tx * 8 + lx, y = ty * 8 + ly;
                    if (!denoise) { out.store(x, y, 0, 0.0f); out.store(x, y, 1, 0.0f); continue; }
                    const float cv = in.fetch(x, y, 0), cvar = in.fetch(x, y, 1);
                    // compute_variance_center (:65-88)
                    const float k2[2][2] = { { 1.0f / 4.0f, 1.0f / 8.0f }, { 1.0f / 8.0f, 1.0f / 16.0f } };
                    float var = 0.0f;
                    for (int yy = -1; yy <= 1; yy++)
                        for (int xx = -1; xx <= 1; xx++) var += in.fetch(x + xx, y + yy, 1) * k2[xx < 0 ? -xx : xx][yy < 0 ? -yy : yy];
                    vec3  current_normal = octohedral_to_direction(g2.fetch(x, y, 0), g2.fetch(x, y, 1));
                    float center_depth   = g3.fetch(x, y, 3);
                    if (center_depth < 0.0f) { out.store(x, y, 0, cv); out.store(x, y, 1, cvar); continue; }
                    const float phi_v = phi_visibility * std::sqrt(fmax2(0.0f, 1e-10f + var));
                    float sum_w = 1.0f, sum_v = cv, sum_var = cvar;
                    for (int yy = -radius; yy <= radius; yy++)
                        for (int xx = -radius; xx <= radius; xx++)
                        {
                            const int  px = x + xx * step_size, py = y + yy * step_size;
                            const bool inside = px >= 0 && py >= 0 && px < w && py < h;
                            const float kernel = kernel_weights[xx < 0 ? -xx : xx] * kernel_weights[yy < 0 ? -yy : yy];
                            if (inside && (xx != 0 || yy != 0))
                            {
                                const float sv = in.fetch(px, py, 0), svar = in.fetch(px, py, 1);
                                vec3  sample_normal = octohedral_to_direction(g2.fetch(px, py, 0), g2.fetch(px, py, 1));
                                float sample_depth  = g3.fetch(px, py, 3);
                                // edge_stopping.glsl:31-62
                                const float wZ = det_exp(-std::fabs(center_depth - sample_depth) / sigma_depth);
                                const float wN = det_pow_auto(clampf(dot(current_normal, sample_normal), 0.0f, 1.0f), phi_normal);
                                const float wL = std::fabs(cv - sv) / phi_v;
                                const float wgt = det_exp((0.0f - fmax2(wL, 0.0f)) - fmax2(wZ, 0.0f)) * wN;
                                const float wv  = wgt * kernel;
                                sum_w += wv;
                                sum_v += wv * sv;
                                sum_var += (wv * wv) * svar;
                            }
                        }
                    float ov = sum_v / sum_w, ovar = sum_var / (sum_w * sum_w);
                    if (power != 0.0f) ov = det_pow_auto(ov, power);
                    out.store(x, y, 0, ov); out.store(x, y, 1, ovar);
                }
        }
}

// nearest-filtered textureLod with clamp-to-edge (all pass samplers are nearest: g_buffer.cpp:328-340,
// ray_traced_shadows.cpp:363,394,482...).
static inline void nearest_xy(float u, float v, int w, int h, int* x, int* y)
{
    int ix = (int)std::floor(u * (float)w), iy = (int)std::floor(v * (float)h);
    *x = ix < 0 ? 0 : (ix > w - 1 ? w - 1 : ix);
    *y = iy < 0 ? 0 : (iy > h - 1 ? h - 1 : iy);
}

// S6 / A5 / R6 upsample (shadows_upsample.comp:62-109, ao_upsample.comp:63-112, reflections_upsample.comp).
//   channels = 1: low-res input is channel 0 of an fp16 image with in_channels channels, output R16F
//   channels = 4: RGBA16F in/out (reflections)
//   sky_value: value written for sky pixels (0 shadows/reflections, 1 AO); power: 0 = none (AO: 1.2)
void orc_upsample(int W, int H, int w, int h, const uint16_t* gb2_full, const uint16_t* gb3_full, const uint16_t* gb2_mip,
                  const uint16_t* gb3_mip, const uint16_t* in_lowres, int in_channels, int channels, float sky_value, float power,
                  uint16_t* out_full)
{
    ImgH<4> G2 { gb2_full, W, H }, G3 { gb3_full, W, H }, g2 { gb2_mip, w, h }, g3 { gb3_mip, w, h };
    const float kx[4] = { 0.0f, 1.0f, -1.0f, 0.0f }, ky[4] = { 1.0f, 0.0f, 0.0f, -1.0f };
    const float tsx = 1.0f / (float)w, tsy = 1.0f / (float)h;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
        {
            uint16_t* o = out_full + ((size_t)y * W + x) * channels;
            float hi_depth = G3.fetch(x, y, 3);
            if (hi_depth == -1.0f)
            {
                for (int c = 0; c < channels; c++) o[c] = f32_to_f16(sky_value);
                continue;
            }
            float tu = ((float)x + 0.5f) / (float)W, tv = ((float)y + 0.5f) / (float)H;
            vec3  hi_normal = octohedral_to_direction(G2.fetch(x, y, 0), G2.fetch(x, y, 1));
            float up[4] = { 0, 0, 0, 0 }, total_w = 0.0f;
            for (int i = 0; i < 4; i++)
            {
                float cu = tu + kx[i] * tsx, cv = tv + ky[i] * tsy;
                int   sx, sy;
                nearest_xy(cu, cv, w, h, &sx, &sy);
                float coarse_depth = g3.fetch(sx, sy, 3);
                if (coarse_depth == -1.0f) continue;
                vec3 coarse_normal = octohedral_to_direction(g2.fetch(sx, sy, 0), g2.fetch(sx, sy, 1));
                // compute_edge_stopping_weight with NORMAL weight only (wL = 1.0)
                const float wZ  = det_exp(-std::fabs(hi_depth - coarse_depth) / 1.0f);
                const float wN  = det_pow_auto(clampf(dot(hi_normal, coarse_normal), 0.0f, 1.0f), 32.0f);
                const float wgt = det_exp((0.0f - 1.0f) - fmax2(wZ, 0.0f)) * wN;
                for (int c = 0; c < channels; c++)
                    up[c] += f16_to_f32(in_lowres[((size_t)sy * w + sx) * in_channels + c]) * wgt;
                total_w += wgt;
            }
            for (int c = 0; c < channels; c++)
            {
                float r = up[c] / fmax2(total_w, 0.00000001f);
                if (power != 0.0f) r = det_pow_auto(r, power);
                o[c] = f32_to_f16(r);
            }
        }
}

} // extern "C"
