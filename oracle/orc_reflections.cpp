// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).  Pinned against the reference's own shaders: tests/test_ref_shaders.py.
// Restatement of the ray-traced reflections pass:
//   R1 reflections/reflections_ray_trace.rgen:119-171 (+ importance_sample_ggx :78-105),
//      .rchit:117-150 (+ indirect_lighting :87-111), .rmiss:26-30
//   R3 reflections/reflections_denoise_reprojection.comp:174-289 (+ clip_aabb :111-129,
//      neighborhood_standard_deviation :133-157, compute_max_accumulated_frame :162-168)
//   R4 reflections_denoise_copy_tiles.comp:34-38 (folded into R5 through the tile class)
//   R5 reflections/reflections_denoise_atrous.comp:94-181
//   R6 reflections_upsample.comp:62-109 -> orc_upsample(channels = 4)
// NB ray_traced_reflections.cpp:962-991 clears m_first_frame inside clear_images(), i.e. BEFORE
// ray_trace() evaluates `sample_gi && !m_first_frame` (:1017-1018): the flags are therefore never
// forced off; the oracle takes them as given.
#include <cstdlib>
#include "orc_api.h"
#include "orc_reproject.h"
#include "orc_shading.h"

using namespace orc;

namespace orc {

static inline vec3 reflect3(vec3 I, vec3 N) { return I - N * (2.0f * dot(N, I)); }

// reflections_ray_trace.rgen:78-105 (the PDF it also computes is unused by the caller)
static inline vec3 importance_sample_ggx(float Ex, float Ey, vec3 N, float roughness)
{
    float a = roughness * roughness, m2 = a * a;
    float phi = 2.0f * ORC_M_PI * Ex;
    float cos_theta = std::sqrt((1.0f - Ey) / (1.0f + (m2 - 1.0f) * Ey));
    float sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);
    float s, c;
    det_sincos(phi, &s, &c);
    vec3 H  = v3(c * sin_theta, s * sin_theta, cos_theta);
    vec3 up = std::fabs(N.z) < 0.999f ? v3(0, 0, 1) : v3(1, 0, 0);
    vec3 tangent   = normalize(cross(up, N));
    vec3 bitangent = cross(N, tangent);
    vec3 sv        = (tangent * H.x + bitangent * H.y) + N * H.z;
    return normalize(sv);
}

struct EnvMaps
{
    CubeH           sky;
    const uint16_t* prefiltered; // levels of [6][s][s][4], s = size >> level
    int             pre_size, pre_levels;
    const uint16_t* lut;         // [n][n][2]
    int             lut_size;
    inline vec3 prefiltered_fetch(vec3 dir, float lod) const
    {
        int level = (int)std::floor(lod + 0.5f);
        level     = level < 0 ? 0 : (level > pre_levels - 1 ? pre_levels - 1 : level);
        size_t off = 0;
        for (int l = 0; l < level; l++) off += (size_t)6 * (pre_size >> l) * (pre_size >> l) * 4;
        CubeH c { prefiltered + off, pre_size >> level };
        return c.fetch(dir);
    }
    inline vec2 lut_fetch(float u, float v) const
    {
        int ix = (int)std::floor(u * (float)lut_size), iy = (int)std::floor(v * (float)lut_size);
        ix = ix < 0 ? 0 : (ix > lut_size - 1 ? lut_size - 1 : ix);
        iy = iy < 0 ? 0 : (iy > lut_size - 1 ? lut_size - 1 : iy);
        const uint16_t* q = lut + ((size_t)iy * lut_size + ix) * 2;
        return vec2 { f16_to_f32(q[0]), f16_to_f32(q[1]) };
    }
};

} // namespace orc

extern "C" {

// R1.  out: RGBA16F (rgb = min(colour, 0.7), a = ray length or -1).
void orc_reflections_ray_trace(const void* scene_, const void* ubo_, const void* ddgi_, int w, int h, const float* depth, const uint16_t* gb2,
                               const uint16_t* gb3, const uint8_t* sobol, const uint8_t* scrambling_ranking, const orc_refl_trace_params* prm,
                               const uint16_t* sky, int sky_size, const uint16_t* prefiltered, int pre_size, int pre_levels, const uint16_t* lut,
                               int lut_size, const uint16_t* irradiance, const uint16_t* depth_atlas, uint16_t* out, uint64_t* rays_out)
{
    const Scene&        scene = *(const Scene*)scene_;
    const UBO&          ubo   = *(const UBO*)ubo_;
    const DDGIUniforms& d     = *(const DDGIUniforms*)ddgi_;
    BlueNoise           bn { sobol, scrambling_ranking };
    EnvMaps             env { CubeH { sky, sky_size }, prefiltered, pre_size, pre_levels, lut, lut_size };
    ImgH<4>             g2 { gb2, w, h }, g3 { gb3, w, h };
    uint64_t            rays = 0;
#pragma omp parallel for schedule(dynamic, 2) reduction(+ : rays)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            uint16_t*   o  = out + ((size_t)y * w + x) * 4;
            const float dp = depth[(size_t)y * w + x];
            if (dp == 1.0f) { o[0] = o[1] = o[2] = 0; o[3] = f32_to_f16(-1.0f); continue; }
            const float roughness = g3.fetch(x, y, 0);
            const float tu = ((float)x + 0.5f) / (float)w, tv = ((float)y + 0.5f) / (float)h;
            const vec3  P  = world_position_from_depth(tu, tv, dp, ubo.view_proj_inverse);
            const vec3  N  = octohedral_to_direction(g2.fetch(x, y, 0), g2.fetch(x, y, 1));
            const vec3  Wo = normalize(v3(ubo.cam_pos[0], ubo.cam_pos[1], ubo.cam_pos[2]) - P);
            const vec3  ray_origin = P + N * prm->bias;
            vec3  color = v3(0, 0, 0);
            float ray_length = -1.0f;
            bool  trace = false;
            vec3  dir = v3(0, 0, 0);
            if (roughness < 0.05f) { dir = reflect3(-Wo, N); trace = true; }
            else if (roughness > 0.75f && prm->approximate_with_ddgi == 1)
            {
                vec3 R = reflect3(-Wo, N);
                color  = prm->rough_ddgi_intensity * sample_irradiance(d, P, R, Wo, irradiance, depth_atlas);
            }
            else
            {
                float r0 = sample_blue_noise(x, y, (int)prm->num_frames, 0, bn) * prm->trim;
                float r1 = sample_blue_noise(x, y, (int)prm->num_frames, 1, bn) * prm->trim;
                vec3  Wh = importance_sample_ggx(r0, r1, N, roughness);
                dir      = reflect3(-Wo, Wh);
                trace    = true;
            }
            if (trace)
            {
                rays++;
                Hit hit = scene.closest_hit(ray_origin, dir, 0.001f, 10000.0f);
                if (hit.prim < 0) { color = env.sky.fetch(dir); ray_length = -1.0f; } // rmiss
                else
                {
                    SurfaceHit sh = surface_at(scene, hit);
                    const vec3 hWo = -dir;
                    const vec3 F0  = mix3(v3(0.04f, 0.04f, 0.04f), sh.albedo, sh.metallic);
                    const vec3 c_diffuse = mix3(sh.albedo * (v3(1.0f, 1.0f, 1.0f) - F0), v3(0, 0, 0), sh.metallic);
                    vec3 Lo = direct_lighting(scene, ubo.light, hWo, sh.N, sh.P, F0, c_diffuse, sh.roughness, v3(1.0f, 1.0f, 1.0f), false, 0, 0, nullptr, &rays);
                    if (prm->sample_gi == 1)
                    {
                        const vec3 R  = reflect3(-hWo, sh.N);
                        float ndv     = fmax2(dot(sh.N, hWo), 0.0f);
                        vec3  F       = fresnel_schlick_roughness(ndv, F0, sh.roughness);
                        vec3  kD      = (v3(1.0f, 1.0f, 1.0f) - F) * (1.0f - sh.metallic);
                        vec3  pre     = env.prefiltered_fetch(R, sh.roughness * 4.0f);
                        vec2  brdf    = env.lut_fetch(ndv, sh.roughness);
                        vec3  specular = (pre * (F * brdf.x + v3(brdf.y, brdf.y, brdf.y))) * prm->ibl_indirect_specular_intensity;
                        vec3  diffuse  = (prm->gi_intensity * c_diffuse) * sample_irradiance(d, sh.P, sh.N, hWo, irradiance, depth_atlas);
                        Lo = Lo + (kD * diffuse + specular);
                    }
                    color      = Lo;
                    ray_length = 0.001f + hit.t;
                }
            }
            o[0] = f32_to_f16(fmin2(color.x, 0.7f)); o[1] = f32_to_f16(fmin2(color.y, 0.7f)); o[2] = f32_to_f16(fmin2(color.z, 0.7f));
            o[3] = f32_to_f16(ray_length);
        }
    if (rays_out) *rays_out = rays;
}

// R3.  input RGBA16F (R1 output); history colour RGBA16F, history moments RGBA16F.
// tile_class: 1 = a-trous, 0 = copy tile.
void orc_reflections_temporal(const void* ubo_, int w, int h, const uint16_t* input, const float* depth, const uint16_t* gb2, const uint16_t* gb3,
                              const float* prev_depth, const uint16_t* prev_gb2, const uint16_t* prev_gb3, const uint16_t* hist_color,
                              const uint16_t* hist_moments, const float* camera_delta, float alpha, float moments_alpha, int approximate_with_ddgi,
                              uint16_t* out_color, uint16_t* out_moments, uint8_t* tile_class)
{
    const UBO& ubo = *(const UBO*)ubo_;
    ImgH<4>    in { input, w, h }, g2 { gb2, w, h }, g3 { gb3, w, h }, pg2 { prev_gb2, w, h }, pg3 { prev_gb3, w, h }, hc { hist_color, w, h }, hm { hist_moments, w, h };
    ImgF       pd { prev_depth, w, h };
    ImgHW<4>   oc { out_color, w, h }, om { out_moments, w, h };
    const int  tw = ceil_div(w, 8), th = ceil_div(h, 8);
    const bool moving = length(v3(camera_delta[0], camera_delta[1], camera_delta[2])) > 0.0f;
#pragma omp parallel for schedule(dynamic, 2)
    for (int ty = 0; ty < th; ty++)
        for (int tx = 0; tx < tw; tx++)
        {
            bool should_denoise = false;
            for (int ly = 0; ly < 8; ly++)
                for (int lx = 0; lx < 8; lx++)
                {
                    const int x = tx * 8 + lx, y = ty * 8 + ly;
                    if (x >= w || y >= h) continue; // edge threads read roughness 0 (< 0.05): they can never vote for denoising, stores dropped
                    const float d = depth[(size_t)y * w + x];
                    const float roughness = g3.fetch(x, y, 0);
                    float orad[4] = { 0, 0, 0, 0 }, omom[4] = { 0, 0, 0, 0 };
                    if (d != 1.0f)
                    {
                        const vec3  color = v3(in.fetch(x, y, 0), in.fetch(x, y, 1), in.fetch(x, y, 2));
                        const float ray_length = in.fetch(x, y, 3);
                        float hcol[3], hmom[2], hlen;
                        ReprojectIn ri;
                        ri.x = x; ri.y = y; ri.depth = d;
                        ri.cam_pos = v3(ubo.cam_pos[0], ubo.cam_pos[1], ubo.cam_pos[2]);
                        ri.prev_view_proj = &ubo.prev_view_proj; ri.ray_length = ray_length;
                        ri.view_proj_inverse = &ubo.view_proj_inverse;
                        ri.gb2 = g2; ri.gb3 = g3; ri.pgb2 = pg2; ri.pgb3 = pg3; ri.pdepth = pd; ri.w = w; ri.h = h;
                        bool success = reproject<false, true, true, 4>(ri, hc, &hm, nullptr, hcol, hmom, &hlen);
                        hlen = fmin2(32.0f, success ? hlen + 1.0f : 1.0f);
                        vec3 history_color = v3(hcol[0], hcol[1], hcol[2]);
                        if (success)
                        {
                            // neighborhood_standard_deviation: dx outer, dy inner, fp32 running sums
                            vec3 m1 = v3(0, 0, 0), m2 = v3(0, 0, 0);
                            for (int dx = -8; dx <= 8; dx++)
                                for (int dy = -8; dy <= 8; dy++)
                                {
                                    vec3 s = v3(in.fetch(x + dx, y + dy, 0), in.fetch(x + dx, y + dy, 1), in.fetch(x + dx, y + dy, 2));
                                    m1 = m1 + s;
                                    m2 = m2 + s * s;
                                }
                            const float wgt = 289.0f;
                            vec3 mean = m1 / wgt;
                            vec3 var  = (m2 / wgt) - (mean * mean);
                            vec3 sd   = v3(std::sqrt(fmax2(var.x, 0.0f)), std::sqrt(fmax2(var.y, 0.0f)), std::sqrt(fmax2(var.z, 0.0f)));
                            vec3 amin = mean - sd, amax = mean + sd;
                            // clip_aabb
                            vec3 center = 0.5f * (amax + amin);
                            vec3 extent = 0.5f * (amax - amin) + v3(0.001f, 0.001f, 0.001f);
                            vec3 cv  = history_color - center;
                            vec3 cvc = v3(std::fabs(cv.x / extent.x), std::fabs(cv.y / extent.y), std::fabs(cv.z / extent.z));
                            float mx = fmax2(fmax2(cvc.x, cvc.y), cvc.z);
                            if (mx > 1.0f) history_color = center + cv / mx;
                        }
                        const float max_acc = moving ? 8.0f : hlen;
                        const float a  = success ? fmax2(alpha, 1.0f / max_acc) : 1.0f;
                        const float am = success ? fmax2(moments_alpha, 1.0f / max_acc) : 1.0f;
                        float mo0 = luminance(color), mo1 = mo0 * mo0;
                        mo0 = mixf(hmom[0], mo0, am);
                        mo1 = mixf(hmom[1], mo1, am);
                        const float variance = fmax2(0.0f, mo1 - mo0 * mo0);
                        vec3 acc = mix3(history_color, color, a);
                        omom[0] = mo0; omom[1] = mo1; omom[2] = hlen; omom[3] = 0.0f;
                        orad[0] = acc.x; orad[1] = acc.y; orad[2] = acc.z; orad[3] = variance;
                    }
                    for (int c = 0; c < 4; c++) { om.store(x, y, c, omom[c]); oc.store(x, y, c, orad[c]); }
                    if (d != 1.0f && roughness >= 0.05f)
                    {
                        if (approximate_with_ddgi == 1) { if (roughness <= 0.75f) should_denoise = true; }
                        else should_denoise = true;
                    }
                }
            tile_class[(size_t)ty * tw + tx] = should_denoise ? 1 : 0;
        }
}

// R4 + R5.
void orc_reflections_atrous(int w, int h, const uint16_t* in_color, const float* depth, const uint16_t* gb2, const uint16_t* gb3,
                            const uint8_t* tile_class, int radius, int step_size, float phi_color, float phi_normal, float sigma_depth,
                            int approximate_with_ddgi, uint16_t* out_color)
{
    ImgH<4>   in { in_color, w, h }, g2 { gb2, w, h }, g3 { gb3, w, h };
    ImgHW<4>  out { out_color, w, h };
    const int tw = ceil_div(w, 8), th = ceil_div(h, 8);
    const float kw[3] = { 1.0f, 2.0f / 3.0f, 1.0f / 6.0f };
#pragma omp parallel for schedule(dynamic, 2)
    for (int ty = 0; ty < th; ty++)
        for (int tx = 0; tx < tw; tx++)
        {
            const bool denoise = tile_class[(size_t)ty * tw + tx] != 0;
            for (int ly = 0; ly < 8; ly++)
                for (int lx = 0; lx < 8; lx++)
                {
                    const int x = tx * 8 + lx, y = ty * 8 + ly;
                    if (x >= w || y >= h) continue;
                    float cc[4] = { in.fetch(x, y, 0), in.fetch(x, y, 1), in.fetch(x, y, 2), in.fetch(x, y, 3) };
                    if (!denoise) { for (int c = 0; c < 4; c++) out.store(x, y, c, cc[c]); continue; }
                    const float center_luma = luminance(v3(cc[0], cc[1], cc[2]));
                    const float k2[2][2] = { { 1.0f / 4.0f, 1.0f / 8.0f }, { 1.0f / 8.0f, 1.0f / 16.0f } };
                    float var = 0.0f;
                    for (int yy = -1; yy <= 1; yy++)
                        for (int xx = -1; xx <= 1; xx++) var += in.fetch(x + xx, y + yy, 3) * k2[xx < 0 ? -xx : xx][yy < 0 ? -yy : yy];
                    const vec3  cn = octohedral_to_direction(g2.fetch(x, y, 0), g2.fetch(x, y, 1));
                    const float center_depth = g3.fetch(x, y, 3);
                    const float d = depth[(size_t)y * w + x], roughness = g3.fetch(x, y, 0);
                    if (d == 1.0f) { for (int c = 0; c < 4; c++) out.store(x, y, c, 0.0f); continue; }
                    if (roughness < 0.05f || (approximate_with_ddgi == 1 && roughness > 0.75f)) { for (int c = 0; c < 4; c++) out.store(x, y, c, cc[c]); continue; }
                    const float phi_c = phi_color * std::sqrt(fmax2(0.0f, 1e-10f + var));
                    float sum_w = 1.0f, sum[4] = { cc[0], cc[1], cc[2], cc[3] };
                    for (int yy = -radius; yy <= radius; yy++)
                        for (int xx = -radius; xx <= radius; xx++)
                        {
                            const int  px = x + xx * step_size, py = y + yy * step_size;
                            const bool inside = px >= 0 && py >= 0 && px < w && py < h;
                            const float kernel = kw[xx < 0 ? -xx : xx] * kw[yy < 0 ? -yy : yy];
                            if (inside && (xx != 0 || yy != 0))
                            {
                                float sc[4] = { in.fetch(px, py, 0), in.fetch(px, py, 1), in.fetch(px, py, 2), in.fetch(px, py, 3) };
                                const float sl = luminance(v3(sc[0], sc[1], sc[2]));
                                const vec3  sn = octohedral_to_direction(g2.fetch(px, py, 0), g2.fetch(px, py, 1));
                                const float sdp = g3.fetch(px, py, 3);
                                const float wZ = det_exp(-std::fabs(center_depth - sdp) / sigma_depth);
                                const float wN = det_pow_auto(clampf(dot(cn, sn), 0.0f, 1.0f), phi_normal);
                                const float wL = std::fabs(center_luma - sl) / phi_c;
                                const float wgt = det_exp((0.0f - fmax2(wL, 0.0f)) - fmax2(wZ, 0.0f)) * wN;
                                const float wc  = wgt * kernel;
                                sum_w += wc;
                                sum[0] += wc * sc[0]; sum[1] += wc * sc[1]; sum[2] += wc * sc[2];
                                sum[3] += (wc * wc) * sc[3];
                            }
                        }
                    out.store(x, y, 0, sum[0] / sum_w); out.store(x, y, 1, sum[1] / sum_w); out.store(x, y, 2, sum[2] / sum_w);
                    out.store(x, y, 3, sum[3] / (sum_w * sum_w));
                }
        }
}

} // extern "C"
