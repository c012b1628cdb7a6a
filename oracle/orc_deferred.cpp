// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).  Pinned against the reference's own shaders: tests/test_ref_shaders.py.
// Restatement of the deferred composite (SURVEY.md §8f "next" row 1): shaders/deferred.frag:177-205
// (+ evaluate_sh9_irradiance :115-141, project_onto_sh9 :96-113, indirect_lighting :151-173,
// direct_lighting of lighting.glsl:117-196 WITHOUT RAY_TRACING / SOFT_SHADOWS).  No sky test: the reference
// shades every pixel of the full-screen triangle and draws the skybox afterwards.
#include "orc_api.h"
#include "orc_shading.h"

using namespace orc;

extern "C" {

// flags: bit0 shadow, bit1 ao, bit2 reflections, bit3 gi  (ShadingPushConstants, deferred_shading.cpp:14-20)
// sh9: [9][4] floats (s_IrradianceSH, 9x1 texels).  shadow / ao: R16F-like, `*_channels` halfs per texel, channel 0 used;
// reflections / gi: RGBA16F.  out: RGBA16F.
void orc_deferred_shade(const void* ubo_, int w, int h, const uint8_t* gb1, const uint16_t* gb2, const uint16_t* gb3, const float* depth,
                        const uint16_t* shadow, int shadow_channels, const uint16_t* ao, int ao_channels, const uint16_t* reflections,
                        const uint16_t* gi, int flags, const float* sh9, const uint16_t* prefiltered, int pre_size, int pre_levels,
                        const uint16_t* lut, int lut_size, uint16_t* out);

// render_skybox (deferred_shading.cpp:734-789; skybox.vert/.frag): the cube drawn after the shading passes the depth test
// exactly where the G-buffer left depth 1.  The interpolated cube position lies on the ray through the pixel centre; the
// rasteriser's interpolation itself is not reproducible, so the lookup direction is PINNED to that ray, computed as the
// reference computes a pixel's ray elsewhere (ground_truth_path_trace.rgen:70-72).  sky: [6][S][S][4] fp16, NEAREST.
void orc_deferred_skybox(const void* ubo_, int w, int h, const float* depth, const uint16_t* sky, int sky_size, uint16_t* out)
{
    const UBO& ubo = *(const UBO*)ubo_;
    CubeH      cube { sky, sky_size };
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            const size_t i = (size_t)y * w + x;
            if (depth[i] != 1.0f) continue;
            const float tu = ((float)x + 0.5f) / (float)w, tv = ((float)y + 0.5f) / (float)h;
            const vec4  target = mul(ubo.proj_inverse, vec4 { tu * 2.0f - 1.0f, tv * 2.0f - 1.0f, 1.0f, 1.0f });
            const vec3  tn     = normalize(v3(target.x, target.y, target.z));
            const vec4  dir    = mul(ubo.view_inverse, vec4 { tn.x, tn.y, tn.z, 0.0f });
            const vec3  env    = cube.fetch(v3(dir.x, dir.y, dir.z));
            out[i * 4 + 0] = f32_to_f16(env.x); out[i * 4 + 1] = f32_to_f16(env.y); out[i * 4 + 2] = f32_to_f16(env.z); out[i * 4 + 3] = f32_to_f16(1.0f);
        }
}

void orc_deferred_shade(const void* ubo_, int w, int h, const uint8_t* gb1, const uint16_t* gb2, const uint16_t* gb3, const float* depth,
                        const uint16_t* shadow, int shadow_channels, const uint16_t* ao, int ao_channels, const uint16_t* reflections,
                        const uint16_t* gi, int flags, const float* sh9, const uint16_t* prefiltered, int pre_size, int pre_levels,
                        const uint16_t* lut, int lut_size, uint16_t* out)
{
    const UBO& ubo = *(const UBO*)ubo_;
    ImgH<4>    g2 { gb2, w, h }, g3 { gb3, w, h };
    const float Pi = 3.141592654f, CosineA0 = Pi, CosineA1 = (2.0f * Pi) / 3.0f, CosineA2 = Pi * 0.25f;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            const size_t i = (size_t)y * w + x;
            const float tu = ((float)x + 0.5f) / (float)w, tv = ((float)y + 0.5f) / (float)h;
            const vec3  albedo = v3((float)gb1[i * 4] / 255.0f, (float)gb1[i * 4 + 1] / 255.0f, (float)gb1[i * 4 + 2] / 255.0f);
            const float metallic = (float)gb1[i * 4 + 3] / 255.0f;
            const float roughness = g3.fetch(x, y, 0);
            const vec3  P = world_position_from_depth(tu, tv, depth[i], ubo.view_proj_inverse);
            const float visibility = (flags & 1) ? f16_to_f32(shadow[i * shadow_channels]) : 1.0f;
            const float aov = (flags & 2) ? f16_to_f32(ao[i * ao_channels]) : 1.0f;
            const vec3  N  = octohedral_to_direction(g2.fetch(x, y, 0), g2.fetch(x, y, 1));
            const vec3  Wo = normalize(v3(ubo.cam_pos[0], ubo.cam_pos[1], ubo.cam_pos[2]) - P);
            const vec3  F0 = mix3(v3(0.04f, 0.04f, 0.04f), albedo, metallic);
            const vec3  c_diffuse = mix3(albedo * (v3(1.0f, 1.0f, 1.0f) - F0), v3(0, 0, 0), metallic);
            vec3 Lo = v3(0, 0, 0);
            {   // direct lighting, no shadow ray
                vec3  Li, Wi, Wh;
                float t_max, attenuation;
                fetch_light_hard(ubo.light, Wo, P, N, &Li, &Wi, &Wh, &t_max, &attenuation);
                vec3 brdf = evaluate_uber_brdf(c_diffuse, roughness, N, F0, Wo, Wh, Wi);
                vec3 d    = ((v3(1.0f, 1.0f, 1.0f) * brdf) * attenuation) * Li;
                Lo = Lo + d * visibility;
            }
            {   // indirect lighting
                const vec3 I  = -Wo;
                const vec3 Rr = I - N * (2.0f * dot(N, I)); // reflect(-Wo, N)
                const float ndv = fmax2(dot(N, Wo), 0.0f);
                vec3 F  = fresnel_schlick_roughness(ndv, F0, roughness);
                vec3 kD = (v3(1.0f, 1.0f, 1.0f) - F) * (1.0f - metallic);
                vec3 irradiance;
                if (flags & 8) irradiance = v3(f16_to_f32(gi[i * 4]), f16_to_f32(gi[i * 4 + 1]), f16_to_f32(gi[i * 4 + 2]));
                else
                {
                    float c[9];
                    c[0] = 0.282095f;
                    c[1] = -0.488603f * N.y;
                    c[2] = 0.488603f * N.z;
                    c[3] = -0.488603f * N.x;
                    c[4] = 1.092548f * N.x * N.y;
                    c[5] = -1.092548f * N.y * N.z;
                    c[6] = 0.315392f * (3.0f * N.z * N.z - 1.0f);
                    c[7] = -1.092548f * N.x * N.z;
                    c[8] = 0.546274f * (N.x * N.x - N.y * N.y);
                    c[0] *= CosineA0;
                    c[1] *= CosineA1; c[2] *= CosineA1; c[3] *= CosineA1;
                    c[4] *= CosineA2; c[5] *= CosineA2; c[6] *= CosineA2; c[7] *= CosineA2; c[8] *= CosineA2;
                    vec3 col = v3(0, 0, 0);
                    for (int k = 0; k < 9; k++) col = col + v3(sh9[k * 4], sh9[k * 4 + 1], sh9[k * 4 + 2]) * c[k];
                    col = v3(fmax2(0.0f, col.x), fmax2(0.0f, col.y), fmax2(0.0f, col.z));
                    irradiance = col / Pi;
                }
                vec3 diffuse = irradiance * c_diffuse;
                vec3 pre;
                if (flags & 4) pre = v3(f16_to_f32(reflections[i * 4]), f16_to_f32(reflections[i * 4 + 1]), f16_to_f32(reflections[i * 4 + 2]));
                else
                {
                    int level = (int)std::floor(roughness * 4.0f + 0.5f);
                    level     = level < 0 ? 0 : (level > pre_levels - 1 ? pre_levels - 1 : level);
                    size_t off = 0;
                    for (int l = 0; l < level; l++) off += (size_t)6 * (pre_size >> l) * (pre_size >> l) * 4;
                    CubeH cm { prefiltered + off, pre_size >> level };
                    pre = cm.fetch(Rr);
                }
                int ix = (int)std::floor(ndv * (float)lut_size), iy = (int)std::floor(roughness * (float)lut_size);
                ix = ix < 0 ? 0 : (ix > lut_size - 1 ? lut_size - 1 : ix);
                iy = iy < 0 ? 0 : (iy > lut_size - 1 ? lut_size - 1 : iy);
                const float bx = f16_to_f32(lut[((size_t)iy * lut_size + ix) * 2]), by = f16_to_f32(lut[((size_t)iy * lut_size + ix) * 2 + 1]);
                vec3 specular = (pre * (F * bx + v3(by, by, by))) * 2.0f;
                Lo = Lo + (kD * diffuse + specular) * aov;
            }
            out[i * 4 + 0] = f32_to_f16(Lo.x); out[i * 4 + 1] = f32_to_f16(Lo.y); out[i * 4 + 2] = f32_to_f16(Lo.z); out[i * 4 + 3] = f32_to_f16(1.0f);
        }
}

} // extern "C"
