// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).  Pinned against the reference's own shaders: tests/test_ref_shaders.py.
// Restatement of the ground-truth path tracer:
//   ground_truth/ground_truth_path_trace.rgen:52-112  primary ray with RNG jitter, running mean in RGBA16F
//   ground_truth/ground_truth_path_trace.rchit:114-142 direct lighting at the first hit: punctual light with
//       SOFT_SHADOWS + sky light (lighting.glsl:117-196 with SOFT_SHADOWS | RAY_THROUGHPUT | SAMPLE_SKY_LIGHT)
//   ground_truth/ground_truth_path_trace.rmiss:24-32  sky cubemap
// indirect_lighting (rchit:67-108) only consumes random numbers and returns p_IndirectPayload.L = 0 because the
// recursive traceRayEXT is commented out (rchit:95-105); nothing reads the RNG afterwards, so it is omitted here.
// Pinned: function arguments are evaluated left to right (rng1 = the two floats after the pixel jitter, rng2 = the next
// two); the sky cubemap is fetched nearest.
#include "orc_api.h"
#include "orc_shading.h"

using namespace orc;

namespace orc {
void fetch_light_properties_shadow(const Light& light, vec3 P, vec3 N, float rx, float ry, vec3* Wi, float* t_max, float* attenuation); // orc_shadows.cpp
}

extern "C" {

// prev / cur: [h][w][4] fp16 bit patterns.  rows [y0, y1) are rendered.
void orc_ground_truth_render(const void* scene_, const void* ubo_, const uint16_t* sky, int sky_size, int w, int h, int y0, int y1, uint32_t num_frames,
                             float roughness_multiplier, const uint16_t* prev, uint16_t* cur, uint64_t* rays_out)
{
    const Scene& scene = *(const Scene*)scene_;
    const UBO&   ubo   = *(const UBO*)ubo_;
    CubeH        cube { sky, sky_size };
    uint64_t     rays = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : rays)
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < w; x++)
        {
            RNG rng = rng_init((uint32_t)x, (uint32_t)y, num_frames);
            const float jx = next_float(rng), jy = next_float(rng);
            const float px = ((float)x + 0.5f) + jx, py = ((float)y + 0.5f) + jy;
            const float tx = (px / (float)w) * 2.0f - 1.0f, ty = (py / (float)h) * 2.0f - 1.0f;
            const vec4  origin = mul(ubo.view_inverse, vec4 { 0.0f, 0.0f, 0.0f, 1.0f });
            const vec4  target = mul(ubo.proj_inverse, vec4 { tx, ty, 1.0f, 1.0f });
            const vec3  tn     = normalize(v3(target.x, target.y, target.z));
            const vec4  dir4   = mul(ubo.view_inverse, vec4 { tn.x, tn.y, tn.z, 0.0f });
            const vec3  o = v3(origin.x, origin.y, origin.z), d = v3(dir4.x, dir4.y, dir4.z);
            vec3 L;
            rays++;
            const Hit hit = scene.closest_hit(o, d, 0.001f, 10000.0f);
            if (hit.prim < 0) L = cube.fetch(d);
            else
            {
                SurfaceHit sh = surface_at(scene, hit);
                sh.N = normalize(sh.N);   // rchit:131 normalises fetch_normal()'s result once more (observable with normal maps)
                const float roughness = sh.roughness * roughness_multiplier;
                const vec3  Wo = -d;
                const vec3  F0 = mix3(v3(0.04f, 0.04f, 0.04f), sh.albedo, sh.metallic);
                const vec3  c_diffuse = mix3(sh.albedo * (v3(1.0f, 1.0f, 1.0f) - F0), v3(0, 0, 0), sh.metallic);
                const float r1x = next_float(rng), r1y = next_float(rng), r2x = next_float(rng), r2y = next_float(rng);
                const vec3  T = v3(1.0f, 1.0f, 1.0f);
                vec3        Lo = v3(0, 0, 0);
                const vec3  ray_origin = sh.P + sh.N * 0.1f;
                {
                    vec3  Wi;
                    float t_max, attenuation;
                    fetch_light_properties_shadow(ubo.light, sh.P, sh.N, r1x, r1y, &Wi, &t_max, &attenuation);
                    const vec3 Li = v3(ubo.light.data2[0], ubo.light.data2[1], ubo.light.data2[2]) * ubo.light.data0[3];
                    const vec3 Wh = normalize(Wo + Wi);
                    if (attenuation > 0.0f)
                    {
                        rays++;
                        attenuation = attenuation * (scene.any_hit(ray_origin, Wi, 0.01f, t_max) ? 0.0f : 1.0f);
                    }
                    const vec3 brdf = evaluate_uber_brdf(c_diffuse, roughness, sh.N, F0, Wo, Wh, Wi);
                    Lo = Lo + ((T * brdf) * attenuation) * Li;
                }
                {
                    const vec3 Wi = sample_cosine_lobe(sh.N, r2x, r2y);
                    vec3       Li = cube.fetch(Wi);
                    const vec3 Wh = normalize(Wo + Wi);
                    rays++;
                    Li = Li * (scene.any_hit(ray_origin, Wi, 0.01f, 10000.0f) ? 0.0f : 1.0f);
                    const vec3 brdf = evaluate_uber_brdf(c_diffuse, roughness, sh.N, F0, Wo, Wh, Wi);
                    Lo = Lo + (T * brdf) * Li;
                }
                L = Lo;
            }
            const vec3 clamped = v3(fmin2(L.x, 1.0f), fmin2(L.y, 1.0f), fmin2(L.z, 1.0f)); // RADIANCE_CLAMP_COLOR (common.glsl:19)
            vec3 out = clamped;
            const size_t o4 = ((size_t)y * w + x) * 4;
            if (num_frames != 0)
            {
                const vec3 pc = v3(f16_to_f32(prev[o4 + 0]), f16_to_f32(prev[o4 + 1]), f16_to_f32(prev[o4 + 2]));
                const float n = (float)num_frames;
                out = v3(pc.x + (clamped.x - pc.x) / n, pc.y + (clamped.y - pc.y) / n, pc.z + (clamped.z - pc.z) / n);
            }
            cur[o4 + 0] = f32_to_f16(out.x); cur[o4 + 1] = f32_to_f16(out.y); cur[o4 + 2] = f32_to_f16(out.z); cur[o4 + 3] = f32_to_f16(1.0f);
        }
    if (rays_out) *rays_out = rays;
}

} // extern "C"
