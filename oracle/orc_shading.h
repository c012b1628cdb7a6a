// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).  Pinned against the reference's own shaders: tests/test_ref_shaders.py.
// Hit shading shared by the reflection and DDGI closest-hit shaders:
//   brdf.glsl:36-142          GGX / Schlick terms, evaluate_uber_brdf
//   lighting.glsl:6-196       fetch_light_properties (non-soft) + direct_lighting (+ SAMPLE_SKY_LIGHT)
//   gi_common.glsl:10-320     DDGIUniforms, probe addressing, oct coding, sample_irradiance
//   random.glsl:11-56         xoroshiro64* RNG
//   scene_descriptor_set.glsl:102-220  interpolated_vertex / transform_vertex / fetch_* (constant or textured
//                             materials; instances flattened => identity model)
// Pinned where the reference leaves it to samplers / absent assets:
//   * the sky / prefiltered environment cubemaps are INPUTS ([6][S][S] RGBA16F), fetched NEAREST
//     with the Vulkan face-selection rule;
//   * DDGI atlases are sampled bilinearly (ddgi.cpp:478,499) with fp32 weights,
//     mix(mix(t00,t10,fx), mix(t01,t11,fx), fy), clamp-to-edge.
#pragma once
#include "orc_bvh.h"
#include "orc_common.h"

namespace orc {

#define ORC_EPSILON 0.0001f

// ------------------------------------------------------------------------------------------- RNG
struct RNG { uint32_t x, y; };
static inline uint32_t rng_rotl(uint32_t x, uint32_t k) { return (x << k) | (x >> (32 - k)); }
static inline uint32_t rng_next(RNG& r)
{
    uint32_t result = r.x * 0x9e3779bbu;
    r.y ^= r.x;
    r.x = rng_rotl(r.x, 26) ^ r.y ^ (r.y << 9);
    r.y = rng_rotl(r.y, 13);
    return result;
}
static inline uint32_t rng_hash(uint32_t seed)
{
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u;
    seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    seed = seed ^ (seed >> 15);
    return seed;
}
static inline RNG rng_init(uint32_t idx, uint32_t idy, uint32_t frame)
{
    RNG r;
    r.x = rng_hash((idx << 16) | idy);
    r.y = rng_hash(frame);
    rng_next(r);
    return r;
}
static inline float next_float(RNG& r) { return u2f(0x3f800000u | (rng_next(r) >> 9)) - 1.0f; }

// ------------------------------------------------------------------------------------------- env
struct CubeH // [6][S][S][4] fp16, faces +X -X +Y -Y +Z -Z
{
    const uint16_t* p;
    int             S;
    inline vec3 fetch(vec3 d) const
    {
        float ax = std::fabs(d.x), ay = std::fabs(d.y), az = std::fabs(d.z);
        int   face;
        float sc, tc, ma;
        if (ax >= ay && ax >= az) { ma = ax; if (d.x >= 0.0f) { face = 0; sc = -d.z; tc = -d.y; } else { face = 1; sc = d.z; tc = -d.y; } }
        else if (ay >= az) { ma = ay; if (d.y >= 0.0f) { face = 2; sc = d.x; tc = d.z; } else { face = 3; sc = d.x; tc = -d.z; } }
        else { ma = az; if (d.z >= 0.0f) { face = 4; sc = d.x; tc = -d.y; } else { face = 5; sc = -d.x; tc = -d.y; } }
        float s = 0.5f * (sc / ma + 1.0f), t = 0.5f * (tc / ma + 1.0f);
        int   ix = (int)std::floor(s * (float)S), iy = (int)std::floor(t * (float)S);
        ix = ix < 0 ? 0 : (ix > S - 1 ? S - 1 : ix);
        iy = iy < 0 ? 0 : (iy > S - 1 ? S - 1 : iy);
        const uint16_t* q = p + (((size_t)face * S + iy) * S + ix) * 4;
        return v3(f16_to_f32(q[0]), f16_to_f32(q[1]), f16_to_f32(q[2]));
    }
};

// ------------------------------------------------------------------------------------------- BRDF
static inline float D_ggx(float ndoth, float alpha)
{
    float a2 = alpha * alpha;
    float denom = (ndoth * ndoth) * (a2 - 1.0f) + 1.0f;
    return a2 / fmax2(ORC_EPSILON, (ORC_M_PI * denom * denom));
}
static inline float G1_schlick_ggx(float roughness, float ndotv)
{
    float k = ((roughness + 1.0f) * (roughness + 1.0f)) / 8.0f;
    return ndotv / fmax2(ORC_EPSILON, (ndotv * (1.0f - k) + k));
}
static inline float G_schlick_ggx(float ndotl, float ndotv, float roughness) { return G1_schlick_ggx(roughness, ndotl) * G1_schlick_ggx(roughness, ndotv); }
static inline vec3  F_schlick(vec3 f0, float vdoth)
{
    float p = det_powi(1.0f - vdoth, 5);
    return f0 + (v3(1.0f, 1.0f, 1.0f) - f0) * p;
}
static inline vec3 evaluate_specular_brdf(float roughness, vec3 F, float ndoth, float ndotl, float ndotv)
{
    float alpha = roughness * roughness;
    vec3  num   = (D_ggx(ndoth, alpha) * F) * G_schlick_ggx(ndotl, ndotv, roughness);
    return num / fmax2(ORC_EPSILON, (4.0f * ndotl * ndotv));
}
static inline vec3 evaluate_uber_brdf(vec3 diffuse_color, float roughness, vec3 N, vec3 F0, vec3 Wo, vec3 Wh, vec3 Wi)
{
    float NdotL = fmax2(dot(N, Wi), 0.0f), NdotV = fmax2(dot(N, Wo), 0.0f), NdotH = fmax2(dot(N, Wh), 0.0f), VdotH = fmax2(dot(Wi, Wh), 0.0f);
    vec3  F        = F_schlick(F0, VdotH);
    vec3  specular = evaluate_specular_brdf(roughness, F, NdotH, NdotL, NdotV);
    vec3  diffuse  = diffuse_color / ORC_M_PI;
    return (v3(1.0f, 1.0f, 1.0f) - F) * diffuse + specular;
}
static inline vec3 fresnel_schlick_roughness(float cos_theta, vec3 F0, float roughness)
{
    float r1 = 1.0f - roughness;
    vec3  m  = v3(fmax2(r1, F0.x), fmax2(r1, F0.y), fmax2(r1, F0.z));
    float p  = det_powi(fmax2(1.0f - cos_theta, 0.0f), 5);
    return F0 + (m - F0) * p;
}
vec3 sample_cosine_lobe(vec3 n, float rx, float ry); // orc_ao.cpp (brdf.glsl:20-32)
void make_rotation_matrix(vec3 z, vec3* x, vec3* y);   // orc_ao.cpp (brdf.glsl:8-16)

// ------------------------------------------------------------------------------------------- DDGI
struct DDGIUniforms // ddgi.cpp:14-32 == gi_common.glsl:10-28, scalar layout, 88 bytes
{
    float grid_start_position[3];
    float grid_step[3];
    int   probe_counts[3];
    float max_distance, depth_sharpness, hysteresis, normal_bias, energy_preservation;
    int   irradiance_probe_side_length, irradiance_texture_width, irradiance_texture_height;
    int   depth_probe_side_length, depth_texture_width, depth_texture_height;
    int   rays_per_probe, visibility_test;
};
static_assert(sizeof(DDGIUniforms) == 88, "DDGIUniforms layout");

static inline float sign_not_zero(float k) { return k >= 0.0f ? 1.0f : -1.0f; }
static inline vec2  gi_oct_encode(vec3 v)
{
    float l1  = (std::fabs(v.x) + std::fabs(v.y)) + std::fabs(v.z);
    float inv = 1.0f / l1;
    float rx = v.x * inv, ry = v.y * inv;
    if (v.z < 0.0f)
    {
        float nx = (1.0f - std::fabs(ry)) * sign_not_zero(rx);
        float ny = (1.0f - std::fabs(rx)) * sign_not_zero(ry);
        rx = nx; ry = ny;
    }
    return vec2 { rx, ry };
}
static inline vec3 gi_oct_decode(float ox, float oy)
{
    vec3 v = v3(ox, oy, 1.0f - std::fabs(ox) - std::fabs(oy));
    if (v.z < 0.0f)
    {
        float nx = (1.0f - std::fabs(v.y)) * sign_not_zero(v.x);
        float ny = (1.0f - std::fabs(v.x)) * sign_not_zero(v.y);
        v.x = nx; v.y = ny;
    }
    return normalize(v);
}
static inline vec3 grid_coord_to_position(const DDGIUniforms& d, int cx, int cy, int cz)
{
    return v3(d.grid_step[0] * (float)cx + d.grid_start_position[0], d.grid_step[1] * (float)cy + d.grid_start_position[1],
              d.grid_step[2] * (float)cz + d.grid_start_position[2]);
}
static inline vec3 probe_location(const DDGIUniforms& d, int index)
{
    int cx = index % d.probe_counts[0];
    int cy = (index % (d.probe_counts[0] * d.probe_counts[1])) / d.probe_counts[0];
    int cz = index / (d.probe_counts[0] * d.probe_counts[1]);
    return grid_coord_to_position(d, cx, cy, cz);
}
// gi_common.glsl:164-184
static inline vec2 texture_coord_from_direction(vec3 dir, int probe_index, int tw, int th, int side)
{
    vec2  oc = gi_oct_encode(normalize(dir));
    float zx = (oc.x + 1.0f) * 0.5f, zy = (oc.y + 1.0f) * 0.5f;
    float pwb = (float)side + 2.0f;
    float ox = (zx * (float)side) / (float)tw, oy = (zy * (float)side) / (float)th;
    int   per_row = (tw - 2) / (int)pwb;
    float tlx = std::fmod((float)probe_index, (float)per_row) * pwb + 2.0f; // mod(int,int) promoted to float
    float tly = (float)(probe_index / per_row) * pwb + 2.0f;
    return vec2 { tlx / (float)tw + ox, tly / (float)th + oy };
}
// bilinear textureLod on an fp16 atlas with C channels (returns up to 3)
template <int C>
static inline vec3 atlas_bilinear(const uint16_t* p, int w, int h, float u, float v)
{
    float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    float fx0 = std::floor(x), fy0 = std::floor(y);
    float fx = x - fx0, fy = y - fy0;
    int   x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    auto  cl = [](int a, int n) { return a < 0 ? 0 : (a > n - 1 ? n - 1 : a); };
    x0 = cl(x0, w); x1 = cl(x1, w); y0 = cl(y0, h); y1 = cl(y1, h);
    float r[3] = { 0, 0, 0 };
    for (int c = 0; c < (C < 3 ? C : 3); c++)
    {
        float t00 = f16_to_f32(p[((size_t)y0 * w + x0) * C + c]), t10 = f16_to_f32(p[((size_t)y0 * w + x1) * C + c]);
        float t01 = f16_to_f32(p[((size_t)y1 * w + x0) * C + c]), t11 = f16_to_f32(p[((size_t)y1 * w + x1) * C + c]);
        r[c] = mixf(mixf(t00, t10, fx), mixf(t01, t11, fx), fy);
    }
    return v3(r[0], r[1], r[2]);
}

// gi_common.glsl:188-320
static inline vec3 sample_irradiance(const DDGIUniforms& d, vec3 P, vec3 N, vec3 Wo, const uint16_t* irradiance, const uint16_t* depth)
{
    const vec3 gs = v3(d.grid_step[0], d.grid_step[1], d.grid_step[2]);
    const vec3 g0 = v3(d.grid_start_position[0], d.grid_start_position[1], d.grid_start_position[2]);
    auto clampi = [](int a, int lo, int hi) { return a < lo ? lo : (a > hi ? hi : a); };
    // base_grid_coord: clamp(ivec3((X - start) / step), 0, counts - 1)
    int bx = clampi((int)((P.x - g0.x) / gs.x), 0, d.probe_counts[0] - 1);
    int by = clampi((int)((P.y - g0.y) / gs.y), 0, d.probe_counts[1] - 1);
    int bz = clampi((int)((P.z - g0.z) / gs.z), 0, d.probe_counts[2] - 1);
    vec3 base_pos = grid_coord_to_position(d, bx, by, bz);
    vec3 sum_irr = v3(0, 0, 0);
    float sum_w = 0.0f;
    vec3 alpha = v3(clampf((P.x - base_pos.x) / gs.x, 0.0f, 1.0f), clampf((P.y - base_pos.y) / gs.y, 0.0f, 1.0f), clampf((P.z - base_pos.z) / gs.z, 0.0f, 1.0f));
    for (int i = 0; i < 8; ++i)
    {
        int ox = i & 1, oy = (i >> 1) & 1, oz = (i >> 2) & 1;
        int cx = clampi(bx + ox, 0, d.probe_counts[0] - 1), cy = clampi(by + oy, 0, d.probe_counts[1] - 1), cz = clampi(bz + oz, 0, d.probe_counts[2] - 1);
        int p  = cx + cy * d.probe_counts[0] + cz * d.probe_counts[0] * d.probe_counts[1];
        vec3 probe_pos = grid_coord_to_position(d, cx, cy, cz);
        vec3 probe_to_point = (P - probe_pos) + (N + 3.0f * Wo) * d.normal_bias;
        vec3 dir = normalize(-probe_to_point);
        vec3 tri = v3(mixf(1.0f - alpha.x, alpha.x, (float)ox), mixf(1.0f - alpha.y, alpha.y, (float)oy), mixf(1.0f - alpha.z, alpha.z, (float)oz));
        float weight = 1.0f;
        {
            vec3  tdp = normalize(probe_pos - P);
            float t   = fmax2(0.0001f, (dot(tdp, N) + 1.0f) * 0.5f);
            weight    = weight * (t * t + 0.2f);
        }
        if (d.visibility_test == 1)
        {
            vec2  tc   = texture_coord_from_direction(-dir, p, d.depth_texture_width, d.depth_texture_height, d.depth_probe_side_length);
            float dist = length(probe_to_point);
            vec3  temp = atlas_bilinear<2>(depth, d.depth_texture_width, d.depth_texture_height, tc.x, tc.y);
            float mean = temp.x;
            float variance = std::fabs(temp.x * temp.x - temp.y);
            float dm  = fmax2(dist - mean, 0.0f);
            float che = variance / (variance + dm * dm);
            che       = fmax2(che * che * che, 0.0f);
            weight    = weight * ((dist <= mean) ? 1.0f : che);
        }
        weight = fmax2(0.000001f, weight);
        vec2 tc = texture_coord_from_direction(normalize(N), p, d.irradiance_texture_width, d.irradiance_texture_height, d.irradiance_probe_side_length);
        vec3 probe_irr = atlas_bilinear<4>(irradiance, d.irradiance_texture_width, d.irradiance_texture_height, tc.x, tc.y);
        const float crush = 0.2f;
        if (weight < crush) weight = weight * (weight * weight * (1.0f / (crush * crush)));
        weight = weight * (tri.x * tri.y * tri.z);
        probe_irr = v3(std::sqrt(probe_irr.x), std::sqrt(probe_irr.y), std::sqrt(probe_irr.z)); // LINEAR_BLENDING undefined => sqrt space (quirk 10)
        sum_irr = sum_irr + weight * probe_irr;
        sum_w += weight;
    }
    vec3 net = sum_irr / sum_w;
    net.x = (net.x != net.x) ? 0.5f : net.x;
    net.y = (net.y != net.y) ? 0.5f : net.y;
    net.z = (net.z != net.z) ? 0.5f : net.z;
    net = net * net;
    net = net * d.energy_preservation;
    return (0.5f * ORC_M_PI) * net;
}

// ------------------------------------------------------------------------------------------- hit shading
struct SurfaceHit
{
    vec3  P, N;
    vec3  albedo;
    float roughness, metallic;
};

// texture(s_Textures[i], uv) in a ray-tracing stage: no derivatives, so level 0.  Pinned sampler (the reference's is made
// in the un-vendored framework): bilinear with fp32 weights at uv*size - 0.5, REPEAT addressing, UNORM8 texel = b / 255.
static inline vec4 sample_texture(const Scene::Texture& t, float u, float v)
{
    const float px = u * (float)t.w - 0.5f, py = v * (float)t.h - 0.5f;
    const float fx0 = std::floor(px), fy0 = std::floor(py);
    const float fx = px - fx0, fy = py - fy0;
    auto wrap = [](int c, int n) { c %= n; return c < 0 ? c + n : c; };
    const int x0 = wrap((int)fx0, t.w), x1 = wrap((int)fx0 + 1, t.w), y0 = wrap((int)fy0, t.h), y1 = wrap((int)fy0 + 1, t.h);
    auto texel = [&](int x, int y, int c) { return (float)t.rgba[((size_t)y * t.w + x) * 4 + c] / 255.0f; };
    float r[4];
    for (int c = 0; c < 4; c++)
    {
        const float top = texel(x0, y0, c) * (1.0f - fx) + texel(x1, y0, c) * fx;
        const float bot = texel(x0, y1, c) * (1.0f - fx) + texel(x1, y1, c) * fx;
        r[c] = top * (1.0f - fy) + bot * fy;
    }
    return vec4 { r[0], r[1], r[2], r[3] };
}
static inline float comp4(vec4 v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }

// interpolated_vertex + transform_vertex (identity model) + fetch_albedo / fetch_roughness / fetch_metallic / fetch_normal
// (scene_descriptor_set.glsl:133-220).  Quirk kept: every hit shader calls fetch_normal(material, tangent, TANGENT, normal, uv)
// (reflections_ray_trace.rchit:134, gi_ray_trace.rchit:112, ground_truth_path_trace.rchit:131), so the TBN matrix of a
// normal-mapped material is (T, T, N).
// mat3(model_matrix) * v and model_matrix * vec4(p, 1), rows summed left to right (transform_vertex, scene_descriptor_set.glsl:150-160)
static inline vec3 inst_mul3(const float* m, vec3 v)
{
    return v3((m[0] * v.x + m[4] * v.y) + m[8] * v.z, (m[1] * v.x + m[5] * v.y) + m[9] * v.z, (m[2] * v.x + m[6] * v.y) + m[10] * v.z);
}
static inline vec3 inst_point(const float* m, vec3 p)
{
    return v3(((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12] * 1.0f, ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13] * 1.0f, ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14] * 1.0f);
}
static inline SurfaceHit surface_at(const Scene& s, const Hit& h)
{
    SurfaceHit o;
    // instanced scene: interpolate the mesh's object-space attributes, then transform_vertex with the instance's matrix
    const Scene::Instance* ir = s.tri_instance.empty() ? nullptr : &s.instances[s.tri_instance[(size_t)h.prim]];
    const size_t q = ir ? (size_t)ir->mesh_tri_base + ((uint32_t)h.prim - ir->first_tri) : (size_t)h.prim;
    vec3 pv0, pv1, pv2;
    if (ir)
    {
        const float* p = &s.mesh_positions[q * 9];
        pv0 = v3(p[0], p[1], p[2]); pv1 = v3(p[3], p[4], p[5]); pv2 = v3(p[6], p[7], p[8]);
    }
    else { const Tri& t = s.tris[h.prim]; pv0 = t.v0; pv1 = t.v1; pv2 = t.v2; }
    float b0 = 1.0f - h.u - h.v, b1 = h.u, b2 = h.v;
    o.P = (pv0 * b0 + pv1 * b1) + pv2 * b2;
    if (ir) o.P = inst_point(ir->m, o.P);
    const std::vector<float>& normals = ir ? s.mesh_normals : s.tri_normals;
    vec3 n;
    if (!normals.empty())
    {
        const float* qn = &normals[q * 9];
        n = (v3(qn[0], qn[1], qn[2]) * b0 + v3(qn[3], qn[4], qn[5]) * b1) + v3(qn[6], qn[7], qn[8]) * b2;
    }
    else
        n = cross(pv1 - pv0, pv2 - pv0);
    o.N = ir ? normalize(inst_mul3(ir->m, normalize(n))) : normalize(normalize(n)); // interpolated_vertex normalises, transform_vertex normalises again
    const std::vector<uint32_t>& tmat = ir ? s.mesh_material : s.tri_material;
    uint32_t mat = tmat.empty() ? 0u : tmat[q];
    if (!s.materials.empty())
    {
        const float* m = &s.materials[(size_t)mat * 8];
        o.albedo = v3(m[0], m[1], m[2]); o.metallic = m[3]; o.roughness = fmax2(m[4], 0.1f);
    }
    else { o.albedo = v3(0.8f, 0.8f, 0.8f); o.metallic = 0.0f; o.roughness = 0.5f; }
    if (!s.mat_tex.empty())
    {
        const int32_t* mt = &s.mat_tex[(size_t)mat * 6];
        float tu = 0.0f, tv = 0.0f;
        const std::vector<float>& uvs = ir ? s.mesh_uvs : s.tri_uvs;
        if (!uvs.empty())
        {
            const float* qu = &uvs[q * 6];
            tu = (qu[0] * b0 + qu[2] * b1) + qu[4] * b2;
            tv = (qu[1] * b0 + qu[3] * b1) + qu[5] * b2;
        }
        if (mt[0] >= 0) { vec4 c = sample_texture(s.textures[mt[0]], tu, tv); o.albedo = v3(c.x, c.y, c.z); }
        if (mt[2] >= 0) o.roughness = fmax2(comp4(sample_texture(s.textures[mt[2]], tu, tv), mt[4]), 0.1f);
        if (mt[3] >= 0) o.metallic = comp4(sample_texture(s.textures[mt[3]], tu, tv), mt[5]);
        if (mt[1] >= 0)
        {
            vec3 tg = v3(1.0f, 0.0f, 0.0f);
            const std::vector<float>& tans = ir ? s.mesh_tangents : s.tri_tangents;
            if (!tans.empty())
            {
                const float* qt = &tans[q * 9];
                tg = (v3(qt[0], qt[1], qt[2]) * b0 + v3(qt[3], qt[4], qt[5]) * b1) + v3(qt[6], qt[7], qt[8]) * b2;
            }
            tg = ir ? normalize(inst_mul3(ir->m, normalize(tg))) : normalize(normalize(tg));   // interpolated_vertex, then transform_vertex
            const vec3 T = normalize(tg), Nn = normalize(o.N);   // get_normal_from_map: TBN = (T, T, N) (quirk above)
            vec4 c  = sample_texture(s.textures[mt[1]], tu, tv);
            vec3 tn = normalize(v3(c.x, c.y, c.z) * 2.0f - v3(1.0f, 1.0f, 1.0f));
            o.N     = normalize((T * tn.x + T * tn.y) + Nn * tn.z);
        }
    }
    return o;
}

// fetch_light_properties (lighting.glsl:6-111) without SOFT_SHADOWS, with RAY_TRACING
static inline void fetch_light_hard(const Light& L, vec3 Wo, vec3 P, vec3 N, vec3* Li, vec3* Wi, vec3* Wh, float* t_max, float* attenuation)
{
    const int  type = (int)L.data3[0];
    const vec3 ldir = v3(L.data0[0], L.data0[1], L.data0[2]);
    *Li = v3(L.data2[0], L.data2[1], L.data2[2]) * L.data0[3];
    if (type == 0) { *Wi = ldir; *t_max = 10000.0f; *attenuation = 1.0f; }
    else
    {
        vec3  to_light = v3(L.data1[0], L.data1[1], L.data1[2]) - P;
        *Wi            = normalize(to_light);
        float dist     = length(to_light);
        *t_max         = dist;
        if (type == 1) *attenuation = 1.0f / (dist * dist);
        else
        {
            float aa     = smoothstepf(L.data3[1], L.data3[2], dot(*Wi, ldir));
            *attenuation = aa / (dist * dist);
        }
    }
    *Wh          = normalize(Wo + *Wi);
    *attenuation = *attenuation * clampf(dot(N, *Wi), 0.0f, 1.0f);
}

// direct_lighting (lighting.glsl:117-196): RAY_TRACING always; T = throughput; sky = SAMPLE_SKY_LIGHT
static inline vec3 direct_lighting(const Scene& scene, const Light& light, vec3 Wo, vec3 N, vec3 P, vec3 F0, vec3 diffuse_color, float roughness,
                                   vec3 T, bool sample_sky, float r2x, float r2y, const CubeH* sky, uint64_t* rays)
{
    vec3 Lo = v3(0, 0, 0);
    vec3 ray_origin = P + N * 0.1f;
    {
        vec3  Li, Wi, Wh;
        float t_max, attenuation;
        fetch_light_hard(light, Wo, P, N, &Li, &Wi, &Wh, &t_max, &attenuation);
        if (attenuation > 0.0f)
        {
            if (rays) (*rays)++;
            attenuation = attenuation * (scene.any_hit(ray_origin, Wi, 0.01f, t_max) ? 0.0f : 1.0f);
        }
        vec3 brdf = evaluate_uber_brdf(diffuse_color, roughness, N, F0, Wo, Wh, Wi);
        Lo = Lo + ((T * brdf) * attenuation) * Li;
    }
    if (sample_sky)
    {
        vec3 Wi = sample_cosine_lobe(N, r2x, r2y);
        vec3 Li = sky->fetch(Wi);
        vec3 Wh = normalize(Wo + Wi);
        if (rays) (*rays)++;
        Li = Li * (scene.any_hit(ray_origin, Wi, 0.01f, 10000.0f) ? 0.0f : 1.0f);
        vec3 brdf = evaluate_uber_brdf(diffuse_color, roughness, N, F0, Wo, Wh, Wi);
        Lo = Lo + (T * brdf) * Li;
    }
    return Lo;
}

} // namespace orc
