// Hit shading shared by the reflection and DDGI trace kernels — device restatement of
//   brdf.glsl:36-142 (GGX / Schlick / evaluate_uber_brdf), lighting.glsl:6-196 (fetch_light_properties
//   without SOFT_SHADOWS, direct_lighting with RAY_THROUGHPUT / SAMPLE_SKY_LIGHT),
//   gi_common.glsl:39-320 (probe addressing, oct coding, sample_irradiance), random.glsl:11-56,
//   scene_descriptor_set.glsl:102-220 (interpolated_vertex / transform_vertex / fetch_* with constant or textured
//   materials; instances are flattened at scene build, so the model matrix is the identity).
// Pinned where the reference defers to samplers / absent assets: environment cubemaps
// are fetched NEAREST with the Vulkan face-selection rule; DDGI atlases are sampled bilinearly
// (ddgi.cpp:478,499) with fp32 weights mix(mix(t00,t10,fx), mix(t01,t11,fx), fy), clamp-to-edge.
#pragma once
#include "../../include/hr_api.h"
#include "traverse.h"

#ifdef HR_PROBE_FAST_HIT
// developer probe (docs/EXPERIMENTS.md R5.6): an UPPER BOUND on what tolerance-mode hit shading could save — every correctly rounded
// quotient / root / normalisation of this header through the hardware approximations, for every kernel that includes it (the exact mode of
// such a build is NOT bit-exact; never shipped)
namespace hr { HR_DEV f3 probe_normalize3(f3 a) { return scale3(a, __builtin_amdgcn_rsqf(dot3(a, a))); } }
#define __fdiv_rn(a, b) ((a) * __builtin_amdgcn_rcpf(b))
#define hr_sqrt(x) __builtin_amdgcn_sqrtf(x)
#define normalize3(v) probe_normalize3(v)
#endif

namespace hr {

#define HR_EPSILON 0.0001f

// ---- random.glsl ------------------------------------------------------------------------------
struct Rng { uint32_t x, y; };
HR_DEV uint32_t rng_rotl(uint32_t x, uint32_t k) { return (x << k) | (x >> (32 - k)); }
HR_DEV uint32_t rng_next(Rng& r)
{
    uint32_t result = r.x * 0x9e3779bbu;
    r.y ^= r.x;
    r.x = rng_rotl(r.x, 26) ^ r.y ^ (r.y << 9);
    r.y = rng_rotl(r.y, 13);
    return result;
}
HR_DEV uint32_t rng_hash(uint32_t seed)
{
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u;
    seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    seed = seed ^ (seed >> 15);
    return seed;
}
HR_DEV Rng rng_init(uint32_t idx, uint32_t idy, uint32_t frame)
{
    Rng r;
    r.x = rng_hash((idx << 16) | idy);
    r.y = rng_hash(frame);
    rng_next(r);
    return r;
}
HR_DEV float next_float(Rng& r) { return __uint_as_float(0x3f800000u | (rng_next(r) >> 9)) - 1.0f; }

// ---- environment ------------------------------------------------------------------------------
struct CubeMap
{
    const uint2* p; // [6][S][S] RGBA16F
    int          S;
    HR_DEV f3 fetch(f3 d) const
    {
        const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
        int   face;
        float sc, tc, ma;
        if (ax >= ay && ax >= az) { ma = ax; if (d.x >= 0.0f) { face = 0; sc = -d.z; tc = -d.y; } else { face = 1; sc = d.z; tc = -d.y; } }
        else if (ay >= az) { ma = ay; if (d.y >= 0.0f) { face = 2; sc = d.x; tc = d.z; } else { face = 3; sc = d.x; tc = -d.z; } }
        else { ma = az; if (d.z >= 0.0f) { face = 4; sc = d.x; tc = -d.y; } else { face = 5; sc = -d.x; tc = -d.y; } }
        const float s = 0.5f * (__fdiv_rn(sc, ma) + 1.0f), t = 0.5f * (__fdiv_rn(tc, ma) + 1.0f);
        int ix = (int)floorf(s * (float)S), iy = (int)floorf(t * (float)S);
        ix = ix < 0 ? 0 : (ix > S - 1 ? S - 1 : ix);
        iy = iy < 0 ? 0 : (iy > S - 1 ? S - 1 : iy);
        const uint2 q = p[((size_t)face * S + iy) * S + ix];
        return mk3(h2f_lo(q.x), h2f_hi(q.x), h2f_lo(q.y));
    }
};

// ---- brdf.glsl -----------------------------------------------------------------------------------
HR_DEV f3 mul3(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
HR_DEV f3 div3s(f3 a, float s) { return mk3(__fdiv_rn(a.x, s), __fdiv_rn(a.y, s), __fdiv_rn(a.z, s)); }
HR_DEV f3 mix3(f3 a, f3 b, float t) { return add3(scale3(a, 1.0f - t), scale3(b, t)); }
HR_DEV f3 one3() { return mk3(1.0f, 1.0f, 1.0f); }

HR_DEV float D_ggx(float ndoth, float alpha)
{
    const float a2 = alpha * alpha;
    const float denom = (ndoth * ndoth) * (a2 - 1.0f) + 1.0f;
    return __fdiv_rn(a2, max2(HR_EPSILON, (HR_M_PI * denom * denom)));
}
HR_DEV float G1_schlick_ggx(float roughness, float ndotv)
{
    const float k = __fdiv_rn((roughness + 1.0f) * (roughness + 1.0f), 8.0f);
    return __fdiv_rn(ndotv, max2(HR_EPSILON, (ndotv * (1.0f - k) + k)));
}
HR_DEV float G_schlick_ggx(float ndotl, float ndotv, float roughness) { return G1_schlick_ggx(roughness, ndotl) * G1_schlick_ggx(roughness, ndotv); }
HR_DEV f3 F_schlick(f3 f0, float vdoth)
{
    const float p = det_powi(1.0f - vdoth, 5);
    return add3(f0, scale3(sub3(one3(), f0), p));
}
HR_DEV f3 evaluate_specular_brdf(float roughness, f3 F, float ndoth, float ndotl, float ndotv)
{
    const float alpha = roughness * roughness;
    const f3    num   = scale3(scale3(F, D_ggx(ndoth, alpha)), G_schlick_ggx(ndotl, ndotv, roughness));
    return div3s(num, max2(HR_EPSILON, (4.0f * ndotl * ndotv)));
}
HR_DEV f3 evaluate_uber_brdf(f3 diffuse_color, float roughness, f3 N, f3 F0, f3 Wo, f3 Wh, f3 Wi)
{
    const float NdotL = max2(dot3(N, Wi), 0.0f), NdotV = max2(dot3(N, Wo), 0.0f), NdotH = max2(dot3(N, Wh), 0.0f), VdotH = max2(dot3(Wi, Wh), 0.0f);
    const f3 F        = F_schlick(F0, VdotH);
    const f3 specular = evaluate_specular_brdf(roughness, F, NdotH, NdotL, NdotV);
    const f3 diffuse  = div3s(diffuse_color, HR_M_PI);
    return add3(mul3(sub3(one3(), F), diffuse), specular);
}
HR_DEV f3 fresnel_schlick_roughness(float cos_theta, f3 F0, float roughness)
{
    const float r1 = 1.0f - roughness;
    const f3    m  = mk3(max2(r1, F0.x), max2(r1, F0.y), max2(r1, F0.z));
    const float p  = det_powi(max2(1.0f - cos_theta, 0.0f), 5);
    return add3(F0, scale3(sub3(m, F0), p));
}
// brdf.glsl:8-32
HR_DEV f3 sample_cosine_lobe_n(f3 n, float rx, float ry)
{
    rx = max2(0.00001f, rx);
    ry = max2(0.00001f, ry);
    const float phi = 2.0f * HR_M_PI * ry;
    const float ct = hr_sqrt(rx), st = hr_sqrt(1.0f - rx);
    float s, c;
    det_sincos(phi, s, c);
    const f3 t   = mk3(st * c, st * s, ct);
    const f3 ref = fabsf(dot3(n, mk3(0.0f, 1.0f, 0.0f))) > 0.99f ? mk3(0.0f, 0.0f, 1.0f) : mk3(0.0f, 1.0f, 0.0f);
    const f3 x   = normalize3(cross3(ref, n));
    const f3 y   = cross3(n, x);
    return normalize3(mk3((x.x * t.x + y.x * t.y) + n.x * t.z, (x.y * t.x + y.y * t.y) + n.y * t.z, (x.z * t.x + y.z * t.y) + n.z * t.z));
}

// ---- gi_common.glsl ----------------------------------------------------------------------------------
typedef hr_ddgi_uniforms DDGIU;

HR_DEV float sign_not_zero(float k) { return k >= 0.0f ? 1.0f : -1.0f; }
HR_DEV void gi_oct_encode(f3 v, float& rx, float& ry)
{
    const float l1  = (fabsf(v.x) + fabsf(v.y)) + fabsf(v.z);
    const float inv = __fdiv_rn(1.0f, l1);
    rx = v.x * inv;
    ry = v.y * inv;
    if (v.z < 0.0f)
    {
        const float nx = (1.0f - fabsf(ry)) * sign_not_zero(rx);
        const float ny = (1.0f - fabsf(rx)) * sign_not_zero(ry);
        rx = nx; ry = ny;
    }
}
HR_DEV f3 gi_oct_decode(float ox, float oy)
{
    f3 v = mk3(ox, oy, 1.0f - fabsf(ox) - fabsf(oy));
    if (v.z < 0.0f)
    {
        const float nx = (1.0f - fabsf(v.y)) * sign_not_zero(v.x);
        const float ny = (1.0f - fabsf(v.x)) * sign_not_zero(v.y);
        v.x = nx; v.y = ny;
    }
    return normalize3(v);
}
HR_DEV f3 grid_coord_to_position(const DDGIU& d, int cx, int cy, int cz)
{
    return mk3(d.grid_step[0] * (float)cx + d.grid_start_position[0], d.grid_step[1] * (float)cy + d.grid_start_position[1],
               d.grid_step[2] * (float)cz + d.grid_start_position[2]);
}
HR_DEV f3 probe_location(const DDGIU& d, int index)
{
    const int cx = index % d.probe_counts[0];
    const int cy = (index % (d.probe_counts[0] * d.probe_counts[1])) / d.probe_counts[0];
    const int cz = index / (d.probe_counts[0] * d.probe_counts[1]);
    return grid_coord_to_position(d, cx, cy, cz);
}
// (col, row) = the probe's cell in the atlas = (probe_index % per_row, probe_index / per_row) with per_row = (tw - 2) / (side + 2)
// (gi_common.glsl:164-184).  The atlas is probe_counts.x * probe_counts.y cells wide by construction (ddgi.cpp:197-201, checked in
// hr_ddgi_create), so for probe (cx, cy, cz) the cell is (cx + cy * nx, cz): the same integers without two per-lane divisions.
// ... in two halves: the offset of the direction inside a probe's cell (the same for every probe) and the cell's corner.  The four
// divisions are by the atlas extents: with the denominator half of the correctly rounded sequence done once (DivBy, device_math.h — the very
// FMAs the compiler's expansion of `/` performs, bit-identical results) a quotient costs five FMAs instead of ~eleven instructions.
// inrange (the caller's, once per gather — wave-uniform, the atlas extents are): both denominators are `fast` and at most 3e5.  All four
// numerators then lie in div_by's fast range by construction — a cell corner is 2 .. the atlas extent, an in-cell offset z * side is 0 or at
// least 2^-25 * side (z = (o + 1) / 2 with |o| <= 1 + an ulp) — and the quotients need no per-lane range branches (div_by_if): reflections trace
// 165 -> 164 / 454 -> 444 us, exact DDGI sample 180 -> 173 / 672 -> 646 us (1080p / 4K, profiles/r6_h/passbench_inrange_gather.txt).  Instantiating the
// whole probe loop for both values of the flag (no uniform branches left in it) measured no better: 162 / 447 and 180 / 677.
HR_DEV bool atlas_div_inrange(const DivBy& Dw, const DivBy& Dh) { return Dw.fast && Dh.fast && Dw.d <= 3e5f && Dh.d <= 3e5f; }
HR_DEV void texture_coord_in_cell(f3 dir, const DivBy& Dw, const DivBy& Dh, bool inrange, int side, float& cx, float& cy)
{
    float ox, oy;
    gi_oct_encode(normalize3(dir), ox, oy);
    const float zx = (ox + 1.0f) * 0.5f, zy = (oy + 1.0f) * 0.5f;
    cx = div_by_if(inrange, zx * (float)side, Dw); cy = div_by_if(inrange, zy * (float)side, Dh);
}
HR_DEV void texture_coord_of_cell(float cx, float cy, int col, int row, const DivBy& Dw, const DivBy& Dh, bool inrange, int side, float& u, float& v)
{
    const float pwb = (float)side + 2.0f;
    const float tlx = (float)col * pwb + 2.0f;
    const float tly = (float)row * pwb + 2.0f;
    u = div_by_if(inrange, tlx, Dw) + cx;
    v = div_by_if(inrange, tly, Dh) + cy;
}
HR_DEV void texture_coord_from_cell(f3 dir, int col, int row, const DivBy& Dw, const DivBy& Dh, bool inrange, int side, float& u, float& v)
{
    float cx, cy;
    texture_coord_in_cell(dir, Dw, Dh, inrange, side, cx, cy);
    texture_coord_of_cell(cx, cy, col, row, Dw, Dh, inrange, side, u, v);
}
HR_DEV void texture_coord_from_cell(f3 dir, int col, int row, int tw, int th, int side, float& u, float& v)
{
    const DivBy Dw = div_prepare((float)tw), Dh = div_prepare((float)th);
    texture_coord_from_cell(dir, col, row, Dw, Dh, atlas_div_inrange(Dw, Dh), side, u, v);
}
HR_DEV void texture_coord_from_direction(f3 dir, int probe_index, int tw, int th, int side, float& u, float& v)
{
    const int per_row = (tw - 2) / (side + 2);
    texture_coord_from_cell(dir, probe_index % per_row, probe_index / per_row, tw, th, side, u, v);   // mod() of small integers is exact
}
HR_DEV int clampi(int a, int lo, int hi) { return a < lo ? lo : (a > hi ? hi : a); }

struct AtlasRGBA { const uint2* p; int w, h; };
struct AtlasRG { const uint32_t* p; int w, h; };
HR_DEV void bilinear_setup(float u, float v, int w, int h, int& x0, int& x1, int& y0, int& y1, float& fx, float& fy)
{
    const float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    const float fx0 = floorf(x), fy0 = floorf(y);
    fx = x - fx0; fy = y - fy0;
    x0 = (int)fx0; y0 = (int)fy0; x1 = x0 + 1; y1 = y0 + 1;
    x0 = clampi(x0, 0, w - 1); x1 = clampi(x1, 0, w - 1); y0 = clampi(y0, 0, h - 1); y1 = clampi(y1, 0, h - 1);
}
HR_DEV f3 atlas_bilinear_rgb(const AtlasRGBA& a, float u, float v)
{
    int x0, x1, y0, y1; float fx, fy;
    bilinear_setup(u, v, a.w, a.h, x0, x1, y0, y1, fx, fy);
    const uint2 t00 = a.p[(size_t)y0 * a.w + x0], t10 = a.p[(size_t)y0 * a.w + x1], t01 = a.p[(size_t)y1 * a.w + x0], t11 = a.p[(size_t)y1 * a.w + x1];
    f3 r;
    r.x = mix1(mix1(h2f_lo(t00.x), h2f_lo(t10.x), fx), mix1(h2f_lo(t01.x), h2f_lo(t11.x), fx), fy);
    r.y = mix1(mix1(h2f_hi(t00.x), h2f_hi(t10.x), fx), mix1(h2f_hi(t01.x), h2f_hi(t11.x), fx), fy);
    r.z = mix1(mix1(h2f_lo(t00.y), h2f_lo(t10.y), fx), mix1(h2f_lo(t01.y), h2f_lo(t11.y), fx), fy);
    return r;
}
HR_DEV void atlas_bilinear_rg(const AtlasRG& a, float u, float v, float& r0, float& r1)
{
    int x0, x1, y0, y1; float fx, fy;
    bilinear_setup(u, v, a.w, a.h, x0, x1, y0, y1, fx, fy);
    const uint32_t t00 = a.p[(size_t)y0 * a.w + x0], t10 = a.p[(size_t)y0 * a.w + x1], t01 = a.p[(size_t)y1 * a.w + x0], t11 = a.p[(size_t)y1 * a.w + x1];
    r0 = mix1(mix1(h2f_lo(t00), h2f_lo(t10), fx), mix1(h2f_lo(t01), h2f_lo(t11), fx), fy);
    r1 = mix1(mix1(h2f_hi(t00), h2f_hi(t10), fx), mix1(h2f_hi(t01), h2f_hi(t11), fx), fy);
}

// gi_common.glsl:188-316: the weighted sqrt-space mean `net` (NaN components replaced by 0.5), before the squaring and scaling of :317-320 —
// split into the part that places the shading point in the grid (:190-203) and the body of the eight-probe loop (:207-296), so that the
// loop can run per lane (sample_irradiance_net) or one probe per lane (sample_irradiance_net_coop) on the very same operations.
struct IrrCell { int bx, by, bz; f3 alpha; float icx, icy; };   // + the offset of N's texel inside a probe's irradiance cell (the same for all eight probes)
struct IrrDiv { DivBy iw, ih, dw, dh; bool irr_in, dep_in; };     // the four atlas extents as prepared denominators (+ atlas_div_inrange of each pair)
HR_DEV IrrDiv irradiance_div(const DDGIU& d)
{
    IrrDiv D;
    D.iw = div_prepare((float)d.irradiance_texture_width); D.ih = div_prepare((float)d.irradiance_texture_height);
    D.dw = div_prepare((float)d.depth_texture_width); D.dh = div_prepare((float)d.depth_texture_height);
    D.irr_in = atlas_div_inrange(D.iw, D.ih); D.dep_in = atlas_div_inrange(D.dw, D.dh);
    return D;
}
HR_DEV IrrCell irradiance_cell(const DDGIU& d, const IrrDiv& D, f3 P, f3 N)
{
    const f3 gs = mk3(d.grid_step[0], d.grid_step[1], d.grid_step[2]);
    const f3 g0 = mk3(d.grid_start_position[0], d.grid_start_position[1], d.grid_start_position[2]);
    IrrCell c;
    c.bx = clampi((int)__fdiv_rn(P.x - g0.x, gs.x), 0, d.probe_counts[0] - 1);
    c.by = clampi((int)__fdiv_rn(P.y - g0.y, gs.y), 0, d.probe_counts[1] - 1);
    c.bz = clampi((int)__fdiv_rn(P.z - g0.z, gs.z), 0, d.probe_counts[2] - 1);
    const f3 base_pos = grid_coord_to_position(d, c.bx, c.by, c.bz);
    c.alpha = mk3(clamp1(__fdiv_rn(P.x - base_pos.x, gs.x), 0.0f, 1.0f), clamp1(__fdiv_rn(P.y - base_pos.y, gs.y), 0.0f, 1.0f),
                  clamp1(__fdiv_rn(P.z - base_pos.z, gs.z), 0.0f, 1.0f));
    texture_coord_in_cell(normalize3(N), D.iw, D.ih, D.irr_in, d.irradiance_probe_side_length, c.icx, c.icy);
    return c;
}
struct IrrTerm { f3 s; float w; };   // sqrt(probe irradiance) * weight, weight
HR_DEV IrrTerm irradiance_probe_term(const DDGIU& d, const IrrDiv& D, f3 P, f3 N, f3 Wo, const AtlasRGBA& irradiance, const AtlasRG& depth, const IrrCell& c, const int i)
{
    const int bx = c.bx, by = c.by, bz = c.bz;
    const f3  alpha = c.alpha;
    const int ox = i & 1, oy = (i >> 1) & 1, oz = (i >> 2) & 1;
    const int cx = clampi(bx + ox, 0, d.probe_counts[0] - 1), cy = clampi(by + oy, 0, d.probe_counts[1] - 1), cz = clampi(bz + oz, 0, d.probe_counts[2] - 1);
    const int col = cx + cy * d.probe_counts[0];   // probe cx + cy * nx + cz * nx * ny sits in atlas cell (col, cz)
    const f3 probe_pos      = grid_coord_to_position(d, cx, cy, cz);
    const f3 probe_to_point = add3(sub3(P, probe_pos), scale3(add3(N, scale3(Wo, 3.0f)), d.normal_bias));
    const f3 dir            = normalize3(neg3(probe_to_point));
    const f3 tri = mk3(mix1(1.0f - alpha.x, alpha.x, (float)ox), mix1(1.0f - alpha.y, alpha.y, (float)oy), mix1(1.0f - alpha.z, alpha.z, (float)oz));
    float weight = 1.0f;
    {
        const f3    tdp = normalize3(sub3(probe_pos, P));
        const float t   = max2(0.0001f, (dot3(tdp, N) + 1.0f) * 0.5f);
        weight          = weight * (t * t + 0.2f);
    }
    if (d.visibility_test == 1)
    {
        float u, v, mean, m2;
        texture_coord_from_cell(neg3(dir), col, cz, D.dw, D.dh, D.dep_in, d.depth_probe_side_length, u, v);
        const float dist = len3(probe_to_point);
        atlas_bilinear_rg(depth, u, v, mean, m2);
        const float variance = fabsf(mean * mean - m2);
        const float dm  = max2(dist - mean, 0.0f);
        float che = __fdiv_rn(variance, variance + dm * dm);
        che       = max2(che * che * che, 0.0f);
        weight    = weight * ((dist <= mean) ? 1.0f : che);
    }
    weight = max2(0.000001f, weight);
    float u, v;
    texture_coord_of_cell(c.icx, c.icy, col, cz, D.iw, D.ih, D.irr_in, d.irradiance_probe_side_length, u, v);
    f3 probe_irr = atlas_bilinear_rgb(irradiance, u, v);
    const float crush = 0.2f;
    if (weight < crush) weight = weight * (weight * weight * (1.0f / (crush * crush)));
    weight    = weight * (tri.x * tri.y * tri.z);
    probe_irr = mk3(hr_sqrt(probe_irr.x), hr_sqrt(probe_irr.y), hr_sqrt(probe_irr.z)); // sqrt-space blending (LINEAR_BLENDING undefined)
    IrrTerm t;
    t.s = scale3(probe_irr, weight);
    t.w = weight;
    return t;
}
HR_DEV f3 irradiance_net_from_sums(f3 sum_irr, float sum_w)
{
    f3 net = div3s(sum_irr, sum_w);
    net.x = (net.x != net.x) ? 0.5f : net.x;
    net.y = (net.y != net.y) ? 0.5f : net.y;
    net.z = (net.z != net.z) ? 0.5f : net.z;
    return net;
}
HR_DEV f3 sample_irradiance_net(const DDGIU& d, f3 P, f3 N, f3 Wo, const AtlasRGBA& irradiance, const AtlasRG& depth)
{
    const IrrDiv  D = irradiance_div(d);
    const IrrCell c = irradiance_cell(d, D, P, N);
    f3    sum_irr = mk3(0.0f, 0.0f, 0.0f);
    float sum_w   = 0.0f;
    // deliberately NOT unrolled: fully unrolled the eight probes' fetches overlap, but the kernels that inline this need
    // 168-178 VGPRs (2-3 waves per SIMD) — DDGI trace 0.43 -> 0.69 ms, reflections trace 0.31 -> 0.47 ms, sample pass unchanged
#ifndef HR_IRR_UNROLL
#define HR_IRR_UNROLL 1   // round 6, partial unrolling re-measured (profiles/r6_d/ab_exact_gather_unroll.txt): 2 -> reflections trace 174 / 481 us (1: 172 / 467),
#endif                    // DDGI trace 262 / 260 (1: 262 / 259); 4 -> 183 / 530 and 275 / 282.  The rolled loop stays.
    constexpr int kIrrUnroll = HR_IRR_UNROLL;   // a constant, not the macro, in the pragma: -save-temps (tools/isa_stats.py) re-parses preprocessed text
#pragma unroll kIrrUnroll
    for (int i = 0; i < 8; ++i)
    {
        const IrrTerm t = irradiance_probe_term(d, D, P, N, Wo, irradiance, depth, c, i);
        sum_irr = add3(sum_irr, t.s);
        sum_w += t.w;
    }
    return irradiance_net_from_sums(sum_irr, sum_w);
}
// The same gather as ONE call per wave (round 6): every lane of the wave calls, `want` says which lanes have a shading point.  With
// HR_IRR_COOP_MAX = 0 (shipping) those lanes run the per-lane loop above.  The WAVE-COOPERATIVE form behind it (developer A/B,
// -DHR_IRR_COOP_MAX=k) computes the eight probe terms of a point on eight lanes at once — a wave with k points takes ceil(k / 8) turns
// instead of eight; every term is the value irradiance_probe_term computes for that point and probe and the owner adds them in probe order,
// so `net` is the per-lane loop's bit for bit (the GPU suite passes on it) — but it measured slower at every threshold, see below.
// lds: kIrrCoopLdsFloats floats of the wave's own LDS that nothing else uses during the call (the traversal stack between two walks).
constexpr int kIrrCoopIn = 16, kIrrCoopLdsFloats = 8 * kIrrCoopIn + 8 * 32;
#ifndef HR_IRR_COOP_MAX
#define HR_IRR_COOP_MAX 0    // points per wave up to which the shared form is used.  0 (shipping): never — measured on the reflections trace kernel at
                             // 1080p / 4K (profiles/r6_c/ab_coop_threshold.txt): 0 -> 170-174 / 461-472 us, 24 -> 175 / 505, 40 -> 174 / 505, 56 -> 176 / 498,
                             // 64 -> 179 / 519; DDGI probe trace 254 -> 266 with it.  Sharing never pays: waves with few points are rare (materials and
                             // hit / miss regions are larger than an 8x8 tile), and every turn costs two LDS round trips behind wave barriers plus an owner-only
                             // summation.  What DID pay is having ONE gather site per wave instead of two (reflections.hip): 185 -> 172 us, 488 -> 466 us.
#endif
constexpr int kIrrCoopMaxPoints = HR_IRR_COOP_MAX;
static_assert(kIrrCoopLdsFloats <= HR_STACK_ENTRIES * 64, "the cooperative gather borrows the wave's traversal stack");
HR_DEV f3 sample_irradiance_net_coop(bool want, const DDGIU& d, f3 P, f3 N, f3 Wo, const AtlasRGBA& irradiance, const AtlasRG& depth, float* lds, int lane)
{
    const unsigned long long m = __ballot(want);
    const int k = __popcll(m), rank = (int)lanes_below(m);
    f3 net = mk3(0.0f, 0.0f, 0.0f);
    if (k == 0) return net;
    // A (nearly) full wave gains nothing from sharing — eight turns either way, and the shared form pays an LDS round trip per turn: per-lane loop
    if (k > kIrrCoopMaxPoints)
    {
        if (want) net = sample_irradiance_net(d, P, N, Wo, irradiance, depth);
        return net;
    }
    float* in = lds;                      // [8][kIrrCoopIn]  P, N, Wo, the point's IrrCell — of the eight points of this turn
    float* tm = lds + 8 * kIrrCoopIn;     // [8][8][4]        their probe terms: s.xyz, w
    const IrrDiv D = irradiance_div(d);
    IrrCell mc;
    if (want) mc = irradiance_cell(d, D, P, N);   // once per point, by its owner
    const int g = lane >> 3, i = lane & 7;
    for (int b0 = 0; b0 < k; b0 += 8)   // wave-uniform
    {
        const bool mine = want && rank >= b0 && rank < b0 + 8;
        if (mine)
        {
            float* q = in + (rank - b0) * kIrrCoopIn;
            q[0] = P.x; q[1] = P.y; q[2] = P.z; q[3] = N.x; q[4] = N.y; q[5] = N.z; q[6] = Wo.x; q[7] = Wo.y; q[8] = Wo.z;
            q[9] = __int_as_float(mc.bx | (mc.by << 10) | (mc.bz << 20)); q[10] = mc.alpha.x; q[11] = mc.alpha.y; q[12] = mc.alpha.z; q[13] = mc.icx; q[14] = mc.icy;
        }
        wave_fence();
        if (b0 + g < k)
        {
            const float* q = in + g * kIrrCoopIn;
            const f3 p = mk3(q[0], q[1], q[2]), n = mk3(q[3], q[4], q[5]), wo = mk3(q[6], q[7], q[8]);
            IrrCell c;
            const int pk = __float_as_int(q[9]);
            c.bx = pk & 1023; c.by = (pk >> 10) & 1023; c.bz = (pk >> 20) & 1023;
            c.alpha = mk3(q[10], q[11], q[12]); c.icx = q[13]; c.icy = q[14];
            const IrrTerm t = irradiance_probe_term(d, D, p, n, wo, irradiance, depth, c, i);
            float* o = tm + (g * 8 + i) * 4;
            o[0] = t.s.x; o[1] = t.s.y; o[2] = t.s.z; o[3] = t.w;
        }
        wave_fence();
        if (mine)
        {
            const float* o = tm + (rank - b0) * 32;
            f3    sum_irr = mk3(0.0f, 0.0f, 0.0f);
            float sum_w   = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; j++)
            {
                sum_irr = add3(sum_irr, mk3(o[j * 4], o[j * 4 + 1], o[j * 4 + 2]));
                sum_w += o[j * 4 + 3];
            }
            net = irradiance_net_from_sums(sum_irr, sum_w);
        }
        wave_fence();
    }
    return net;
}
// net^2 * energy_preservation * pi / 2 (gi_common.glsl:317-320)
HR_DEV f3 irradiance_from_net(const DDGIU& d, f3 net)
{
    net = mul3(net, net);
    net = scale3(net, d.energy_preservation);
    return scale3(net, 0.5f * HR_M_PI);
}
// gi_common.glsl:188-320
HR_DEV f3 sample_irradiance(const DDGIU& d, f3 P, f3 N, f3 Wo, const AtlasRGBA& irradiance, const AtlasRG& depth)
{
    return irradiance_from_net(d, sample_irradiance_net(d, P, N, Wo, irradiance, depth));
}

// ---- scene access at a hit -----------------------------------------------------------------------------
struct SceneShading
{
    const float*    positions;   // [n][3][3] by original triangle index
    const float*    normals;     // [n][3][3] or null
    const uint32_t* tri_material;
    const float*    materials;   // [m][8]
    // textured materials (all null when the scene has none)
    const float*    uvs;         // [n][3][2] or null
    const float*    tangents;    // [n][3][3] or null
    const int32_t*  mat_tex;     // [m][6]: albedo, normal, roughness, metallic texture (-1 none), roughness channel, metallic channel
    const uint4*    tex_table;   // per texture: texel offset, width, height
    const uint32_t* tex_data;    // RGBA8 texels
    // instanced scenes (null otherwise): positions / normals / uvs / tangents / tri_material are then the OBJECT-space per-mesh arrays,
    // indexed by inst[tri_instance[prim]].mesh_tri_base + (prim - first_tri)
    const uint32_t*    tri_instance;
    const InstanceRec* inst;
};
// fills the members above from a scene (host)
static inline void scene_shading_from(const hr_scene* scene, SceneShading& sh)
{
    const bool inst = scene->n_instances > 0;
    sh.tri_instance = inst ? (const uint32_t*)scene->tri_instance.p : nullptr;
    sh.inst         = inst ? (const InstanceRec*)scene->inst_records.p : nullptr;
    sh.positions    = (const float*)(inst ? scene->mesh_positions.p : scene->positions.p);
    sh.normals      = scene->has_normals ? (const float*)(inst ? scene->mesh_normals.p : scene->tri_normals.p) : nullptr;
    sh.tri_material = scene->has_material ? (const uint32_t*)(inst ? scene->mesh_material.p : scene->tri_material.p) : nullptr;
    sh.materials    = scene->n_materials ? (const float*)scene->materials.p : nullptr;
    sh.uvs          = scene->has_uvs ? (const float*)(inst ? scene->mesh_uvs.p : scene->tri_uvs.p) : nullptr;
    sh.tangents     = scene->has_tangents ? (const float*)(inst ? scene->mesh_tangents.p : scene->tri_tangents.p) : nullptr;
    sh.mat_tex      = scene->has_textures ? (const int32_t*)scene->mat_tex.p : nullptr;
    sh.tex_table    = scene->has_textures ? (const uint4*)scene->tex_table.p : nullptr;
    sh.tex_data     = scene->has_textures ? (const uint32_t*)scene->tex_data.p : nullptr;
}
struct SurfaceHit
{
    f3    P, N, albedo;
    float roughness, metallic;
};
// texture(s_Textures[i], uv) in a ray-tracing stage: level 0; pinned sampler: bilinear with fp32 weights at uv*size - 0.5,
// REPEAT addressing, UNORM8 texel = b / 255 (checked against the reference's hit shaders by
// tests/test_ref_shaders.py::test_textured_materials_in_hit_shaders).  out[4] = r, g, b, a.
HR_DEV void sample_texture(const SceneShading& s, int tex, float u, float v, float* out)
{
    const uint4 t  = s.tex_table[tex];
    const int   w  = (int)t.y, h = (int)t.z;
    const float px = u * (float)w - 0.5f, py = v * (float)h - 0.5f;
    const float fx0 = floorf(px), fy0 = floorf(py);
    const float fx = px - fx0, fy = py - fy0;
    int x0 = (int)fx0 % w, x1 = ((int)fx0 + 1) % w, y0 = (int)fy0 % h, y1 = ((int)fy0 + 1) % h;
    x0 = x0 < 0 ? x0 + w : x0; x1 = x1 < 0 ? x1 + w : x1; y0 = y0 < 0 ? y0 + h : y0; y1 = y1 < 0 ? y1 + h : y1;
    const uint32_t* base = s.tex_data + t.x;
    const uint32_t t00 = base[(size_t)y0 * w + x0], t10 = base[(size_t)y0 * w + x1], t01 = base[(size_t)y1 * w + x0], t11 = base[(size_t)y1 * w + x1];
#pragma unroll
    for (int c = 0; c < 4; c++)
    {
        const float a = __fdiv_rn((float)((t00 >> (8 * c)) & 0xffu), 255.0f), b = __fdiv_rn((float)((t10 >> (8 * c)) & 0xffu), 255.0f);
        const float e = __fdiv_rn((float)((t01 >> (8 * c)) & 0xffu), 255.0f), g = __fdiv_rn((float)((t11 >> (8 * c)) & 0xffu), 255.0f);
        const float top = a * (1.0f - fx) + b * fx, bot = e * (1.0f - fx) + g * fx;
        out[c] = top * (1.0f - fy) + bot * fy;
    }
}
// interpolated_vertex + transform_vertex (identity model) + fetch_albedo / fetch_roughness / fetch_metallic / fetch_normal
// (scene_descriptor_set.glsl:133-220).  Quirk kept: the hit shaders call fetch_normal(material, tangent, TANGENT, normal, uv)
// (reflections_ray_trace.rchit:134, gi_ray_trace.rchit:112, ground_truth_path_trace.rchit:131): TBN = (T, T, N).
// mat3(model_matrix) * v and model_matrix * vec4(p, 1), each row summed left to right (transform_vertex, scene_descriptor_set.glsl:150-160)
HR_DEV f3 inst_mul3(const float* __restrict__ m, f3 v)
{
    return mk3((m[0] * v.x + m[4] * v.y) + m[8] * v.z, (m[1] * v.x + m[5] * v.y) + m[9] * v.z, (m[2] * v.x + m[6] * v.y) + m[10] * v.z);
}
HR_DEV f3 inst_point(const float* __restrict__ m, f3 p)
{
    return mk3(((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12] * 1.0f, ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13] * 1.0f, ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14] * 1.0f);
}
HR_DEV SurfaceHit surface_at(const SceneShading& s, const HitRec& h)
{
    SurfaceHit o;
    // instanced scene: the attributes live per MESH in object space; interpolate there, then transform_vertex with the instance's matrix
    const InstanceRec* ir = s.tri_instance ? s.inst + s.tri_instance[h.prim] : nullptr;
    const size_t       q  = ir ? (size_t)ir->mesh_tri_base + ((uint32_t)h.prim - ir->first_tri) : (size_t)h.prim;
    const float* p = s.positions + q * 9;
    const f3 v0 = mk3(p[0], p[1], p[2]), v1 = mk3(p[3], p[4], p[5]), v2 = mk3(p[6], p[7], p[8]);
    const float b0 = 1.0f - h.u - h.v, b1 = h.u, b2 = h.v;
    o.P = add3(add3(scale3(v0, b0), scale3(v1, b1)), scale3(v2, b2));
    if (ir) o.P = inst_point(ir->m, o.P);
    f3 n;
    if (s.normals)
    {
        const float* qn = s.normals + q * 9;
        n = add3(add3(scale3(mk3(qn[0], qn[1], qn[2]), b0), scale3(mk3(qn[3], qn[4], qn[5]), b1)), scale3(mk3(qn[6], qn[7], qn[8]), b2));
    }
    else
        n = cross3(sub3(v1, v0), sub3(v2, v0));
    o.N = ir ? normalize3(inst_mul3(ir->m, normalize3(n))) : normalize3(normalize3(n));
    const uint32_t mat = s.tri_material ? s.tri_material[q] : 0u;
    if (s.materials)
    {
        const float* m = s.materials + (size_t)mat * 8;
        o.albedo = mk3(m[0], m[1], m[2]); o.metallic = m[3]; o.roughness = max2(m[4], 0.1f);
    }
    else { o.albedo = mk3(0.8f, 0.8f, 0.8f); o.metallic = 0.0f; o.roughness = 0.5f; }
    if (s.mat_tex)
    {
        const int32_t* mt = s.mat_tex + (size_t)mat * 6;
        float tu = 0.0f, tv = 0.0f;
        if (s.uvs)
        {
            const float* qu = s.uvs + q * 6;
            tu = (qu[0] * b0 + qu[2] * b1) + qu[4] * b2;
            tv = (qu[1] * b0 + qu[3] * b1) + qu[5] * b2;
        }
        float c[4];
        if (mt[0] >= 0) { sample_texture(s, mt[0], tu, tv, c); o.albedo = mk3(c[0], c[1], c[2]); }
        if (mt[2] >= 0) { sample_texture(s, mt[2], tu, tv, c); o.roughness = max2(c[mt[4] & 3], 0.1f); }
        if (mt[3] >= 0) { sample_texture(s, mt[3], tu, tv, c); o.metallic = c[mt[5] & 3]; }
        if (mt[1] >= 0)
        {
            f3 tg = mk3(1.0f, 0.0f, 0.0f);
            if (s.tangents)
            {
                const float* qt = s.tangents + q * 9;
                tg = add3(add3(scale3(mk3(qt[0], qt[1], qt[2]), b0), scale3(mk3(qt[3], qt[4], qt[5]), b1)), scale3(mk3(qt[6], qt[7], qt[8]), b2));
            }
            tg = ir ? normalize3(inst_mul3(ir->m, normalize3(tg))) : normalize3(normalize3(tg));   // interpolated_vertex, then transform_vertex
            const f3 T = normalize3(tg), Nn = normalize3(o.N);     // get_normal_from_map: TBN = (T, T, N)
            sample_texture(s, mt[1], tu, tv, c);
            const f3 tn = normalize3(sub3(scale3(mk3(c[0], c[1], c[2]), 2.0f), one3()));
            o.N = normalize3(add3(add3(scale3(T, tn.x), scale3(T, tn.y)), scale3(Nn, tn.z)));
        }
    }
    return o;
}

// fetch_light_properties without SOFT_SHADOWS (lighting.glsl:6-111)
HR_DEV void fetch_light_hard(const hr_light& L, f3 Wo, f3 P, f3 N, f3& Li, f3& Wi, f3& Wh, float& t_max, float& attenuation)
{
    const int type = (int)L.data3[0];
    const f3  ldir = mk3(L.data0[0], L.data0[1], L.data0[2]);
    Li = scale3(mk3(L.data2[0], L.data2[1], L.data2[2]), L.data0[3]);
    if (type == 0) { Wi = ldir; t_max = 10000.0f; attenuation = 1.0f; }
    else
    {
        const f3 to_light = sub3(mk3(L.data1[0], L.data1[1], L.data1[2]), P);
        Wi                = normalize3(to_light);
        const float dist  = len3(to_light);
        t_max             = dist;
        if (type == 1) attenuation = __fdiv_rn(1.0f, dist * dist);
        else attenuation = __fdiv_rn(smoothstep1(L.data3[1], L.data3[2], dot3(Wi, ldir)), dist * dist);
    }
    Wh          = normalize3(add3(Wo, Wi));
    attenuation = attenuation * clamp1(dot3(N, Wi), 0.0f, 1.0f);
}

// (shared by the shadow pass and the ground-truth path tracer)
// lighting.glsl:6-111 with SOFT_SHADOWS | SHADOW_RAY_ONLY | RAY_TRACING
HR_DEV void fetch_light_shadow(const hr_light& L, f3 P, f3 N, float rx, float ry, f3& Wi, float& t_max, float& attenuation)
{
    const int type = (int)L.data3[0];
    const f3  ldir = mk3(L.data0[0], L.data0[1], L.data0[2]);
    f3        light_dir;
    float     radius;
    if (type == 0)
    {
        light_dir   = ldir;
        radius      = L.data1[3];
        t_max       = 10000.0f;
        attenuation = 1.0f;
    }
    else
    {
        f3    to_light = sub3(mk3(L.data1[0], L.data1[1], L.data1[2]), P);
        light_dir      = normalize3(to_light);
        float dist     = len3(to_light);
        radius         = __fdiv_rn(L.data1[3], dist);
        t_max          = dist;
        attenuation    = __fdiv_rn(1.0f, dist * dist);
    }
    f3    tangent   = normalize3(cross3(light_dir, mk3(0.0f, 1.0f, 0.0f)));
    f3    bitangent = normalize3(cross3(tangent, light_dir));
    float pr        = radius * hr_sqrt(rx);
    float pa        = ry * 2.0f * HR_M_PI;
    float s, c;
    det_sincos(pa, s, c);
    float dx = pr * c, dy = pr * s;
    Wi       = normalize3(add3(add3(light_dir, scale3(tangent, dx)), scale3(bitangent, dy)));
    if (type == 2)
    {
        float aa    = smoothstep1(L.data3[1], L.data3[2], dot3(Wi, ldir));
        attenuation = __fdiv_rn(aa, t_max * t_max);
    }
    attenuation = attenuation * clamp1(dot3(N, Wi), 0.0f, 1.0f);
}

struct TraceCtx
{
    const Node8*  nodes;
    const TriGPU* tris;
    uint32_t*     wave_stack;
    int           lane;
    DivCounters*  dv = nullptr;   // developer instrumentation (traverse.h)
    uint32_t      nn = 0, nt = 0; // direct_lighting<true>: node steps / triangle tests of the light and sky rays (hr_*_trace_stats)
};

// direct_lighting (lighting.glsl:117-196)
template <bool STATS = false>
HR_DEV f3 direct_lighting(TraceCtx& tc, const hr_light& light, f3 Wo, f3 N, f3 P, f3 F0, f3 diffuse_color, float roughness, f3 T,
                          bool sample_sky, float r2x, float r2y, const CubeMap& sky, uint32_t& rays)
{
    f3       Lo = mk3(0.0f, 0.0f, 0.0f);
    const f3 ray_origin = add3(P, scale3(N, 0.1f));
    uint32_t nn = 0, nt = 0;
    {
        f3    Li, Wi, Wh;
        float t_max, attenuation;
        fetch_light_hard(light, Wo, P, N, Li, Wi, Wh, t_max, attenuation);
#ifndef HR_ABL_NO_SECONDARY   // developer ablation (tools/ablate.sh)
        if (attenuation > 0.0f)
        {
            rays++;
            attenuation = attenuation * (trace_any<STATS>(tc.nodes, tc.tris, ray_origin, Wi, 0.01f, t_max, tc.wave_stack, tc.lane, nn, nt, 0u, tc.dv) ? 0.0f : 1.0f);
        }
#endif
        const f3 brdf = evaluate_uber_brdf(diffuse_color, roughness, N, F0, Wo, Wh, Wi);
        Lo = add3(Lo, mul3(scale3(mul3(T, brdf), attenuation), Li));
    }
    if (sample_sky)
    {
        const f3 Wi = sample_cosine_lobe_n(N, r2x, r2y);
        f3       Li = sky.fetch(Wi);
        const f3 Wh = normalize3(add3(Wo, Wi));
#ifndef HR_ABL_NO_SECONDARY
        rays++;
        Li = scale3(Li, trace_any<STATS>(tc.nodes, tc.tris, ray_origin, Wi, 0.01f, 10000.0f, tc.wave_stack, tc.lane, nn, nt, 0u, tc.dv) ? 0.0f : 1.0f);
#endif
        const f3 brdf = evaluate_uber_brdf(diffuse_color, roughness, N, F0, Wo, Wh, Wi);
        Lo = add3(Lo, mul3(mul3(T, brdf), Li));
    }
    if (STATS) { tc.nn += nn; tc.nt += nt; }
    return Lo;
}

#ifdef HR_DEV_PATHS   // only the A/B paths that lost use it (ddgi.hip wavefront kernels, -DDDGI_SEQ)
// direct_lighting split for the wavefront path (trace_queue.h): everything but the two visibility rays.  With occlusion o1 / o2 of
// the light ray and the sky ray known, direct_lighting's result is, operation for operation,
//     Lo = (ray1 && o1) ? 0 : P1;   if (sky && !o2) Lo = Lo + P2;
// (a factor 0 turns the term into a signed zero, and x + (+-0) == x for every x the first line can produce).
struct DirectSplit
{
    f3    origin;        // of both rays
    f3    Wi1, Wi2;
    float t_max1;
    bool  ray1;          // the light ray is traced (attenuation > 0)
    f3    P1, P2;
};
HR_DEV DirectSplit direct_lighting_split(const hr_light& light, f3 Wo, f3 N, f3 P, f3 F0, f3 diffuse_color, float roughness, f3 T, bool sample_sky,
                                         float r2x, float r2y, const CubeMap& sky)
{
    DirectSplit s;
    s.origin = add3(P, scale3(N, 0.1f));
    {
        f3    Li, Wi, Wh;
        float t_max, attenuation;
        fetch_light_hard(light, Wo, P, N, Li, Wi, Wh, t_max, attenuation);
        s.ray1 = attenuation > 0.0f;
        s.Wi1 = Wi; s.t_max1 = t_max;
        if (s.ray1) attenuation = attenuation * 1.0f;
        const f3 brdf = evaluate_uber_brdf(diffuse_color, roughness, N, F0, Wo, Wh, Wi);
        s.P1 = add3(mk3(0.0f, 0.0f, 0.0f), mul3(scale3(mul3(T, brdf), attenuation), Li));
    }
    s.Wi2 = mk3(0.0f, 0.0f, 1.0f); s.P2 = mk3(0.0f, 0.0f, 0.0f);
    if (sample_sky)
    {
        const f3 Wi = sample_cosine_lobe_n(N, r2x, r2y);
        const f3 Li = scale3(sky.fetch(Wi), 1.0f);
        const f3 Wh = normalize3(add3(Wo, Wi));
        const f3 brdf = evaluate_uber_brdf(diffuse_color, roughness, N, F0, Wo, Wh, Wi);
        s.Wi2 = Wi;
        s.P2  = mul3(mul3(T, brdf), Li);
    }
    return s;
}
#endif // HR_DEV_PATHS

} // namespace hr

#ifdef HR_PROBE_FAST_HIT
#undef __fdiv_rn
#undef hr_sqrt
#undef normalize3
#endif
