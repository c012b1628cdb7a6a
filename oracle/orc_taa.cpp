// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).  Pinned against the reference's own shaders: tests/test_ref_shaders.py.
// Restatement of the temporal anti-aliasing resolve:
//   taa.comp:378-420 main, :245-372 temporal_reprojection (defines USE_DILATION, MINMAX_3X3_ROUNDED, USE_CLIPPING,
//   UNJITTER_*, HDR_CORRECTION; no YCoCg, no USE_OPTIMIZATIONS), :123-151 clip_aabb, :155-187 find_closest_fragment_3x3
//   temporal_aa.cpp:30-43 halton_sequence, :64-81 update (jitter)
// Pinned sampler behaviour (the reference leaves it to Vulkan): s_Current / s_Prev are bilinear (temporal_aa.cpp:255),
// clamp-to-edge, fp32 weights mix(mix(t00,t10,fx), mix(t01,t11,fx), fy) at uv*size - 0.5; the G-buffer samplers
// (velocity = GB2.zw, depth) are nearest, clamp-to-edge.
#include "orc_api.h"
#include "orc_common.h"

using namespace orc;

namespace {

struct V4 { float x, y, z, w; };
inline V4 add(V4 a, V4 b) { return V4 { a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w }; }
inline V4 sub(V4 a, V4 b) { return V4 { a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w }; }
inline V4 scale(V4 a, float s) { return V4 { a.x * s, a.y * s, a.z * s, a.w * s }; }
inline V4 divs(V4 a, float s) { return V4 { a.x / s, a.y / s, a.z / s, a.w / s }; }
inline V4 vmin(V4 a, V4 b) { return V4 { fmin2(a.x, b.x), fmin2(a.y, b.y), fmin2(a.z, b.z), fmin2(a.w, b.w) }; }
inline V4 vmax(V4 a, V4 b) { return V4 { fmax2(a.x, b.x), fmax2(a.y, b.y), fmax2(a.z, b.z), fmax2(a.w, b.w) }; }

struct Tex4 // RGBA16F
{
    const uint16_t* p; int w, h;
    V4 texel(int x, int y) const
    {
        x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
        y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
        const uint16_t* q = p + ((size_t)y * w + x) * 4;
        return V4 { f16_to_f32(q[0]), f16_to_f32(q[1]), f16_to_f32(q[2]), f16_to_f32(q[3]) };
    }
    V4 bilinear(float u, float v) const
    {
        const float fx = u * (float)w - 0.5f, fy = v * (float)h - 0.5f;
        const float x0 = std::floor(fx), y0 = std::floor(fy);
        const float ax = fx - x0, ay = fy - y0;
        const int   ix = (int)x0, iy = (int)y0;
        const V4 t00 = texel(ix, iy), t10 = texel(ix + 1, iy), t01 = texel(ix, iy + 1), t11 = texel(ix + 1, iy + 1);
        auto mix4 = [](V4 a, V4 b, float t) { return add(scale(a, 1.0f - t), scale(b, t)); };
        return mix4(mix4(t00, t10, ax), mix4(t01, t11, ax), ay);
    }
    V4 nearest(float u, float v) const { return texel((int)std::floor(u * (float)w), (int)std::floor(v * (float)h)); }
};

struct Tex1 // R32F
{
    const float* p; int w, h;
    float nearest(float u, float v) const
    {
        int x = (int)std::floor(u * (float)w), y = (int)std::floor(v * (float)h);
        x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
        y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
        return p[(size_t)y * w + x];
    }
};

inline float lum(V4 c) { return fmax2((c.x * 0.299f + c.y * 0.587f) + c.z * 0.114f, 0.0001f); } // common.glsl:141-144

// taa.comp:123-151, the non-optimised variant
inline V4 clip_aabb(V4 aabb_min, V4 aabb_max, V4 p, V4 q)
{
    V4 r = sub(q, p);
    const float rmaxx = aabb_max.x - p.x, rmaxy = aabb_max.y - p.y, rmaxz = aabb_max.z - p.z;
    const float rminx = aabb_min.x - p.x, rminy = aabb_min.y - p.y, rminz = aabb_min.z - p.z;
    const float eps = 0.00000001f;
    if (r.x > rmaxx + eps) r = scale(r, rmaxx / r.x);
    if (r.y > rmaxy + eps) r = scale(r, rmaxy / r.y);
    if (r.z > rmaxz + eps) r = scale(r, rmaxz / r.z);
    if (r.x < rminx - eps) r = scale(r, rminx / r.x);
    if (r.y < rminy - eps) r = scale(r, rminy / r.y);
    if (r.z < rminz - eps) r = scale(r, rminz / r.z);
    return add(p, r);
}

} // namespace

extern "C" {

// temporal_aa.cpp:30-43 (float arithmetic as written)
float orc_halton(int base, int index)
{
    float result = 0.0f, f = 1.0f;
    while (index > 0)
    {
        f /= (float)base;
        result += f * (float)(index % base);
        index = (int)std::floor((float)index / (float)base);
    }
    return result;
}

// TemporalAA::update (temporal_aa.cpp:64-81): jitter[4] = (current.xy, prev.xy); prev_current = last frame's current
void orc_taa_jitter(uint32_t num_frames, int w, int h, int enabled, const float* prev_current, float* jitter)
{
    if (!enabled) { jitter[0] = jitter[1] = jitter[2] = jitter[3] = 0.0f; return; }
    jitter[2] = prev_current[0]; jitter[3] = prev_current[1];
    const int i = (int)(num_frames % 16u) + 1; // m_jitter_samples[k] = halton(k + 1)
    jitter[0] = (2.0f * orc_halton(2, i) - 1.0f) / (float)w;
    jitter[1] = (2.0f * orc_halton(3, i) - 1.0f) / (float)h;
}

// color / prev / out: [h][w][4] fp16; gb2: [h][w][4] fp16 (zw = motion); depth: [h][w] f32
void orc_taa_resolve(int w, int h, const uint16_t* color, const uint16_t* prev, const uint16_t* gb2, const float* depth, const float* jitter,
                     float feedback_min, float feedback_max, int sharpen, uint16_t* out)
{
    const Tex4 cur { color, w, h }, prv { prev, w, h }, vel { gb2, w, h };
    const Tex1 dep { depth, w, h };
    const float tsx = 1.0f / (float)w, tsy = 1.0f / (float)h; // temporal_aa.cpp:127
    const float jx = jitter[0], jy = jitter[1];
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            const float tcx = ((float)x + 0.5f) * tsx, tcy = ((float)y + 0.5f) * tsy;
            const float uvx = tcx + jx, uvy = tcy + jy;
            // find_closest_fragment_3x3
            const float ddx = std::fabs(tsx), ddy = std::fabs(tsy);
            float dminx = -1.0f, dminy = -1.0f, dminz = dep.nearest((uvx - 0.0f) - ddx, (uvy - ddy) - 0.0f);
            {
                const float ox[9] = { -1, 0, 1, -1, 0, 1, -1, 0, 1 }, oy[9] = { -1, -1, -1, 0, 0, 0, 1, 1, 1 };
                for (int k = 1; k < 9; k++)
                {
                    // uv -/+ dv -/+ du: vec2 arithmetic, the unused component adds 0.0
                    float sx = uvx, sy = uvy;
                    if (oy[k] < 0) { sx = sx - 0.0f; sy = sy - ddy; } else if (oy[k] > 0) { sx = sx + 0.0f; sy = sy + ddy; }
                    if (ox[k] < 0) { sx = sx - ddx; sy = sy - 0.0f; } else if (ox[k] > 0) { sx = sx + ddx; sy = sy + 0.0f; }
                    const float z = dep.nearest(sx, sy);
                    if (dminz > z) { dminx = ox[k]; dminy = oy[k]; dminz = z; }
                }
            }
            const float cfx = uvx + ddx * dminx, cfy = uvy + ddy * dminy;
            const V4    v   = vel.nearest(cfx, cfy);
            const float svx = v.z, svy = v.w;
            // temporal_reprojection(tex_coord, ss_vel, vs_dist)
            V4 texel0 = cur.bilinear(tcx + jx, tcy + jy);
            V4 texel1 = prv.bilinear(tcx + svx, tcy + svy);
            const float ux = tcx + jx, uy = tcy + jy;
            const V4 ctl = cur.bilinear((ux - 0.0f) - tsx, (uy - tsy) - 0.0f), ctc = cur.bilinear(ux - 0.0f, uy - tsy), ctr = cur.bilinear((ux - 0.0f) + tsx, (uy - tsy) + 0.0f);
            const V4 cml = cur.bilinear(ux - tsx, uy - 0.0f), cmc = cur.bilinear(ux, uy), cmr = cur.bilinear(ux + tsx, uy + 0.0f);
            const V4 cbl = cur.bilinear((ux + 0.0f) - tsx, (uy + tsy) - 0.0f), cbc = cur.bilinear(ux + 0.0f, uy + tsy), cbr = cur.bilinear((ux + 0.0f) + tsx, (uy + tsy) + 0.0f);
            V4 cmin = vmin(ctl, vmin(ctc, vmin(ctr, vmin(cml, vmin(cmc, vmin(cmr, vmin(cbl, vmin(cbc, cbr))))))));
            V4 cmax = vmax(ctl, vmax(ctc, vmax(ctr, vmax(cml, vmax(cmc, vmax(cmr, vmax(cbl, vmax(cbc, cbr))))))));
            V4 cavg = divs(add(add(add(add(add(add(add(add(ctl, ctc), ctr), cml), cmc), cmr), cbl), cbc), cbr), 9.0f);
            const V4 cmin5 = vmin(ctc, vmin(cml, vmin(cmc, vmin(cmr, cbc))));
            const V4 cmax5 = vmax(ctc, vmax(cml, vmax(cmc, vmax(cmr, cbc))));
            const V4 cavg5 = divs(add(add(add(add(ctc, cml), cmc), cmr), cbc), 5.0f);
            cmin = scale(add(cmin, cmin5), 0.5f);
            cmax = scale(add(cmax, cmax5), 0.5f);
            cavg = scale(add(cavg, cavg5), 0.5f);
            texel1 = clip_aabb(cmin, cmax, vmin(vmax(cavg, cmin), cmax), texel1);
            const float lum0 = lum(texel0), lum1 = lum(texel1);
            const float unbiased_diff = std::fabs(lum0 - lum1) / fmax2(lum0, fmax2(lum1, 0.2f));
            const float uw = 1.0f - unbiased_diff, uw2 = uw * uw;
            const float k_feedback = mixf(feedback_min, feedback_max, uw2);
            if (sharpen == 1)
            {
                V4 sum { 0.0f, 0.0f, 0.0f, 0.0f };
                sum = add(sum, scale(cml, -1.0f));
                sum = add(sum, scale(ctc, -1.0f));
                sum = add(sum, scale(texel0, 5.0f));
                sum = add(sum, scale(cbc, -1.0f));
                sum = add(sum, scale(cmr, -1.0f));
                texel0 = sum;
            }
            auto tm = [](float c) { return c / (c + 1.0f); };
            const float t0[3] = { tm(texel0.x), tm(texel0.y), tm(texel0.z) }, t1[3] = { tm(texel1.x), tm(texel1.y), tm(texel1.z) };
            uint16_t* o = out + ((size_t)y * w + x) * 4;
            for (int c = 0; c < 3; c++)
            {
                float b = mixf(t0[c], t1[c], k_feedback);
                b = b / fmax2(1.0f - b, 0.00000001f);
                o[c] = f32_to_f16(clampf(b, 0.0f, 1.0f));
            }
            o[3] = f32_to_f16(1.0f);
        }
}

// tone_map.frag:50-68 (ToneMap::render, tone_map.cpp:98-143).  color RGBA16F read through the bilinear sampler at the pixel
// centres (the descriptor is TemporalAA's output_ds, temporal_aa.cpp:255); out: [h][w][4] fp32 = FS_OUT_Color.
void orc_tone_map(int w, int h, const uint16_t* color, int single_channel, float exposure, float* out)
{
    const Tex4 src { color, w, h };
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            const V4 c = src.bilinear(((float)x + 0.5f) / (float)w, ((float)y + 0.5f) / (float)h);
            float rgb[3];
            if (single_channel == 1) rgb[0] = rgb[1] = rgb[2] = c.x;
            else
            {
                const float in[3] = { c.x * exposure, c.y * exposure, c.z * exposure };
                for (int k = 0; k < 3; k++)
                {
                    const float v = in[k];
                    const float aces = clampf((v * (2.51f * v + 0.03f)) / (v * (2.43f * v + 0.59f) + 0.14f), 0.0f, 1.0f);
                    rgb[k] = det_pow_auto(aces, 1.0f / 2.2f);
                }
            }
            float* o = out + ((size_t)y * w + x) * 4;
            o[0] = rgb[0]; o[1] = rgb[1]; o[2] = rgb[2]; o[3] = 1.0f;
        }
}

} // extern "C"
