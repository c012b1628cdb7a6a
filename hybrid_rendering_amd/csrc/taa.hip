// TemporalAA on MI355X — HIP replacement for src/temporal_aa.cpp (update :64-81, render :84-172) and
// shaders/taa.comp (main :378-420, temporal_reprojection :245-372, clip_aabb :123-151, find_closest_fragment_3x3
// :155-187; build flags USE_DILATION, MINMAX_3X3_ROUNDED, USE_CLIPPING, UNJITTER_*, HDR_CORRECTION).  SURVEY.md §8f row 4.
// Samplers pinned as in the oracle: colour / history bilinear, clamp-to-edge, fp32 weights; G-buffer nearest.
#include "hr_internal.h"
#include "device_math.h"

using namespace hr;

namespace {

struct TexRGBA
{
    const uint2* p; int w, h;
    HR_DEV f4 texel(int x, int y) const
    {
        x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
        y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
        const uint2 q = p[(size_t)y * w + x];
        f4 r; r.x = h2f_lo(q.x); r.y = h2f_hi(q.x); r.z = h2f_lo(q.y); r.w = h2f_hi(q.y);
        return r;
    }
    HR_DEV f4 nearest(float u, float v) const { return texel((int)floorf(u * (float)w), (int)floorf(v * (float)h)); }
};

HR_DEV f4 add4(f4 a, f4 b) { f4 r; r.x = a.x + b.x; r.y = a.y + b.y; r.z = a.z + b.z; r.w = a.w + b.w; return r; }
HR_DEV f4 sub4(f4 a, f4 b) { f4 r; r.x = a.x - b.x; r.y = a.y - b.y; r.z = a.z - b.z; r.w = a.w - b.w; return r; }
HR_DEV f4 scale4(f4 a, float s) { f4 r; r.x = a.x * s; r.y = a.y * s; r.z = a.z * s; r.w = a.w * s; return r; }
HR_DEV f4 div4s(f4 a, float s) { f4 r; r.x = __fdiv_rn(a.x, s); r.y = __fdiv_rn(a.y, s); r.z = __fdiv_rn(a.z, s); r.w = __fdiv_rn(a.w, s); return r; }
HR_DEV f4 min4(f4 a, f4 b) { f4 r; r.x = min2(a.x, b.x); r.y = min2(a.y, b.y); r.z = min2(a.z, b.z); r.w = min2(a.w, b.w); return r; }
HR_DEV f4 max4(f4 a, f4 b) { f4 r; r.x = max2(a.x, b.x); r.y = max2(a.y, b.y); r.z = max2(a.z, b.z); r.w = max2(a.w, b.w); return r; }
HR_DEV f4 mix4(f4 a, f4 b, float t) { return add4(scale4(a, 1.0f - t), scale4(b, t)); }

HR_DEV f4 bilinear(const TexRGBA& t, float u, float v)
{
    const float fx = u * (float)t.w - 0.5f, fy = v * (float)t.h - 0.5f;
    const float x0 = floorf(fx), y0 = floorf(fy);
    const float ax = fx - x0, ay = fy - y0;
    const int   ix = (int)x0, iy = (int)y0;
    const f4 t00 = t.texel(ix, iy), t10 = t.texel(ix + 1, iy), t01 = t.texel(ix, iy + 1), t11 = t.texel(ix + 1, iy + 1);
    return mix4(mix4(t00, t10, ax), mix4(t01, t11, ax), ay);
}

HR_DEV float depth_nearest(const float* p, int w, int h, float u, float v)
{
    int x = (int)floorf(u * (float)w), y = (int)floorf(v * (float)h);
    x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
    y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
    return p[(size_t)y * w + x];
}

HR_DEV f4 clip_aabb(f4 aabb_min, f4 aabb_max, f4 p, f4 q)
{
    f4 r = sub4(q, p);
    const float rmaxx = aabb_max.x - p.x, rmaxy = aabb_max.y - p.y, rmaxz = aabb_max.z - p.z;
    const float rminx = aabb_min.x - p.x, rminy = aabb_min.y - p.y, rminz = aabb_min.z - p.z;
    const float eps = 0.00000001f;
    if (r.x > rmaxx + eps) r = scale4(r, __fdiv_rn(rmaxx, r.x));
    if (r.y > rmaxy + eps) r = scale4(r, __fdiv_rn(rmaxy, r.y));
    if (r.z > rmaxz + eps) r = scale4(r, __fdiv_rn(rmaxz, r.z));
    if (r.x < rminx - eps) r = scale4(r, __fdiv_rn(rminx, r.x));
    if (r.y < rminy - eps) r = scale4(r, __fdiv_rn(rminy, r.y));
    if (r.z < rminz - eps) r = scale4(r, __fdiv_rn(rminz, r.z));
    return add4(p, r);
}

struct TAAArgs
{
    TexRGBA      cur, prev, vel;
    const float* depth;
    uint2*       out;
    int          w, h;
    float        tsx, tsy, jx, jy, feedback_min, feedback_max;
    int          sharpen;
};

struct ToneMapArgs
{
    TexRGBA src;
    float4* out_f;
    uchar4* out_b;
    int     single_channel;
    float   exposure;
};
HR_DEV float aces_film(float x) // tone_map.frag:36-44
{
    const float num = x * (2.51f * x + 0.03f), den = x * (2.43f * x + 0.59f) + 0.14f;
    return clamp1(__fdiv_rn(num, den), 0.0f, 1.0f);
}
__global__ __launch_bounds__(256) void k_tone_map(ToneMapArgs a)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= a.src.w || y >= a.src.h) return;
    const f4 c = bilinear(a.src, __fdiv_rn((float)x + 0.5f, (float)a.src.w), __fdiv_rn((float)y + 0.5f, (float)a.src.h));
    float r, g, b;
    if (a.single_channel == 1) r = g = b = c.x;
    else
    {
        const float ig = __fdiv_rn(1.0f, 2.2f);
        r = det_pow(aces_film(c.x * a.exposure), ig);
        g = det_pow(aces_film(c.y * a.exposure), ig);
        b = det_pow(aces_film(c.z * a.exposure), ig);
    }
    const size_t i = (size_t)y * a.src.w + x;
    if (a.out_f) a.out_f[i] = make_float4(r, g, b, 1.0f);
    if (a.out_b)
    {
        auto q = [](float v) { v = clamp1(v, 0.0f, 1.0f); return (unsigned char)floorf(v * 255.0f + 0.5f); };
        a.out_b[i] = make_uchar4(q(r), q(g), q(b), 255);
    }
}

__global__ __launch_bounds__(256) void k_taa(TAAArgs a)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= a.w || y >= a.h) return;
    const float tcx = ((float)x + 0.5f) * a.tsx, tcy = ((float)y + 0.5f) * a.tsy;
    const float uvx = tcx + a.jx, uvy = tcy + a.jy;
    const float ddx = fabsf(a.tsx), ddy = fabsf(a.tsy);
    // find_closest_fragment_3x3: row-major scan, strict '>' keeps the first minimum
    float dminx = -1.0f, dminy = -1.0f, dminz = depth_nearest(a.depth, a.w, a.h, uvx - ddx, uvy - ddy);
#pragma unroll
    for (int k = 1; k < 9; k++)
    {
        const int   ox = k % 3 - 1, oy = k / 3 - 1;
        const float sx = ox < 0 ? uvx - ddx : (ox > 0 ? uvx + ddx : uvx), sy = oy < 0 ? uvy - ddy : (oy > 0 ? uvy + ddy : uvy);
        const float z = depth_nearest(a.depth, a.w, a.h, sx, sy);
        if (dminz > z) { dminx = (float)ox; dminy = (float)oy; dminz = z; }
    }
    const float cfx = uvx + ddx * dminx, cfy = uvy + ddy * dminy;
    const f4    v   = a.vel.nearest(cfx, cfy);
    const float svx = v.z, svy = v.w;
    f4 texel0 = bilinear(a.cur, tcx + a.jx, tcy + a.jy);
    f4 texel1 = bilinear(a.prev, tcx + svx, tcy + svy);
    const float ux = tcx + a.jx, uy = tcy + a.jy;
    const f4 ctl = bilinear(a.cur, ux - a.tsx, uy - a.tsy), ctc = bilinear(a.cur, ux, uy - a.tsy), ctr = bilinear(a.cur, ux + a.tsx, uy - a.tsy);
    const f4 cml = bilinear(a.cur, ux - a.tsx, uy), cmc = bilinear(a.cur, ux, uy), cmr = bilinear(a.cur, ux + a.tsx, uy);
    const f4 cbl = bilinear(a.cur, ux - a.tsx, uy + a.tsy), cbc = bilinear(a.cur, ux, uy + a.tsy), cbr = bilinear(a.cur, ux + a.tsx, uy + a.tsy);
    f4 cmin = min4(ctl, min4(ctc, min4(ctr, min4(cml, min4(cmc, min4(cmr, min4(cbl, min4(cbc, cbr))))))));
    f4 cmax = max4(ctl, max4(ctc, max4(ctr, max4(cml, max4(cmc, max4(cmr, max4(cbl, max4(cbc, cbr))))))));
    f4 cavg = div4s(add4(add4(add4(add4(add4(add4(add4(add4(ctl, ctc), ctr), cml), cmc), cmr), cbl), cbc), cbr), 9.0f);
    const f4 cmin5 = min4(ctc, min4(cml, min4(cmc, min4(cmr, cbc))));
    const f4 cmax5 = max4(ctc, max4(cml, max4(cmc, max4(cmr, cbc))));
    const f4 cavg5 = div4s(add4(add4(add4(add4(ctc, cml), cmc), cmr), cbc), 5.0f);
    cmin = scale4(add4(cmin, cmin5), 0.5f);
    cmax = scale4(add4(cmax, cmax5), 0.5f);
    cavg = scale4(add4(cavg, cavg5), 0.5f);
    texel1 = clip_aabb(cmin, cmax, min4(max4(cavg, cmin), cmax), texel1);
    const float lum0 = luminance(mk3(texel0.x, texel0.y, texel0.z)), lum1 = luminance(mk3(texel1.x, texel1.y, texel1.z));
    const float unbiased_diff = __fdiv_rn(fabsf(lum0 - lum1), max2(lum0, max2(lum1, 0.2f)));
    const float uw = 1.0f - unbiased_diff, uw2 = uw * uw;
    const float k_feedback = mix1(a.feedback_min, a.feedback_max, uw2);
    if (a.sharpen == 1)
    {
        f4 sum; sum.x = sum.y = sum.z = sum.w = 0.0f;
        sum = add4(sum, scale4(cml, -1.0f));
        sum = add4(sum, scale4(ctc, -1.0f));
        sum = add4(sum, scale4(texel0, 5.0f));
        sum = add4(sum, scale4(cbc, -1.0f));
        sum = add4(sum, scale4(cmr, -1.0f));
        texel0 = sum;
    }
    const float t0[3] = { __fdiv_rn(texel0.x, texel0.x + 1.0f), __fdiv_rn(texel0.y, texel0.y + 1.0f), __fdiv_rn(texel0.z, texel0.z + 1.0f) };
    const float t1[3] = { __fdiv_rn(texel1.x, texel1.x + 1.0f), __fdiv_rn(texel1.y, texel1.y + 1.0f), __fdiv_rn(texel1.z, texel1.z + 1.0f) };
    float o[3];
#pragma unroll
    for (int c = 0; c < 3; c++)
    {
        float b = mix1(t0[c], t1[c], k_feedback);
        b       = __fdiv_rn(b, max2(1.0f - b, 0.00000001f));
        o[c]    = clamp1(b, 0.0f, 1.0f);
    }
    a.out[(size_t)y * a.w + x] = make_uint2(pack_h2(o[0], o[1]), pack_h2(o[2], 1.0f));
}

// temporal_aa.cpp:30-43
float halton_sequence(int base, int index)
{
    float result = 0.0f, f = 1.0f;
    while (index > 0)
    {
        f /= (float)base;
        result += f * (float)(index % base);
        index = (int)floorf((float)index / (float)base);
    }
    return result;
}

} // namespace

struct hr_taa
{
    hr_ctx* ctx = nullptr;
    int     w = 0, h = 0;
    DevBuf  image[2];
    float   current_jitter[2] = { 0.0f, 0.0f }, prev_jitter[2] = { 0.0f, 0.0f };
    float   jitter_samples[16][2];
    StageProfiler prof;
};

extern "C" {

void hr_taa_default_params(hr_taa_params* p)
{
    p->enabled = 1; p->sharpen = 1; p->reset = 1; // temporal_aa.h:55-57
    p->feedback_min = 0.88f; p->feedback_max = 0.97f;
}

hr_status hr_taa_create(hr_ctx* ctx, int32_t width, int32_t height, hr_taa** out)
{
    HR_CHECK_ARG(ctx && out && width > 0 && height > 0);
    HR_HIP(hipSetDevice(ctx->device));
    hr_taa* p = new hr_taa();
    p->ctx = ctx; p->w = width; p->h = height;
    hr_status s;
    for (int i = 0; i < 2; i++)
    {
        if ((s = p->image[i].alloc((size_t)width * height * 8)) != HR_OK) { delete p; return s; }
        HR_HIP(hipMemset(p->image[i].p, 0, p->image[i].bytes));
    }
    for (int i = 1; i <= 16; i++) // temporal_aa.cpp:54-55
    {
        p->jitter_samples[i - 1][0] = 2.0f * halton_sequence(2, i) - 1.0f;
        p->jitter_samples[i - 1][1] = 2.0f * halton_sequence(3, i) - 1.0f;
    }
    *out = p;
    return HR_OK;
}

hr_status hr_taa_destroy(hr_taa* p)
{
    if (!p) return HR_OK;
    (void)hipSetDevice(p->ctx->device);
    (void)hipDeviceSynchronize();
    delete p;
    return HR_OK;
}

hr_status hr_taa_update(hr_taa* p, uint32_t num_frames, const hr_taa_params* prm, float* current_prev_jitter)
{
    HR_CHECK_ARG(p && prm);
    if (prm->enabled)
    {
        p->prev_jitter[0] = p->current_jitter[0]; p->prev_jitter[1] = p->current_jitter[1];
        const float* hs = p->jitter_samples[num_frames % 16u];
        p->current_jitter[0] = hs[0] / (float)p->w;
        p->current_jitter[1] = hs[1] / (float)p->h;
    }
    else p->prev_jitter[0] = p->prev_jitter[1] = p->current_jitter[0] = p->current_jitter[1] = 0.0f;
    if (current_prev_jitter)
    {
        current_prev_jitter[0] = p->current_jitter[0]; current_prev_jitter[1] = p->current_jitter[1];
        current_prev_jitter[2] = p->prev_jitter[0]; current_prev_jitter[3] = p->prev_jitter[1];
    }
    return HR_OK;
}

hr_status hr_taa_set_profiling(hr_taa* p, int32_t e) { HR_CHECK_ARG(p); p->prof.enabled = e != 0; return HR_OK; }
hr_status hr_taa_get_stage_times(hr_taa* p, hr_stage_times* out) { HR_CHECK_ARG(p && out); p->prof.collect(out); return HR_OK; }

hr_status hr_taa_render(hr_taa* p, const hr_image_view* color, const hr_gbuffer_level* g, int32_t ping_pong, const hr_taa_params* prm, void* stream_)
{
    HR_CHECK_ARG(p && color && g && prm);
    if (!prm->enabled) return HR_OK; // temporal_aa.cpp:94
    HR_CHECK_ARG(color->data && color->format == HR_FORMAT_RGBA16F && color->width == p->w && color->height == p->h);
    HR_CHECK_ARG(g->gb2 && g->depth && g->width == p->w && g->height == p->h);
    HR_HIP(hipSetDevice(p->ctx->device));
    hipStream_t st = (hipStream_t)stream_;
    p->prof.begin_frame();
    const int write_idx = ping_pong ? 1 : 0, read_idx = ping_pong ? 0 : 1;
    // m_reset is set by the constructor and by the GUI but never cleared (temporal_aa.cpp:112,184): upstream the history image
    // is re-seeded from the current frame on EVERY frame.  reset = 1 reproduces that; reset = 0 keeps a real history.
    if (prm->reset) HR_HIP(hipMemcpyAsync(p->image[read_idx].p, color->data, (size_t)p->w * p->h * 8, hipMemcpyDeviceToDevice, st));
    TAAArgs a;
    a.cur  = TexRGBA { (const uint2*)color->data, p->w, p->h };
    a.prev = TexRGBA { (const uint2*)p->image[read_idx].p, p->w, p->h };
    a.vel  = TexRGBA { (const uint2*)g->gb2, p->w, p->h };
    a.depth = g->depth; a.out = (uint2*)p->image[write_idx].p; a.w = p->w; a.h = p->h;
    a.tsx = 1.0f / (float)p->w; a.tsy = 1.0f / (float)p->h;
    a.jx = p->current_jitter[0]; a.jy = p->current_jitter[1];
    a.feedback_min = prm->feedback_min; a.feedback_max = prm->feedback_max; a.sharpen = prm->sharpen ? 1 : 0;
    int ev = p->prof.begin("taa", st, (uint64_t)p->w * p->h * 36);
    hipLaunchKernelGGL(k_taa, dim3(cdiv(p->w, 32), cdiv(p->h, 8)), dim3(256), 0, st, a);
    p->prof.end(ev, st);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

// ToneMap::render (tone_map.cpp:98-143, shaders/tone_map.frag:50-68): a stateless full-screen pass over the (TAA) colour
// image — exposure, ACES film curve, pow(1/2.2) — read through the same bilinear sampler at the pixel centres; out_rgba32f
// (nullable) receives FS_OUT_Color, out_rgba8 (nullable) its UNORM8 conversion floor(c * 255 + 0.5) (swap-chain write).
hr_status hr_tone_map(hr_ctx* ctx, const hr_image_view* color, int32_t single_channel, float exposure, float* out_rgba32f, uint8_t* out_rgba8,
                      void* stream_)
{
    HR_CHECK_ARG(ctx && color && color->data && color->format == HR_FORMAT_RGBA16F && color->width > 0 && color->height > 0 && (out_rgba32f || out_rgba8));
    HR_HIP(hipSetDevice(ctx->device));
    ToneMapArgs a;
    a.src = TexRGBA { (const uint2*)color->data, color->width, color->height };
    a.out_f = (float4*)out_rgba32f; a.out_b = (uchar4*)out_rgba8; a.single_channel = single_channel; a.exposure = exposure;
    hipLaunchKernelGGL(k_tone_map, dim3(cdiv(color->width, 32), cdiv(color->height, 8)), dim3(256), 0, (hipStream_t)stream_, a);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

// TemporalAA::output_ds: m_read_ds[ping_pong] = the image written this frame
hr_status hr_taa_output(hr_taa* p, int32_t ping_pong, hr_image_view* v)
{
    HR_CHECK_ARG(p && v);
    v->data = p->image[ping_pong ? 1 : 0].p; v->width = p->w; v->height = p->h; v->row_pitch_bytes = p->w * 8; v->format = HR_FORMAT_RGBA16F;
    return HR_OK;
}

} // extern "C"
