// SVGF-style history reprojection shared by the three temporal kernels — device restatement of
// /root/reference/src/shaders/reprojection.glsl:115-328 (reproject), :52-67 (is_reprojection_valid),
// :71-111 (surface / virtual point reprojection).  Variants as template flags:
//   SINGLE  = REPROJECTION_SINGLE_COLOR_CHANNEL,  MOMENTS = REPROJECTION_MOMENTS,
//   REFL    = REPROJECTION_REFLECTIONS.
// Pinned behaviour: texel fetches outside an image return 0; ivec2(vec2)
// truncates toward zero (quirk 4).
#pragma once
#include "device_math.h"

namespace hr {

// image accessors: tightly packed row-major.  `p` is the address of the (possibly virtual) row 0 of
// the full-frame image; rows [y0, y1) are resident (whole frame: y0 = 0, y1 = height; a row band on
// one GPU of a tiled frame: band + halo).  Texels outside [0,w) x [y0,y1) read as 0.
// Branch-free: an outside tap loads the first resident texel and the result is replaced by 0 — a guarded load costs a
// saveexec / branch pair per tap and keeps the taps of a stencil from being issued together.
struct ImgRGBA16F
{
    const uint2* p;
    int          w, y0, y1;
    HR_DEV uint2 raw(int x, int y) const
    {
        const bool  ok = !(x < 0 || y < y0 || x >= w || y >= y1);
        const uint2 v  = p[ok ? (size_t)y * w + x : (size_t)y0 * w];
        return ok ? v : make_uint2(0u, 0u);
    }
};
struct ImgRG16F
{
    const uint32_t* p;
    int             w, y0, y1;
    HR_DEV uint32_t raw(int x, int y) const
    {
        const bool     ok = !(x < 0 || y < y0 || x >= w || y >= y1);
        const uint32_t v  = p[ok ? (size_t)y * w + x : (size_t)y0 * w];
        return ok ? v : 0u;
    }
};
struct ImgR16F
{
    const uint16_t* p;
    int             w, y0, y1;
    HR_DEV float fetch(int x, int y) const
    {
        const bool     ok = !(x < 0 || y < y0 || x >= w || y >= y1);
        const uint16_t v  = p[ok ? (size_t)y * w + x : (size_t)y0 * w];
        return ok ? h2f(v) : 0.0f;
    }
};
struct ImgR32F
{
    const float* p;
    int          w, y0, y1;
    HR_DEV float fetch(int x, int y) const
    {
        const bool  ok = !(x < 0 || y < y0 || x >= w || y >= y1);
        const float v  = p[ok ? (size_t)y * w + x : (size_t)y0 * w];
        return ok ? v : 0.0f;
    }
};

HR_DEV bool reprojection_valid(int cx, int cy, f3 cur_pos, f3 hist_pos, f3 cur_n, f3 hist_n, float cur_id, float hist_id, int w, int h)
{
    if (cx < 0 || cy < 0 || cx > w - 1 || cy > h - 1) return false;
    if (!(cur_id == hist_id)) return false;
    float dist = fabsf(dot3(sub3(cur_pos, hist_pos), cur_n));
    if (dist > 5.0f) return false;                 // PLANE_DISTANCE  reprojection.glsl:7
    float nd = fabsf(dot3(cur_n, hist_n));
    if (!(nd * nd > 0.1f)) return false;           // NORMAL_DISTANCE reprojection.glsl:6
    return true;
}

struct ReprojIn
{
    int          x, y;       // pixel in the (full) pass image
    float        depth;
    const float* vpi;        // view_proj_inverse (LDS or constant memory)
    // reflections only
    f3           cam_pos;
    const float* prev_vp;
    float        ray_length;
    ImgRGBA16F   gb2, gb3, pgb2, pgb3;
    ImgR32F      pdepth;
    int          w, h;
    // out: the bilinear footprint touched a row of the image that is not resident (row band: motion beyond the history apron)
    bool         apron_miss = false;
    // optional: the caller has already fetched / decoded the centre pixel (shadows: it also stores the decoded normal)
    bool         has_center = false;
    uint2        c2, c3;
    f3           cur_n;
};

// HistT: ImgRG16F (shadows: r = visibility), ImgR16F (AO), ImgRGBA16F (reflections rgb)
template <bool SINGLE, bool MOMENTS, bool REFL, typename HistT>
HR_DEV bool reproject(ReprojIn& in, const HistT& hist, const ImgRGBA16F& hist_moments, const ImgR16F& hist_length, float* hcol, float* hmom, float& history_length)
{
    const int   w = in.w, h = in.h;
    const float fw = (float)w, fh = (float)h;
    const float tu = __fdiv_rn((float)in.x + 0.5f, fw), tv = __fdiv_rn((float)in.y + 0.5f, fh);
    const uint2 c2 = in.has_center ? in.c2 : in.gb2.raw(in.x, in.y), c3 = in.has_center ? in.c3 : in.gb3.raw(in.x, in.y);
    const float mvx = h2f_lo(c2.y), mvy = h2f_hi(c2.y);
    const f3    cur_n   = in.has_center ? in.cur_n : oct_decode(h2f_lo(c2.x), h2f_hi(c2.x));
    const float cur_id  = h2f_lo(c3.y);
    const f3    cur_pos = world_pos_from_depth(tu, tv, in.depth, in.vpi);

    int   hcx, hcy;
    float hfx, hfy, htu, htv;
    htu = tu + mvx;
    htv = tv + mvy;
    if (REFL)
    {
        const float curvature = h2f_hi(c3.x);
        float rx = (float)in.x + mvx * fw, ry = (float)in.y + mvy * fh;
        if (in.ray_length > 0.0f && curvature == 0.0f)
        {
            // virtual_point_reprojection: NB current_coord / size without the half-pixel offset
            float vu = __fdiv_rn((float)in.x, fw), vv = __fdiv_rn((float)in.y, fh);
            f3    ro  = world_pos_from_depth(vu, vv, in.depth, in.vpi);
            f3    cr  = sub3(ro, in.cam_pos);
            float crl = len3(cr);
            cr        = normalize3(cr);
            f3 hp     = add3(in.cam_pos, scale3(cr, crl + in.ray_length));
            f4 rp     = mul_m4(in.prev_vp, hp.x, hp.y, hp.z, 1.0f);
            float px = __fdiv_rn(rp.x, rp.w), py = __fdiv_rn(rp.y, rp.w);
            rx = (px * 0.5f + 0.5f) * fw;
            ry = (py * 0.5f + 0.5f) * fh;
        }
        hcx = (int)rx; hcy = (int)ry;
        hfx = rx; hfy = ry;
    }
    else
    {
        hfx = (float)in.x + mvx * fw;
        hfy = (float)in.y + mvy * fh;
        hcx = (int)(hfx + 0.5f);
        hcy = (int)(hfy + 0.5f);
    }
    constexpr int NC = SINGLE ? 1 : 3;
#pragma unroll
    for (int c = 0; c < NC; c++) hcol[c] = 0.0f;
    if (MOMENTS) { hmom[0] = 0.0f; hmom[1] = 0.0f; }

    const int bx = (int)hfx, by = (int)hfy;
    in.apron_miss = (by >= 0 && by < h && (by < in.pgb2.y0 || by >= in.pgb2.y1)) || (by + 1 >= 0 && by + 1 < h && (by + 1 < in.pgb2.y0 || by + 1 >= in.pgb2.y1));
    // Issue EVERY load of the 2x2 bilinear footprint (previous G-buffer, history colour, history moments) and the
    // history-length texel before the first use: one memory round trip instead of three dependent ones
    // (validity -> history -> length).  Texels of invalid taps are fetched but never used.
    uint2 t2[4], t3[4], tm[4];
    float td[4], tcol[4][NC];
#pragma unroll
    for (int s = 0; s < 4; s++)
    {
        const int lx = bx + (s & 1), ly = by + (s >> 1);
        t2[s] = in.pgb2.raw(lx, ly);
        t3[s] = in.pgb3.raw(lx, ly);
        td[s] = in.pdepth.fetch(lx, ly);
        if constexpr (SINGLE)
        {
            if constexpr (sizeof(hist.p[0]) == 4) tcol[s][0] = h2f_lo(hist.raw(lx, ly));
            else tcol[s][0] = hist.fetch(lx, ly);
        }
        else
        {
            const uint2 t = hist.raw(lx, ly);
            tcol[s][0] = h2f_lo(t.x); tcol[s][1] = h2f_hi(t.x); tcol[s][2] = h2f_lo(t.y);
        }
        if (MOMENTS) tm[s] = hist_moments.raw(lx, ly);
    }
    float len_prefetch;
    if (MOMENTS) len_prefetch = h2f_lo(hist_moments.raw(hcx, hcy).y);
    else len_prefetch = hist_length.fetch(hcx, hcy);

    bool v[4];
    bool valid = false;
#pragma unroll
    for (int s = 0; s < 4; s++)
    {
        f3 hn = oct_decode(h2f_lo(t2[s].x), h2f_hi(t2[s].x));
        f3 hp = world_pos_from_depth(htu, htv, td[s], in.vpi);
        v[s]  = reprojection_valid(hcx, hcy, cur_pos, hp, cur_n, hn, cur_id, h2f_lo(t3[s].y), w, h);
        valid = valid || v[s];
    }
    auto fetch_hist = [&](int px, int py, float* col) {
        if constexpr (SINGLE)
        {
            if constexpr (sizeof(hist.p[0]) == 4) col[0] = h2f_lo(hist.raw(px, py));
            else col[0] = hist.fetch(px, py);
        }
        else
        {
            uint2 t = hist.raw(px, py);
            col[0] = h2f_lo(t.x); col[1] = h2f_hi(t.x); col[2] = h2f_lo(t.y);
        }
    };
    if (valid)
    {
        float sumw = 0.0f;
        float fx = fract1(hfx), fy = fract1(hfy);
        float wgt[4] = { (1.0f - fx) * (1.0f - fy), fx * (1.0f - fy), (1.0f - fx) * fy, fx * fy };
#pragma unroll
        for (int s = 0; s < 4; s++)
        {
            if (v[s])
            {
#pragma unroll
                for (int c = 0; c < NC; c++) hcol[c] += wgt[s] * tcol[s][c];
                if (MOMENTS)
                {
                    hmom[0] += wgt[s] * h2f_lo(tm[s].x);
                    hmom[1] += wgt[s] * h2f_hi(tm[s].x);
                }
                sumw += wgt[s];
            }
        }
        valid = (sumw >= 0.01f);
#pragma unroll
        for (int c = 0; c < NC; c++) hcol[c] = valid ? __fdiv_rn(hcol[c], sumw) : 0.0f;
        if (MOMENTS)
        {
            hmom[0] = valid ? __fdiv_rn(hmom[0], sumw) : 0.0f;
            hmom[1] = valid ? __fdiv_rn(hmom[1], sumw) : 0.0f;
        }
    }
    if (!valid)
    {
        float cnt = 0.0f;
        for (int yy = -1; yy <= 1; yy++)
            for (int xx = -1; xx <= 1; xx++)
            {
                int   px = hcx + xx, py = hcy + yy;
                uint2 s2 = in.pgb2.raw(px, py), s3 = in.pgb3.raw(px, py);
                float sd = in.pdepth.fetch(px, py);
                f3    hn = oct_decode(h2f_lo(s2.x), h2f_hi(s2.x));
                f3    hp = world_pos_from_depth(htu, htv, sd, in.vpi);
                if (reprojection_valid(hcx, hcy, cur_pos, hp, cur_n, hn, cur_id, h2f_lo(s3.y), w, h))
                {
                    float col[3];
                    fetch_hist(px, py, col);
#pragma unroll
                    for (int c = 0; c < NC; c++) hcol[c] += col[c];
                    if (MOMENTS)
                    {
                        uint2 m = hist_moments.raw(px, py);
                        hmom[0] += h2f_lo(m.x);
                        hmom[1] += h2f_hi(m.x);
                    }
                    cnt += 1.0f;
                }
            }
        if (cnt > 0.0f)
        {
            valid = true;
#pragma unroll
            for (int c = 0; c < NC; c++) hcol[c] = __fdiv_rn(hcol[c], cnt);
            if (MOMENTS)
            {
                hmom[0] = __fdiv_rn(hmom[0], cnt);
                hmom[1] = __fdiv_rn(hmom[1], cnt);
            }
        }
    }
    if (valid)
    {
        history_length = len_prefetch;
    }
    else
    {
#pragma unroll
        for (int c = 0; c < NC; c++) hcol[c] = 0.0f;
        if (MOMENTS) { hmom[0] = 0.0f; hmom[1] = 0.0f; }
        history_length = 0.0f;
    }
    return valid;
}

} // namespace hr
