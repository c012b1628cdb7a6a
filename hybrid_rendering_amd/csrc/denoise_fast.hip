// Tolerance-mode ("fast") denoise / resolve kernels: hr_*_params.exact == 0.
//
// Same stages, inputs, outputs and tile-classification rules as the exact kernels in shadows.hip / ao.hip / reflections.hip /
// ddgi.hip / upsample.h (which stay the bit-for-bit parity mode), restructured for the hardware instead of for the order of
// operations of the GLSL:
//   * arithmetic through fast_math.h (v_rcp / v_rsq / v_sqrt / v_exp / v_log, contracted FMAs);
//   * reprojection: the view-projection-inverse products shared by the taps are hoisted (M * (u, v, d, 1) is affine in d), the
//     previous-frame normal is compared un-normalised (cos^2 > 0.1 <=> dot^2 > 0.1 |n|^2), one validity / address computation per
//     tap serves all history images, and only the bytes of a texel that are used are loaded;
//   * 17x17 neighbourhood statistics: packed visibility masks are bit-sliced across the spp planes and popcounted with
//     v_bfe + v_bcnt (accumulating); the reflections' colour statistics are separable sums staged through LDS;
//   * a-trous / bilateral blur: step-specialised, interior tiles skip every bounds test, AO blur stages decoded normals and
//     linear depth of the tile + apron in LDS once instead of re-deriving them per tap;
//   * DDGI probe-grid sample: the octahedral coordinates of the surface normal are computed once, not per probe.
// Validated against the oracle within the stated tolerance by tests/test_gpu_tolerance.py; masks and ray counts do not pass
// through this file.
#include "hr_internal.h"
#include "pass_args.h"
#include "fast_math.h"
#include "mask_window.h"
#include "block_map.h"
#include "tile_order.h"
#include "exact_predicates.h"   // before any contract(fast): the knife-edge fallbacks keep the parity kernels' arithmetic
#include "ddgi_sample_fast.h"

#pragma clang fp contract(fast)

#ifndef HR_REFL_VP_EXACT
#define HR_REFL_VP_EXACT 1   // the reflections' virtual-point history coordinate in the parity arithmetic (0: rsq / rcp, as rounds 2-5)
#endif
#ifndef HR_TAP_REDO
#define HR_TAP_REDO 1      // developer A/B: 0 = no second run of the pixel program (round 4's behaviour)
#endif
#ifndef HR_TAP_BAND
#define HR_TAP_BAND 0.4f   // scale of the guard bands of Reproj::tap_valid (developer A/B; 0: never in doubt, the cold copy stays compiled in).  Measured at 1080p, shadows temporal: 0 -> +0 %, 0.3 -> +2.4 %, 1.0 -> +8 % (how often a wave runs twice); 0.4 keeps >= 3x the error bound of the fast operands
#endif

using namespace hr;

namespace {

// ------------------------------------------------------------------------------------------------------------------------
// reprojection.glsl:115-328, tolerance-mode restatement
struct HistGeom { int w, h, y0, y1; };   // every history / previous-G-buffer image of a pass shares one geometry

struct ReprojOut { float col[3]; float mom[2]; float length; };

// HIST_BPP: bytes per texel of the colour history — 2 (AO, R16F), 4 (shadows, RG16F: .r), 8 (reflections, RGBA16F: .rgb).
// Two phases so that a kernel can put independent work (mask popcounts, LDS passes) between them:
//   issue()    history coordinates + all 21 loads of the 2x2 bilinear footprint and the history-length texel — nothing is waited for;
//   resolve()  validity tests, weighting, normalisation (+ the rare 3x3 fallback with its own loads).
// GEO: where the previous frame's geometry comes from.  false: the caller's previous G-buffer — oct normal from GB2 (.x word of an
// 8-byte texel), mesh id from GB3 (.y word), i.e. two images of which half of every cache line is used.  true: the pass's own
// 8-byte geometry record {oct normal, mesh id | linear z} that its temporal kernel wrote LAST frame (it writes one anyway, for the
// a-trous taps; the two halves of the buffer alternate), and — with MOMENTS — the history length comes out of the 8-byte moments texel of
// the nearest tap, which is always one of the four bilinear taps: 4 images, 16 loads, every line fully used, against 5 images and 21
// loads.  Same values (the record holds copies of the G-buffer words), so the results are bit-identical.  GEO = 1: as described
// (shadows, reflections).  GEO = 2 (AO, whose history value is a single fp16): the record is {oct normal, mesh id | AO history}, i.e. the
// colour history rides in the record too — 2 images + the history length, 9 loads against 17.
template <int HIST_BPP, bool MOMENTS, bool REFL, int GEO = 0>
struct Reproj
{
    static constexpr int NC = HIST_BPP == 8 ? 3 : 1;
    const float* __restrict__ M;
    const void* __restrict__  pgb2;
    const void* __restrict__  pgb3;
    const void* __restrict__  pdepth;
    const void* __restrict__  hist;
    const void* __restrict__  hist_moments;
    const void* __restrict__  hist_len;
    const void* __restrict__  geo;          // GEO: previous frame's {oct normal, mesh id | linear z} records
    HistGeom   g;
    f3         cur_pos, cur_n;
    float      cur_id, hfx, hfy, band;
    uint32_t   near13;       // resolve<NOTE>: bit s (< 4) = bilinear tap s on a knife edge, bit 4 + k = tap k of the 3x3 fallback
    uint32_t   valid13;      // resolve<NOTE>: the verdicts used (same layout); bit 13 = the 3x3 fallback was evaluated
    int        hcx, hcy;
    bool       inb, lok, tok[4], apron_miss;   // apron_miss: the footprint touched an image row that is not resident (row bands)
    fm::Unproj hb;
    uint32_t   g2x[4], g3y[4], mm[4], hx[4], hy[4], lraw;
    uint32_t   mlen[4];
    float      td[4];

    // texel (px, py) of the previous frame passes is_reprojection_valid (reprojection.glsl:52-67) for this pixel; branch-free.
    // NOTE (shadows / AO): `near` = the tap sits inside a guard band around one of the two thresholds, where the fast operands cannot be
    // trusted with the decision (they are good to ~3 ulp of the positions' coordinates; the parity kernels round every operation):
    // |plane distance - 5| < band = HR_TAP_BAND (1e-4 + 3e-6 |p|_1) (cur_pos and the tap's position are each good to ~3 ulp of their
    // coordinates, their difference along the normal to ~4e-7 |p|_1), |cos^2 - 0.1 |n'|^2| <= HR_TAP_BAND 4e-6 |n'|^2 (fast: ~1e-7).  Branch-free, 4 VALU per tap.  The bands decide how often a wave runs its pixels twice
    // (first version, 0.01 + 1.6e-5 |p|_1 and 4e-4: shadows temporal +18 %, AO +15 % at 1080p; docs/EXPERIMENTS.md R5.1).
    template <bool NOTE>
    HR_DEV bool tap_valid(uint32_t q2x, uint32_t q3y, float d, bool& near) const
    {
        const f3    hn = fm::oct_raw(q2x);
        const float dn = fm::dot(cur_n, hn), hh = fm::dot(hn, hn);
        const f3    hp = fm::unproject_at(hb, M, d);
        const float pd = __builtin_fabsf(fm::dot(sub3(cur_pos, hp), cur_n));
        const float t  = dn * dn - 0.1f * hh;
        const int   pre = (int)inb & (int)(cur_id == fm::lo(q3y));
        near = false;
        if constexpr (NOTE) near = (pre & ((int)(__builtin_fabsf(pd - 5.0f) < band) | (int)(__builtin_fabsf(t) <= (HR_TAP_BAND * 4e-6f) * hh))) != 0;
        return (pre & (int)!(pd > 5.0f) & (int)(t > 0.0f)) != 0;
    }
    HR_DEV uint32_t tap_offset(int px, int py, bool& ok) const
    {
        ok = ((int)(px < 0) | (int)(py < g.y0) | (int)(px >= g.w) | (int)(py >= g.y1)) == 0;
        return ok ? (uint32_t)(py * g.w + px) : (uint32_t)(g.y0 * g.w);
    }
    HR_DEV void hist_decode(uint32_t rx, uint32_t ry, float* c) const
    {
        if constexpr (HIST_BPP == 2) c[0] = (float)__builtin_bit_cast(_Float16, (uint16_t)rx);
        else if constexpr (HIST_BPP == 4) c[0] = fm::lo(rx);
        else { c[0] = fm::lo(rx); c[1] = fm::hi(rx); c[2] = fm::lo(ry); }
    }
    HR_DEV void hist_load(uint32_t off, uint32_t& rx, uint32_t& ry) const
    {
        ry = 0u;
        if constexpr (HIST_BPP == 2) rx = fm::ld<uint16_t>(hist, off * 2u);
        else if constexpr (HIST_BPP == 4) rx = fm::ld<uint32_t>(hist, off * 4u);
        else { const uint2 t = fm::ld<uint2>(hist, off * 8u); rx = t.x; ry = t.y; }
    }

    HR_DEV void issue(int x, int y, float depth, uint32_t c2y, float cur_id_, float curvature, f3 cur_n_, f3 cam_pos, const float* __restrict__ prev_vp, float ray_length)
    {
        cur_id = cur_id_; cur_n = cur_n_;
        const float fw = (float)g.w, fh = (float)g.h;
        // (x + 0.5) / w correctly rounded (round 5): the clip -> world products below cancel 3-4 digits, so ONE ulp of tu moved a world
        // position by up to ~1e-3 — 20 ulp of its coordinates — and the plane distance of a history tap with it (measured: the guard band of
        // tap_valid had to be 10x wider than the error of everything else).  With the parity kernels' tu / tv the products are theirs
        // bit for bit (same operation order, fast_math.h) and only the perspective divide differs (rcp_nr: ~1 ulp).  ~10 VALU per pixel.
        const float tu = fm::div_by_inrange((float)x + 0.5f, div_prepare(fw)), tv = fm::div_by_inrange((float)y + 0.5f, div_prepare(fh));
        const float mvx = fm::lo(c2y), mvy = fm::hi(c2y);
        cur_pos = fm::unproject_at(fm::unproject_base(M, tu, tv), M, depth);
        band    = HR_TAP_BAND * 1e-4f + (HR_TAP_BAND * 3e-6f) * (__builtin_fabsf(cur_pos.x) + __builtin_fabsf(cur_pos.y) + __builtin_fabsf(cur_pos.z));   // tap_valid<true>
        hfx = (float)x + mvx * fw; hfy = (float)y + mvy * fh;   // mv (fp16) * extent (<= 2^12) is exact: same texel as the exact mode
        if (REFL)
        {
            if (ray_length > 0.0f && curvature == 0.0f)
            {
#if HR_REFL_VP_EXACT
                // virtual_point_reprojection with the parity kernels' operations (exact_predicates.h: the weights of the history taps — and so the
                // interpolated moments — are then the oracle's; round 5, late)
                exact::virtual_point(M, prev_vp, cam_pos, x, y, g.w, g.h, depth, ray_length, hfx, hfy);
#else
                // virtual_point_reprojection (reprojection.glsl:71-111): NB current_coord / size without the half-pixel offset
                const f3    ro  = fm::unproject_at(fm::unproject_base(M, fm::div_by_inrange((float)x, div_prepare(fw)), fm::div_by_inrange((float)y, div_prepare(fh))), M, depth);
                f3          cr  = sub3(ro, cam_pos);
                const float l2  = fm::dot(cr, cr);
                const float il  = fm::rsq(l2), crl = l2 * il;
                const float k   = (crl + ray_length) * il;
                const f3    hp  = mk3(cam_pos.x + cr.x * k, cam_pos.y + cr.y * k, cam_pos.z + cr.z * k);
                const float pw  = prev_vp[3] * hp.x + prev_vp[7] * hp.y + prev_vp[11] * hp.z + prev_vp[15];
                const float ipw = fm::rcp(pw);
                const float px  = (prev_vp[0] * hp.x + prev_vp[4] * hp.y + prev_vp[8] * hp.z + prev_vp[12]) * ipw;
                const float py  = (prev_vp[1] * hp.x + prev_vp[5] * hp.y + prev_vp[9] * hp.z + prev_vp[13]) * ipw;
                hfx = (px * 0.5f + 0.5f) * fw;
                hfy = (py * 0.5f + 0.5f) * fh;
#endif
            }
            hcx = (int)hfx; hcy = (int)hfy;
        }
        else
        {
            hcx = (int)(hfx + 0.5f);
            hcy = (int)(hfy + 0.5f);
        }
        inb = ((int)(hcx < 0) | (int)(hcy < 0) | (int)(hcx > g.w - 1) | (int)(hcy > g.h - 1)) == 0;
        hb  = fm::unproject_base(M, tu + mvx, tv + mvy);
        // an outside tap loads a resident address (its value is discarded in resolve()), so no load sits behind a branch
        const int bx = (int)hfx, by = (int)hfy;
        apron_miss = ((unsigned)by < (unsigned)g.h && (by < g.y0 || by >= g.y1)) || ((unsigned)(by + 1) < (unsigned)g.h && (by + 1 < g.y0 || by + 1 >= g.y1));
        uint32_t  off[4];
#pragma unroll
        for (int s = 0; s < 4; s++) off[s] = tap_offset(bx + (s & 1), by + (s >> 1), tok[s]);
        const uint32_t loff = tap_offset(hcx, hcy, lok);
#pragma unroll
        for (int s = 0; s < 4; s++)
        {
            if constexpr (GEO != 0)
            {
                const uint2 q = fm::ld<uint2>(geo, off[s] * 8u);
                g2x[s] = q.x; g3y[s] = q.y;
            }
            else
            {
                g2x[s] = fm::ld<uint32_t>(pgb2, off[s] * 8u);
                g3y[s] = fm::ld<uint32_t>(pgb3, off[s] * 8u + 4u);
            }
            td[s]  = fm::ld<float>(pdepth, off[s] * 4u);
            if constexpr (GEO == 2) { hx[s] = g3y[s] >> 16; hy[s] = 0u; }
            else hist_load(off[s], hx[s], hy[s]);
            if constexpr (MOMENTS && GEO != 0)
            {
                const uint2 m = fm::ld<uint2>(hist_moments, off[s] * 8u);
                mm[s] = m.x; mlen[s] = m.y;
            }
            else mm[s] = MOMENTS ? fm::ld<uint32_t>(hist_moments, off[s] * 8u) : 0u;
        }
        if constexpr (MOMENTS && GEO != 0)
        {
            // the nearest history texel (hcx, hcy) is one of the four taps: bx <= hcx <= bx + 1 (truncation on both sides)
            const int sel = (hcx - bx) + 2 * (hcy - by);
            lraw = sel == 0 ? mlen[0] : (sel == 1 ? mlen[1] : (sel == 2 ? mlen[2] : mlen[3]));
        }
        else if (MOMENTS) lraw = fm::ld<uint32_t>(hist_moments, loff * 8u + 4u);
        else lraw = fm::ld<uint16_t>(hist_len, loff * 2u);
    }

    // NOTE = true (shadows, AO): taps on a knife edge are recorded in near13.  `known` (bit layout of near13): taps whose validity is
    // taken from `ovr` — the parity kernels' verdicts of exact_bits() — instead of the fast test (and which are then not noted again).
    template <bool NOTE = false>
    HR_DEV bool resolve(ReprojOut& o, uint32_t ovr = 0u, uint32_t known = 0u)
    {
        const float fx = hfx - __builtin_floorf(hfx), fy = hfy - __builtin_floorf(hfy);
        const float wgt[4] = { (1.0f - fx) * (1.0f - fy), fx * (1.0f - fy), (1.0f - fx) * fy, fx * fy };
        float sumw = 0.0f, col[NC], mom0 = 0.0f, mom1 = 0.0f;
        uint32_t noted = 0u, used = 0u;
#pragma unroll
        for (int c = 0; c < NC; c++) col[c] = 0.0f;
#pragma unroll
        for (int s = 0; s < 4; s++)
        {
            // an out-of-image texel reads as zeros (pinned rule) and is validated as such, as in the exact mode: its weight counts
            // in the normalisation, its (zero) values add nothing
            bool nr;
            bool v = tap_valid<NOTE>(tok[s] ? g2x[s] : 0u, tok[s] ? g3y[s] : 0u, tok[s] ? td[s] : 0.0f, nr);
            if (NOTE) { if ((known >> s) & 1u) v = (ovr >> s) & 1u; else noted |= (uint32_t)nr << s; used |= (uint32_t)v << s; }
            const float ws = v ? wgt[s] : 0.0f, wv = ((int)v & (int)tok[s]) ? wgt[s] : 0.0f;
            float c3[NC];
            hist_decode(hx[s], hy[s], c3);
#pragma unroll
            for (int c = 0; c < NC; c++) col[c] += wv * c3[c];
            // the moments keep the parity kernels' two roundings per term (no FMA): the variance downstream is m2 - m1^2, and where it is tiny
            // (<= 1e-8: a sixth of the surface texels of a young history) ONE fp32 ulp of a moment is a multiple of it — and the a-trous
            // chain divides luminance differences by its square root (docs/EXPERIMENTS.md R5.8)
            if (MOMENTS) { mom0 = fm::mad_rn(wv, fm::lo(mm[s]), mom0); mom1 = fm::mad_rn(wv, fm::hi(mm[s]), mom1); }
            sumw += ws;
        }
        // Normalisations are correctly rounded divisions (one shared denominator: fast_math.h div_by_inrange): a history of
        // all-equal values must come back EXACTLY (sum(w) v / sum(w) == v) — the tile classification tests `ao < 1`, `visibility > 0`
        bool valid = sumw >= 0.01f;
        if (valid)
        {
            const DivBy D = div_prepare(sumw);   // 0.01 <= sumw <= 1; numerators are fp16 values times weights <= 1
#pragma unroll
            for (int c = 0; c < NC; c++) col[c] = fm::div_by_inrange(col[c], D);
            if (MOMENTS) { mom0 = fm::div_by_inrange(mom0, D); mom1 = fm::div_by_inrange(mom1, D); }
        }
        else
        {
            // 3x3 fallback around the nearest history texel (:266-303); rare (disocclusion borders): kept rolled
            float cnt = 0.0f;
#pragma unroll
            for (int c = 0; c < NC; c++) col[c] = 0.0f;
            mom0 = mom1 = 0.0f;
#pragma unroll 1
            for (int k = 0; k < 9; k++)
            {
                bool           ok;
                const uint32_t qo = tap_offset(hcx + k % 3 - 1, hcy + k / 3 - 1, ok);
                uint32_t q2, q3;
                if constexpr (GEO != 0) { const uint2 q = fm::ld<uint2>(geo, qo * 8u); q2 = q.x; q3 = q.y; }
                else { q2 = fm::ld<uint32_t>(pgb2, qo * 8u); q3 = fm::ld<uint32_t>(pgb3, qo * 8u + 4u); }
                const float    qd = fm::ld<float>(pdepth, qo * 4u);
                uint32_t       qx, qy;
                if constexpr (GEO == 2) { qx = q3 >> 16; qy = 0u; }
                else hist_load(qo, qx, qy);
                const uint32_t qm = MOMENTS ? fm::ld<uint32_t>(hist_moments, qo * 8u) : 0u;
                bool nr;
                bool tv = tap_valid<NOTE>(ok ? q2 : 0u, ok ? q3 : 0u, ok ? qd : 0.0f, nr);
                if (NOTE) { if ((known >> (4 + k)) & 1u) tv = (ovr >> (4 + k)) & 1u; else noted |= (uint32_t)nr << (4 + k); used |= ((uint32_t)tv << (4 + k)) | (1u << 13); }
                if (tv)
                {
                    float c3[NC];
                    hist_decode(ok ? qx : 0u, ok ? qy : 0u, c3);
#pragma unroll
                    for (int c = 0; c < NC; c++) col[c] += c3[c];
                    if (MOMENTS) { mom0 += fm::lo(ok ? qm : 0u); mom1 += fm::hi(ok ? qm : 0u); }
                    cnt += 1.0f;
                }
            }
            if (cnt > 0.0f)
            {
                valid = true;
                const DivBy D = div_prepare(cnt);
#pragma unroll
                for (int c = 0; c < NC; c++) col[c] = fm::div_by_inrange(col[c], D);
                if (MOMENTS) { mom0 = fm::div_by_inrange(mom0, D); mom1 = fm::div_by_inrange(mom1, D); }
            }
        }
#pragma unroll
        for (int c = 0; c < NC; c++) o.col[c] = valid ? col[c] : 0.0f;
        o.mom[0] = valid ? mom0 : 0.0f;
        o.mom[1] = valid ? mom1 : 0.0f;
        o.length = ((int)valid & (int)lok) ? fm::lo(lraw) : 0.0f;
        near13 = noted; valid13 = used;
        return valid;
    }

    // The parity kernels' verdicts on the taps `which` of this pixel (bit layout of near13): their geometry is fetched again here (nothing of
    // the hot path stays live across this), every verdict is exact::tap_valid.  Runs for the few waves that noted a tap on a knife edge.
    // (x, y), depth, c2x, c2y: the pixel, its depth and its GB2 words as given to issue().
    HR_DEV uint32_t exact_bits(uint32_t which, int x, int y, float depth, uint32_t c2x, uint32_t c2y) const
    {
        const int bx = (int)hfx, by = (int)hfy;
        uint32_t  bits = 0u;
#pragma unroll 1
        while (which)
        {
            const int k = __builtin_ctz(which);
            which &= which - 1u;
            const int px = k < 4 ? bx + (k & 1) : hcx + (k - 4) % 3 - 1, py = k < 4 ? by + (k >> 1) : hcy + (k - 4) / 3 - 1;
            bool           ok;
            const uint32_t qo = tap_offset(px, py, ok);
            uint32_t q2, q3;
            if constexpr (GEO != 0) { const uint2 q = fm::ld<uint2>(geo, qo * 8u); q2 = q.x; q3 = q.y; }
            else { q2 = fm::ld<uint32_t>(pgb2, qo * 8u); q3 = fm::ld<uint32_t>(pgb3, qo * 8u + 4u); }
            const float qd = fm::ld<float>(pdepth, qo * 4u);
            bits |= (uint32_t)exact::tap_valid(M, x, y, g.w, g.h, depth, c2x, c2y, cur_id, hcx, hcy, ok ? q2 : 0u, ok ? q3 : 0u, ok ? qd : 0.0f) << k;
        }
        return bits;
    }
};

// at the top of a temporal kernel whose argument block carries a GeoApronArgs: true = this workgroup was one of the apron's (pass_args.h)
template <int THREADS>
HR_DEV bool geo_apron_rides(const GeoApronArgs& g)
{
    if (g.row0 < 0 || (int)blockIdx.y < g.row0) return false;
    const int      na = g.a1 - g.a0, total = (na + (g.b1 - g.b0)) * g.w;
    const int      stride = (int)(gridDim.y - (unsigned)g.row0) * (int)gridDim.x * THREADS;
    for (int i = (((int)blockIdx.y - g.row0) * (int)gridDim.x + (int)blockIdx.x) * THREADS + (int)threadIdx.x; i < total; i += stride)
    {
        const int      r = i / g.w, x = i - r * g.w, y = r < na ? g.a0 + r : g.b0 + (r - na);
        const uint32_t o = (uint32_t)(y * g.w + x);
        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(g.out) + o * 8u) = make_uint2(fm::ld<uint32_t>(g.gb2, o * 8u), fm::ld<uint32_t>(g.gb3, o * 8u + 4u));
    }
    return true;
}

// ------------------------------------------------------------------------------------------------------------------------
// shadows_denoise_reprojection.comp:196-293 (+ reset_args / tile classification), tolerance mode
#define FT_WAVES 4
#ifndef FT_AO_BLUR_ROWS
#define FT_AO_BLUR_ROWS 1      // block_map.h block_xy<R>: tile rows per XCD run for the fused AO blur (0: identity, then HR_COLS8 bit 8 applies)
#endif
#ifndef FT_RATROUS01_ROWS
#define FT_RATROUS01_ROWS 1    // ... and for the fused reflections a-trous 0 + 1 (0: identity, HR_COLS8 bit 9)
#endif
#ifndef FT_SHADOWS_EU
#define FT_SHADOWS_EU 6   // minimum waves per SIMD the register allocator must leave room for (round 5: 6 — with the cold copy of the pixel program 5 lets the allocator take 88 VGPRs; 6 = 80 VGPRs, three spill stores on the hot path).  Round 4, without the cold copy: 5 and 6 give the same 77 VGPRs (= 6 waves per SIMD), no
                          // spill: 48-50 us at 1080p, 208 at 4K; 7 (72 VGPRs + 24 B of scratch) 58.7 / 252.8; 8 (64 VGPRs, more scratch) 81.4 / 362 — spills
                          // cost far more than waves buy
#endif
template <int GEO>
__global__ __launch_bounds__(64 * FT_WAVES, FT_SHADOWS_EU) void kf_shadows_temporal(TemporalArgs a)
{
    if (geo_apron_rides<64 * FT_WAVES>(a.apron)) return;   // row bands: the records of the history apron, as extra grid rows behind the sort's
    if (tile_order_rides<64 * FT_WAVES>(a.sort)) return;   // the trace kernel's next launch order, as extra grid rows (tile_order.h)
    const uint2 BLK = block_xy<0>();
    __shared__ uint32_t s_mask[FT_WAVES][1][18];
    __shared__ MaskRows s_rows[FT_WAVES];
    const int  lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // grid = (tile columns / FT_WAVES, tile rows): no integer division per wave
    const int  txr = BLK.x * FT_WAVES + wave;
    const bool tile_ok = txr < a.tiles_x;
    const int  tx = tile_ok ? txr : 0, ty = (int)BLK.y + a.tile_y0;
    const int  lx = lane & 7, ly = lane >> 3;
    const int  x = tx * 8 + lx, y = ty * 8 + ly;
    // order of the memory round trips: centre texels -> (mask words, LDS) -> history taps -> (popcounts) -> resolve
    const bool in_image = tile_ok && x < a.w && y < a.h && y >= a.y0 && y < a.y1;
    const bool edge     = tile_ok && (x >= a.w || y >= a.h);   // thread without a pixel: votes as the unguarded shader thread does
    const uint32_t pix  = in_image ? (uint32_t)(y * a.w + x) : (uint32_t)(a.y0 * a.w);
    const float d_raw   = fm::ld<float>(a.depth.p, pix * 4u);
    const uint2 cg2_raw = fm::ld<uint2>(a.gb2.p, pix * 8u), cg3_raw = fm::ld<uint2>(a.gb3.p, pix * 8u);
    build_mask_rows<false>(s_rows[wave], s_mask[wave], a.mask, 1, a.mw, a.mh, tx, ty, a.y0, a.y1, lane, tile_ok);
    if (!tile_ok) return;
    // The pixel's program, from its centre texels on.  It runs once; a wave in which some history tap sat on a knife edge of the validity
    // test (Reproj::tap_valid notes it: a handful of pixels per frame) runs it a SECOND time — a second, cold copy of the code, entered with
    // nothing of the first run live — in which the noted pixels take the parity kernels' verdicts on their taps (Reproj::exact_bits), and
    // stores again.  Returns near13 (0: no tap of this pixel was in doubt); flag = the pixel's vote in the tile classification.
    bool flag = false;
    auto pixel = [&](const float d_in, const uint2 cg2_in, const uint2 cg3_in, const bool redo) -> uint32_t {
        const float d   = edge ? 0.0f : d_in;
        const uint2 cg2 = edge ? make_uint2(0u, 0u) : cg2_in, cg3 = edge ? make_uint2(0u, 0u) : cg3_in;
        const f3    cn  = fm::oct_unit(cg2.x);
        const bool  live = (in_image || edge) && d != 1.0f;
        Reproj<4, true, false, GEO> rp;
        rp.M = a.vpi; rp.pgb2 = a.pgb2.p; rp.pgb3 = a.pgb3.p; rp.pdepth = a.pdepth.p; rp.hist = a.hist.p; rp.hist_moments = a.hist_moments.p; rp.hist_len = nullptr;
        rp.geo = a.geo_hist;
        rp.g = HistGeom { a.w, a.h, a.pgb2.y0, a.pgb2.y1 };
        const bool reproj = live && !a.debug_skip_reproject;
        uint32_t ovr = 0u, known = 0u;
        if (redo && reproj)
        {
            // Which taps were in doubt is found again the way the first run found it (the fast test is deterministic); THOSE taps get the parity
            // kernels' verdicts (typically one tap: ~250 VALU instead of 3250 for all 13).  If that changes the verdict of a bilinear tap, the
            // 3x3 fallback may run where it did not before (or the other way round) and its taps were never tested for doubt: then all nine
            // get exact verdicts as well.  Straight-line code on purpose: as a loop ("until nothing new is noted") the allocator spilled on
            // the HOT path (+13 %, docs/EXPERIMENTS.md R5.1).
            rp.issue(x, y, d, cg2.y, fm::lo(cg3.y), 0.0f, cn, mk3(0, 0, 0), nullptr, 0.0f);
            ReprojOut r0;
            rp.template resolve<true>(r0);
            if (rp.near13)
            {
                known = rp.near13;
                ovr   = rp.exact_bits(known, x, y, d, cg2.x, cg2.y);
                if ((((ovr ^ rp.valid13) & known) & 0xfu) != 0u)   // a bilinear verdict flipped
                {
                    const uint32_t rest = 0x1ff0u & ~known;
                    ovr |= rp.exact_bits(rest, x, y, d, cg2.x, cg2.y);
                    known |= rest;
                }
            }
        }
        if (reproj) rp.issue(x, y, d, cg2.y, fm::lo(cg3.y), 0.0f, cn, mk3(0, 0, 0), nullptr, 0.0f);
        if (!redo && a.apron_flag && __ballot(reproj && in_image && y >= a.band_y0 && y < a.band_y1 && rp.apron_miss) && lane == 0) atomicOr(a.apron_flag, 1u);
        int sum, own;
        mask_window<false>(s_rows[wave], lx, ly, sum, own);   // 17 bfe + bcnt pairs while the 21 history loads are in flight
        const float mean = fm::div_by_inrange((float)sum, div_prepare(289.0f));

        float out_v = 0.0f, out_var = 0.0f, m0 = 0.0f, m1 = 0.0f, hlen = 0.0f;
        flag = false;
        rp.near13 = 0u;
        if (live)
        {
            const float visibility = (float)own;
            ReprojOut   r;
            bool        success = false;
            if (reproj) success = rp.template resolve<true>(r, ovr, known);
            else { r.col[0] = 0.0f; r.mom[0] = r.mom[1] = 0.0f; r.length = 0.0f; }
            hlen = fm::fmin_(32.0f, success ? r.length + 1.0f : 1.0f);
            float hv = r.col[0];
            if (success)
            {
                const float sd = fm::sqrt1(fm::fmax_(fm::var_rn(mean, mean), 0.0f));
                hv = fm::fmin_(fm::fmax_(hv, mean - 0.5f * sd), mean + 0.5f * sd);
            }
            const float ih = fm::rcp_nr(hlen);
            const float al = success ? fm::fmax_(a.alpha, ih) : 1.0f;
            const float am = success ? fm::fmax_(a.moments_alpha, ih) : 1.0f;
            m0      = fm::mix_rn(r.mom[0], visibility, am);
            m1      = fm::mix_rn(r.mom[1], visibility * visibility, am);
            out_var = fm::fmax_(0.0f, fm::var_rn(m1, m0));
            out_v   = fm::mix_rn(hv, visibility, al);
            flag    = out_v > 0.0f;
        }
        if (in_image)
        {
            *reinterpret_cast<uint2*>(reinterpret_cast<char*>(a.out_moments) + pix * 8u) = make_uint2(fm::pack2(m0, m1), fm::pack2(hlen, 0.0f));
            // for the a-trous iterations: the centre's octahedral normal and linear depth, 8 bytes (copies of the G-buffer's fp16 values —
            // the exact mode keeps 16 bytes of decoded fp32; decoding per tap is cheaper than the extra 8 B x 5 passes of HBM traffic)
            *reinterpret_cast<uint2*>(reinterpret_cast<char*>(a.nd) + pix * 8u)         = make_uint2(cg2.x, cg3.y);
            *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(a.out) + pix * 4u)     = fm::pack2(out_v, out_var);
        }
        return reproj ? rp.near13 : 0u;
    };
    const uint32_t doubt = pixel(d_raw, cg2_raw, cg3_raw, false);
#if HR_TAP_REDO
    if (__any(doubt != 0u))   // rare: the cold copy (re-reads the centre texels: nothing of the first run is kept)
        pixel(fm::ld<float>(a.depth.p, pix * 4u), fm::ld<uint2>(a.gb2.p, pix * 8u), fm::ld<uint2>(a.gb3.p, pix * 8u), true);
#endif
    const unsigned long long any = __ballot(flag);
    if (lane == 0) a.tile_class[(size_t)ty * a.tiles_x + tx] = any ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// shadows_denoise_atrous.comp:94-174 + copy_shadow_tiles (tile class), tolerance mode.  STEP > 0: compile-time tap distance and
// radius 1 (the reference default, fully unrolled); STEP == 0: run-time step and radius.
struct EdgeK { float kz, phi_n, inv_phi_l; bool n32; };
HR_DEV float edge_weight_fast(const EdgeK& k, float cd, float sd, f3 cn, f3 sn, float cl, float sl)
{
    // edge_stopping.glsl:31-62 (NORMAL + LUMA): exp(-max(wL, 0) - max(wZ, 0)) * wN with wZ = exp(-|dz| / sigma) — sic, the depth
    // WEIGHT is used as an exponent term upstream; kept
    const float wZ = fm::exp2f_(-__builtin_fabsf(cd - sd) * k.kz);
    const float dn = fm::sat(fm::dot(cn, sn));
    const float wN = k.n32 ? fm::pow32(dn) : fm::powf_(dn, k.phi_n);
    const float wL = __builtin_fabsf(cl - sl) * k.inv_phi_l;
    return fm::exp2f_((wL + wZ) * -1.44269504088896341f) * wN;
}

template <int STEP, bool N32>
__global__ __launch_bounds__(256) void kf_shadows_atrous(AtrousArgs a)
{
    const uint2 BLK = block_xy<0>();
    const int x = BLK.x * 32 + (threadIdx.x & 31);
    const int y = a.y0 + BLK.y * 8 + (threadIdx.x >> 5);
    if (x >= a.w || y >= a.y1) return;
    const uint32_t o = (uint32_t)(y * a.w + x);
    uint32_t* outp  = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(a.out) + o * 4u);
    uint32_t* out2p = a.out2 ? reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(a.out2) + o * 4u) : nullptr;
    if (!a.tile_class[(size_t)(y >> 3) * a.tiles_x + (x >> 3)])
    {
        *outp = 0u; // shadows_denoise_copy_shadow_tiles.comp:35
        if (out2p) *out2p = 0u;
        return;
    }
    const int step = STEP > 0 ? STEP : a.step;
    const int R    = STEP > 0 ? 1 : a.radius;
    // a workgroup whose whole footprint lies inside the resident rows needs no bounds test at all (uniform branch)
    const int  fx0 = (int)BLK.x * 32, fy0 = a.y0 + (int)BLK.y * 8;
    const int  reach = step * R > 1 ? step * R : 1;
    const bool interior = fx0 - reach >= 0 && fx0 + 31 + reach < a.w && fy0 - reach >= (a.y0 > 0 ? a.y0 : 0) && fy0 + 7 + reach < (a.y1 < a.h ? a.y1 : a.h);
    const uint32_t c   = fm::ld<uint32_t>(a.in.p, o * 4u);
    const uint2    cnd = fm::ld<uint2>(a.nd, o * 8u);   // oct normal (fp16 x 2), mesh id | linear z (fp16 x 2)
    const float    cz  = fm::hi(cnd.y);
    float var = 0.0f;
    // compute_variance_center (:65-88): 3x3 gaussian of the variance channel, unit taps
    if (interior)
    {
#pragma unroll
        for (int k = 0; k < 9; k++)
        {
            const int   xx = k % 3 - 1, yy = k / 3 - 1;
            const float kw = (xx == 0 ? (yy == 0 ? 0.25f : 0.125f) : (yy == 0 ? 0.125f : 0.0625f));
            var += fm::hi(fm::ld<uint32_t>(a.in.p, (uint32_t)((y + yy) * a.w + (x + xx)) * 4u)) * kw;
        }
    }
    else
    {
#pragma unroll
        for (int k = 0; k < 9; k++)
        {
            const int   xx = k % 3 - 1, yy = k / 3 - 1;
            const float kw = (xx == 0 ? (yy == 0 ? 0.25f : 0.125f) : (yy == 0 ? 0.125f : 0.0625f));
            var += fm::hi(a.in.raw(x + xx, y + yy)) * kw;
        }
    }
    uint32_t result = c;
    if (!(cz < 0.0f))
    {
        const f3    cn = fm::oct_unit(cnd.x);
        const float cv = fm::lo(c);
        EdgeK ek;
        ek.kz        = 1.44269504088896341f * fm::rcp(a.sigma_depth);
        ek.phi_n     = a.phi_normal;
        ek.n32       = N32;
        ek.inv_phi_l = fm::rcp(a.phi_visibility * fm::sqrt1(fm::fmax_(0.0f, 1e-10f + var)));
        float sum_w = 1.0f, sum_v = cv, sum_var = fm::hi(c);
        if (STEP > 0)
        {
            uint32_t t_in[8];
            uint2    t_nd[8];
            bool     t_ok[8];
            if (interior)
            {
#pragma unroll
                for (int t = 0; t < 8; t++)
                {
                    const int      k = t < 4 ? t : t + 1, xx = k % 3 - 1, yy = k / 3 - 1;
                    const uint32_t so = (uint32_t)((y + yy * STEP) * a.w + (x + xx * STEP));
                    t_ok[t] = true;
                    t_in[t] = fm::ld<uint32_t>(a.in.p, so * 4u);
                    t_nd[t] = fm::ld<uint2>(a.nd, so * 8u);
                }
            }
            else
            {
#pragma unroll
                for (int t = 0; t < 8; t++)
                {
                    const int k = t < 4 ? t : t + 1, xx = k % 3 - 1, yy = k / 3 - 1;
                    const int px = x + xx * STEP, py = y + yy * STEP;
                    // a tap outside the image is skipped; one outside the resident rows of a band reads zeros, whose weight is 0 (halo
                    // rows only): both get weight 0 below
                    const bool     res = px >= 0 && py >= 0 && px < a.w && py < a.h && py >= a.y0 && py < a.y1;
                    const uint32_t so  = res ? (uint32_t)(py * a.w + px) : o;
                    t_ok[t] = res;
                    t_in[t] = fm::ld<uint32_t>(a.in.p, so * 4u);
                    t_nd[t] = fm::ld<uint2>(a.nd, so * 8u);
                }
            }
#pragma unroll
            for (int t = 0; t < 8; t++)
            {
                const int   k = t < 4 ? t : t + 1, xx = k % 3 - 1, yy = k / 3 - 1;
                const float kk = (xx == 0 ? 1.0f : 2.0f / 3.0f) * (yy == 0 ? 1.0f : 2.0f / 3.0f);
                const float sv = fm::lo(t_in[t]);
                float wv = fm::mul_rn(edge_weight_fast(ek, cz, fm::hi(t_nd[t].y), cn, fm::oct_unit(t_nd[t].x), cv, sv), kk);
                if (!t_ok[t]) wv = 0.0f;
                sum_w += wv;
                sum_v += wv * sv;
                sum_var += (wv * wv) * fm::hi(t_in[t]);
            }
        }
        else
        {
            for (int yy = -R; yy <= R; yy++)
                for (int xx = -R; xx <= R; xx++)
                {
                    const int px = x + xx * step, py = y + yy * step;
                    if (px < 0 || py < 0 || px >= a.w || py >= a.h || (xx == 0 && yy == 0)) continue;
                    const int   axx = xx < 0 ? -xx : xx, ayy = yy < 0 ? -yy : yy;
                    const float kx = axx == 0 ? 1.0f : (axx == 1 ? 2.0f / 3.0f : 1.0f / 6.0f);
                    const float ky = ayy == 0 ? 1.0f : (ayy == 1 ? 2.0f / 3.0f : 1.0f / 6.0f);
                    if (py < a.y0 || py >= a.y1) continue;   // outside the resident rows of a band: zeros, weight 0
                    const uint32_t so = (uint32_t)(py * a.w + px);
                    const uint32_t s  = fm::ld<uint32_t>(a.in.p, so * 4u);
                    const uint2    nd = fm::ld<uint2>(a.nd, so * 8u);
                    const float sv = fm::lo(s);
                    const float wv = fm::mul_rn(edge_weight_fast(ek, cz, fm::hi(nd.y), cn, fm::oct_unit(nd.x), cv, sv), kx * ky);
                    sum_w += wv;
                    sum_v += wv * sv;
                    sum_var += (wv * wv) * fm::hi(s);
                }
        }
        const float iw = fm::rcp(sum_w);
        float ov = sum_v * iw;
        const float ovar = sum_var * (iw * iw);
        if (a.power != 0.0f) ov = fm::powf_(fm::fmax_(ov, 0.0f), a.power);
        result = fm::pack2(ov, ovar);
    }
    *outp = result;
    if (out2p) *out2p = result;
}


// The same filter with its taps staged through LDS (radius 1, tap distance STEP = 1 / 2 / 4 / 8: the reference's four iterations).
// The 32x8 tile and its STEP-wide apron are fetched ONCE per workgroup — (32+2S)x(8+2S) texels instead of 27 loads per pixel, which
// had the kernel bound by the L1 data path, not by HBM or the VALU (SQ counters: 38 % VALU busy, counter traffic 27-40 MB) — and
// each texel's octahedral normal is decoded once instead of once per tap that reads it.  A texel that is not resident (outside the
// image or the band's rows) is staged as (value 0, normal 0): its edge weight is exactly 0 (pow(dot(n, 0), phi) = 0).
#ifndef FT_ATROUS_LDS
#define FT_ATROUS_LDS 1
#endif
#ifndef FT_ATROUS_VOTE
#define FT_ATROUS_VOTE 1   // 0: stage unconditionally — 14.4 -> 17.8 µs (step 1): the shadow-tile workgroups are a large share of the launch
#endif
template <int STEP, bool N32>
__global__ __launch_bounds__(256) void kf_shadows_atrous_lds(AtrousArgs a)
{
    const uint2 BLK = block_xy<0>();
    constexpr int TW = 32 + 2 * STEP, TH = 8 + 2 * STEP;
    __shared__ uint32_t s_in[TH * TW];
    __shared__ float4   s_nz[TH * TW];   // unit normal, linear z
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int bx0 = (int)BLK.x * 32, by0 = a.y0 + (int)BLK.y * 8;
    if (bx0 >= a.w) return;   // a padding column of the launch (block_map.h grid_cols)
    const int x = bx0 + lx, y = by0 + ly;
    const bool inside = x < a.w && y < a.y1;
    const uint32_t cls = inside ? a.tile_class[(size_t)(y >> 3) * a.tiles_x + (x >> 3)] : 0u;
#if FT_ATROUS_VOTE
    if (__syncthreads_or((int)cls))   // a workgroup of shadow tiles only stages nothing
#endif
    {
        const int ry0 = a.y0 > 0 ? a.y0 : 0, ry1 = a.y1 < a.h ? a.y1 : a.h;
        for (int i = threadIdx.x; i < TH * TW; i += 256)
        {
            const int  cy = i / TW, cx = i - cy * TW;
            const int  gx = bx0 - STEP + cx, gy = by0 - STEP + cy;
            const bool res = gx >= 0 && gx < a.w && gy >= ry0 && gy < ry1;
            const uint32_t so = res ? (uint32_t)(gy * a.w + gx) : (uint32_t)(ry0 * a.w);
            const uint32_t v  = fm::ld<uint32_t>(a.in.p, so * 4u);
            const uint2    nd = fm::ld<uint2>(a.nd, so * 8u);
            const f3       n  = fm::oct_unit(nd.x);
            s_in[i] = res ? v : 0u;
            s_nz[i] = res ? make_float4(n.x, n.y, n.z, fm::hi(nd.y)) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        __syncthreads();
    }
    if (!inside) return;
    const uint32_t o = (uint32_t)(y * a.w + x);
    uint32_t* outp  = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(a.out) + o * 4u);
    uint32_t* out2p = a.out2 ? reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(a.out2) + o * 4u) : nullptr;
    if (!cls)
    {
        *outp = 0u; // shadows_denoise_copy_shadow_tiles.comp:35
        if (out2p) *out2p = 0u;
        return;
    }
    const int      ci  = (ly + STEP) * TW + (lx + STEP);
    const uint32_t c   = s_in[ci];
    const float4   cnz = s_nz[ci];
    // compute_variance_center (:65-88): 3x3 gaussian of the variance channel, unit taps
    float var = 0.0f;
#pragma unroll
    for (int k = 0; k < 9; k++)
    {
        const int   xx = k % 3 - 1, yy = k / 3 - 1;
        const float kw = (xx == 0 ? (yy == 0 ? 0.25f : 0.125f) : (yy == 0 ? 0.125f : 0.0625f));
        var += fm::hi(s_in[ci + yy * TW + xx]) * kw;
    }
    uint32_t result = c;
    if (!(cnz.w < 0.0f))
    {
        const f3    cn = mk3(cnz.x, cnz.y, cnz.z);
        const float cv = fm::lo(c);
        EdgeK ek;
        ek.kz        = 1.44269504088896341f * fm::rcp(a.sigma_depth);
        ek.phi_n     = a.phi_normal;
        ek.n32       = N32;
        ek.inv_phi_l = fm::rcp(a.phi_visibility * fm::sqrt1(fm::fmax_(0.0f, 1e-10f + var)));
        float sum_w = 1.0f, sum_v = cv, sum_var = fm::hi(c);
#pragma unroll
        for (int t = 0; t < 8; t++)
        {
            const int      k = t < 4 ? t : t + 1, xx = k % 3 - 1, yy = k / 3 - 1;
            const float    kk = (xx == 0 ? 1.0f : 2.0f / 3.0f) * (yy == 0 ? 1.0f : 2.0f / 3.0f);
            const int      ti = ci + yy * STEP * TW + xx * STEP;
            const uint32_t tv = s_in[ti];
            const float4   tn = s_nz[ti];
            const float    sv = fm::lo(tv);
            const float    wv = fm::mul_rn(edge_weight_fast(ek, cnz.w, tn.w, cn, mk3(tn.x, tn.y, tn.z), cv, sv), kk);
            sum_w += wv;
            sum_v += wv * sv;
            sum_var += (wv * wv) * fm::hi(tv);
        }
        const float iw = fm::rcp(sum_w);
        float ov = sum_v * iw;
        const float ovar = sum_var * (iw * iw);
        if (a.power != 0.0f) ov = fm::powf_(fm::fmax_(ov, 0.0f), a.power);
        result = fm::pack2(ov, ovar);
    }
    *outp = result;
    if (out2p) *out2p = result;
}

// A-trous iterations 0 and 1 (tap distances 1 and 2, radius 1) in ONE launch: iteration 0's image never leaves LDS.
// Per 32 x TH output tile: region A = the tile + 3 texels all round is staged once (value/variance + decoded normal + linear z);
// iteration 0 runs on region B = the tile + 2 (iteration 1 reads B at distance 2 for its taps, at distance 1 around the centre only for its
// 3x3 variance prefilter; iteration 0 reads A at distance 1) and
// is rounded to RG16F exactly as the stored image would be; iteration 1 runs on the tile.  Saves one launch, one 4 B/px image write
// and the second pass's re-fetch of the 8 B/px normal/depth image at the cost of (B + tile) / (2 tile) filter evaluations.
// Texel rules are those of kf_shadows_atrous_lds: a texel outside the image / the band's resident rows is (0, normal 0) in BOTH
// images (weight exactly 0), a texel of a shadow tile is 0 after iteration 0 (copy_shadow_tiles.comp), so the result is
// bit-identical to the two launches (tests/test_gpu_fused.py).  a.out2 = feedback copy of iteration 1, a.out_first2 = of iteration 0.
template <int TH, bool N32>
__global__ __launch_bounds__(256) void kf_shadows_atrous01(AtrousArgs a, uint32_t* out_first2, float power1)
{
    const uint2 BLK = block_xy<0>();
    constexpr int AW = 38, AH = TH + 6, BW = 36, BH = TH + 4;   // A = tile + 3 all round, B = tile + 2 (round-3 advisor: one ring too many each)
    __shared__ uint32_t s_in[AH * AW];
    __shared__ float4   s_nz[AH * AW];   // unit normal, linear z
    __shared__ uint32_t s_mid[BH * BW];  // iteration 0, as the RG16F image would hold it
    const int bx0 = (int)BLK.x * 32, by0 = a.y0 + (int)BLK.y * TH;
    if (bx0 >= a.w) return;   // a padding column of the launch (block_map.h grid_cols)
    const int ry0 = a.y0 > 0 ? a.y0 : 0, ry1 = a.y1 < a.h ? a.y1 : a.h;
    // tile classes of this workgroup's own tiles: all shadow => every output is 0 and nothing is staged
    int any_cls = 0;
    if ((int)threadIdx.x < 4 * (TH / 8))
    {
        const int tx = (bx0 >> 3) + ((int)threadIdx.x & 3), ty = (by0 >> 3) + ((int)threadIdx.x >> 2);
        if (tx < a.tiles_x && ty * 8 < a.y1) any_cls = a.tile_class[(size_t)ty * a.tiles_x + tx];
    }
    if (!__syncthreads_or(any_cls))
    {
        for (int j = threadIdx.x; j < TH * 32; j += 256)
        {
            const int x = bx0 + (j & 31), y = by0 + (j >> 5);
            if (x >= a.w || y >= a.y1) continue;
            const uint32_t o = (uint32_t)(y * a.w + x);
            a.out[o] = 0u;
            if (a.out2) a.out2[o] = 0u;
            if (out_first2) out_first2[o] = 0u;
        }
        return;
    }
    for (int i = threadIdx.x; i < AH * AW; i += 256)
    {
        const int  cy = i / AW, cx = i - cy * AW;
        const int  gx = bx0 - 3 + cx, gy = by0 - 3 + cy;
        const bool res = gx >= 0 && gx < a.w && gy >= ry0 && gy < ry1;
        const uint32_t so = res ? (uint32_t)(gy * a.w + gx) : (uint32_t)(ry0 * a.w);
        const uint32_t v  = fm::ld<uint32_t>(a.in.p, so * 4u);
        const uint2    nd = fm::ld<uint2>(a.nd, so * 8u);
        const f3       n  = fm::oct_unit(nd.x);
        s_in[i] = res ? v : 0u;
        s_nz[i] = res ? make_float4(n.x, n.y, n.z, fm::hi(nd.y)) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    __syncthreads();
    EdgeK ek;
    ek.kz    = 1.44269504088896341f * fm::rcp(a.sigma_depth);
    ek.phi_n = a.phi_normal;
    ek.n32   = N32;
    // one filter evaluation: centre index ci in the A grid, values from `img` (stride IW, centre index vi), taps at distance D
    auto filter = [&](const uint32_t* img, int IW, int vi, int ci, int D, float power) -> uint32_t {
        const uint32_t c   = img[vi];
        const float4   cnz = s_nz[ci];
        float var = 0.0f;
#pragma unroll
        for (int k = 0; k < 9; k++)
        {
            const int   xx = k % 3 - 1, yy = k / 3 - 1;
            const float kw = (xx == 0 ? (yy == 0 ? 0.25f : 0.125f) : (yy == 0 ? 0.125f : 0.0625f));
            var += fm::hi(img[vi + yy * IW + xx]) * kw;
        }
        if (cnz.w < 0.0f) return c;
        const f3    cn = mk3(cnz.x, cnz.y, cnz.z);
        const float cv = fm::lo(c);
        ek.inv_phi_l = fm::rcp(a.phi_visibility * fm::sqrt1(fm::fmax_(0.0f, 1e-10f + var)));
        float sum_w = 1.0f, sum_v = cv, sum_var = fm::hi(c);
#pragma unroll
        for (int t = 0; t < 8; t++)
        {
            const int      k = t < 4 ? t : t + 1, xx = k % 3 - 1, yy = k / 3 - 1;
            const float    kk = (xx == 0 ? 1.0f : 2.0f / 3.0f) * (yy == 0 ? 1.0f : 2.0f / 3.0f);
            const uint32_t tv = img[vi + yy * D * IW + xx * D];
            const float4   tn = s_nz[ci + yy * D * AW + xx * D];
            const float    sv = fm::lo(tv);
            const float    wv = fm::mul_rn(edge_weight_fast(ek, cnz.w, tn.w, cn, mk3(tn.x, tn.y, tn.z), cv, sv), kk);
            sum_w += wv;
            sum_v += wv * sv;
            sum_var += (wv * wv) * fm::hi(tv);
        }
        const float iw = fm::rcp(sum_w);
        float ov = sum_v * iw;
        const float ovar = sum_var * (iw * iw);
        if (power != 0.0f) ov = fm::powf_(fm::fmax_(ov, 0.0f), power);
        return fm::pack2(ov, ovar);
    };
    // iteration 0 on region B
    for (int j = threadIdx.x; j < BH * BW; j += 256)
    {
        const int  cy = j / BW, cx = j - cy * BW;
        const int  gx = bx0 - 2 + cx, gy = by0 - 2 + cy;
        const bool res = gx >= 0 && gx < a.w && gy >= ry0 && gy < ry1;
        uint32_t   r = 0u;
        if (res && a.tile_class[(size_t)(gy >> 3) * a.tiles_x + (gx >> 3)])
        {
            const int ci = (cy + 1) * AW + cx + 1;
            r = filter(s_in, AW, ci, ci, 1, 0.0f);
        }
        s_mid[j] = r;
        if (out_first2 && res && cx >= 2 && cx < 34 && cy >= 2 && cy < 2 + TH && gy < a.y1) out_first2[(uint32_t)(gy * a.w + gx)] = r;
    }
    __syncthreads();
    // iteration 1 on the tile
    for (int j = threadIdx.x; j < TH * 32; j += 256)
    {
        const int lx = j & 31, ly = j >> 5, x = bx0 + lx, y = by0 + ly;
        if (x >= a.w || y >= a.y1) continue;
        const uint32_t o = (uint32_t)(y * a.w + x);
        uint32_t r = 0u;
        if (a.tile_class[(size_t)(y >> 3) * a.tiles_x + (x >> 3)]) r = filter(s_mid, BW, (ly + 2) * BW + lx + 2, (ly + 3) * AW + lx + 3, 2, power1);
        a.out[o] = r;
        if (a.out2) a.out2[o] = r;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// ao_denoise_reprojection.comp:191-260, tolerance mode; spp = 1..4 sample planes (BASELINE configs[2]: 4)
#ifndef FT_AO_EU
#define FT_AO_EU 7   // 72 VGPRs; 8 (64 VGPRs) was right before the cold copy of the pixel program existed: with it the allocator spills in the hot path (+40 %)
#endif
template <bool MULTI, int GEO>
__global__ __launch_bounds__(64 * FT_WAVES, FT_AO_EU) void kf_ao_temporal(AOTemporalArgs a)
{
    if (tile_order_rides<64 * FT_WAVES>(a.sort)) return;   // the trace kernel's next launch order, as extra grid rows (tile_order.h)
    const uint2 BLK = block_xy<0>();
    __shared__ uint32_t s_mask[FT_WAVES][4][18];
    __shared__ MaskRows s_rows[FT_WAVES];
    const int  lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // grid = (tile columns / FT_WAVES, tile rows): no integer division per wave
    const int  txr = BLK.x * FT_WAVES + wave;
    const bool tile_ok = txr < a.tiles_x;
    const int  tx = tile_ok ? txr : 0, ty = (int)BLK.y + a.tile_y0;
    const int  lx = lane & 7, ly = lane >> 3;
    const int  x = tx * 8 + lx, y = ty * 8 + ly;
    const bool in_image = tile_ok && x < a.w && y < a.h && y >= a.y0 && y < a.y1;
    const bool edge     = tile_ok && (x >= a.w || y >= a.h);
    const uint32_t pix  = in_image ? (uint32_t)(y * a.w + x) : (uint32_t)(a.gb2.y0 * a.w);
    const float d_raw   = fm::ld<float>(a.depth.p, pix * 4u);
    const uint2 cg2_raw = fm::ld<uint2>(a.gb2.p, pix * 8u);
    const uint32_t cg3y_raw = fm::ld<uint32_t>(a.gb3.p, pix * 8u + 4u);
    build_mask_rows<true>(s_rows[wave], s_mask[wave], a.mask, MULTI ? a.spp : 1, a.mw, a.mh, tx, ty, a.y0, a.y1, lane, tile_ok);
    if (!tile_ok) return;
    // the pixel's program, run once — and a second time (cold copy) by a wave that noted a history tap on a knife edge: kf_shadows_temporal
    bool flag = false;
    auto pixel = [&](const float d_in, const uint2 cg2_in, const uint32_t cg3y_in, const bool redo) -> uint32_t {
        const float    d    = edge ? 0.0f : d_in;
        const uint2    cg2  = edge ? make_uint2(0u, 0u) : cg2_in;
        const uint32_t cg3y = edge ? 0u : cg3y_in;
        const bool     live = (in_image || edge) && d != 1.0f;
        Reproj<2, false, false, GEO> rp;
        rp.M = a.vpi; rp.pgb2 = a.pgb2.p; rp.pgb3 = a.pgb3.p; rp.pdepth = a.pdepth.p; rp.hist = a.hist.p; rp.hist_moments = nullptr; rp.hist_len = a.hist_len.p;
        rp.geo = a.geo_hist;
        rp.g = HistGeom { a.w, a.h, a.pgb2.y0, a.pgb2.y1 };
        uint32_t ovr = 0u, known = 0u;
        if (redo && live)   // see kf_shadows_temporal
        {
            rp.issue(x, y, d, cg2.y, fm::lo(cg3y), 0.0f, fm::oct_unit(cg2.x), mk3(0, 0, 0), nullptr, 0.0f);
            ReprojOut r0;
            rp.template resolve<true>(r0);
            if (rp.near13)
            {
                known = rp.near13;
                ovr   = rp.exact_bits(known, x, y, d, cg2.x, cg2.y);
                if ((((ovr ^ rp.valid13) & known) & 0xfu) != 0u)
                {
                    const uint32_t rest = 0x1ff0u & ~known;
                    ovr |= rp.exact_bits(rest, x, y, d, cg2.x, cg2.y);
                    known |= rest;
                }
            }
        }
        if (live) rp.issue(x, y, d, cg2.y, fm::lo(cg3y), 0.0f, fm::oct_unit(cg2.x), mk3(0, 0, 0), nullptr, 0.0f);
        if (!redo && a.apron_flag && __ballot(live && in_image && y >= a.band_y0 && y < a.band_y1 && rp.apron_miss) && lane == 0) atomicOr(a.apron_flag, 1u);
        int sum, own;
        mask_window<MULTI>(s_rows[wave], lx, ly, sum, own);   // popcounts while the history loads are in flight
        const float mean = fm::div_by_inrange((float)sum, div_prepare(289.0f * (float)a.spp));
        flag = false;
        rp.near13 = 0u;
        if (in_image || edge)
        {
            float out = 1.0f, hlen = 0.0f;
            if (live)
            {
                const float ao = fm::div_by_inrange((float)own, div_prepare((float)a.spp));
                ReprojOut   r;
                const bool  success = rp.template resolve<true>(r, ovr, known);
                hlen = fm::fmin_(32.0f, success ? r.length + 1.0f : 1.0f);
                float hao = r.col[0];
                if (success)
                {
                    const float sd = fm::sqrt1(fm::fmax_(fm::var_rn(mean, mean), 0.0f));
                    hao = fm::fmin_(fm::fmax_(hao, mean - 0.5f * sd), mean + 0.5f * sd);
                }
                const float al = success ? fm::fmax_(a.alpha, fm::rcp_nr(hlen)) : 1.0f;
                out = fm::mix_rn(hao, ao, al);
            }
            if (in_image)
            {
                *reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(a.out) + pix * 2u)     = fm::half_bits(out);
                *reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(a.out_len) + pix * 2u) = fm::half_bits(hlen);
                // next frame's reprojection record: this pixel's oct normal, mesh id and AO value (copies of what the images hold)
                if (a.geo_out) *reinterpret_cast<uint2*>(reinterpret_cast<char*>(a.geo_out) + pix * 8u) = make_uint2(cg2.x, (cg3y & 0xffffu) | ((uint32_t)fm::half_bits(out) << 16));
            }
            flag = out < 1.0f;
        }
        return live ? rp.near13 : 0u;
    };
    const uint32_t doubt = pixel(d_raw, cg2_raw, cg3y_raw, false);
#if HR_TAP_REDO
    if (__any(doubt != 0u))   // rare: the cold copy (re-reads the centre texels: nothing of the first run is kept)
        pixel(fm::ld<float>(a.depth.p, pix * 4u), fm::ld<uint2>(a.gb2.p, pix * 8u), fm::ld<uint32_t>(a.gb3.p, pix * 8u + 4u), true);
#endif
    const unsigned long long any = __ballot(flag);
    if (lane == 0) a.tile_class[(size_t)ty * a.tiles_x + tx] = any ? 1 : 0;
}


// ------------------------------------------------------------------------------------------------------------------------
// ao_denoise_bilateral_blur.comp:75-139, tolerance mode.  LDS-tiled: the tile + its apron along the blur direction is decoded
// ONCE (linear eye depth, unit normal, AO value: 20 B per texel in LDS) instead of per tap — 8 taps of a radius-4 blur re-derived
// an octahedral decode + a normalisation + a reciprocal each.
HR_DEV float gaussian_weight_fast(float offset, float deviation)
{
    const float d2 = deviation * deviation;
    return fm::rsq(2.0f * HR_M_PI * d2) * fm::expf_(-(offset * offset) * fm::rcp(2.0f * d2));
}

template <int RADIUS>
__global__ __launch_bounds__(256) void kf_ao_blur(AOBlurArgs a)
{
    const uint2 BLK = block_xy<0>();
    constexpr int kMaxTexels = 32 * (8 + 2 * RADIUS);   // the vertical pass is the larger footprint
    __shared__ float4 s_nz[kMaxTexels];                 // unit normal, linear eye depth
    __shared__ float  s_ao[kMaxTexels];
    __shared__ float  s_gauss[2 * RADIUS + 1];
    const int SW = 32 + 2 * RADIUS * a.dx, SH = 8 + 2 * RADIUS * a.dy;
    const int ox = (int)BLK.x * 32 - RADIUS * a.dx, oy = a.y0 + (int)BLK.y * 8 - RADIUS * a.dy;
    // the centre's own reads (tile class, depth) travel with the staging loads: one memory round trip before the barrier
    const int  lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int  x = (int)BLK.x * 32 + lx, y = a.y0 + (int)BLK.y * 8 + ly;
    const bool have = x < a.w && y < a.y1;
    const uint32_t o = have ? (uint32_t)(y * a.w + x) : (uint32_t)(a.y0 * a.w);
    const uint8_t  tclass = a.tile_class[have ? (size_t)(y >> 3) * a.tiles_x + (x >> 3) : 0];
    const float    cdepth = fm::ld<float>(a.depth.p, o * 4u);
    if ((int)threadIdx.x <= 2 * RADIUS) s_gauss[threadIdx.x] = gaussian_weight_fast((float)((int)threadIdx.x - RADIUS), (float)RADIUS * (1.0f / 1.5f));
    for (int i = threadIdx.x; i < SW * SH; i += 256)
    {
        const int  sy = i / SW, sx = i - sy * SW;
        const int  px = ox + sx, py = oy + sy;
        const bool ok = !(px < 0 || py < a.y0 || px >= a.w || py >= a.y1);
        const uint32_t off = ok ? (uint32_t)(py * a.w + px) : (uint32_t)(a.y0 * a.w);
        float    z  = fm::ld<float>(a.depth.p, off * 4u);
        uint32_t g2 = fm::ld<uint32_t>(a.gb2.p, off * 8u);
        uint16_t v  = fm::ld<uint16_t>(a.in.p, off * 2u);
        if (!ok) { z = 0.0f; g2 = 0u; v = 0; }          // texel fetches outside the image read 0 (pinned rule), then decode as usual
        const f3 n = fm::oct_unit(g2);
        s_nz[i] = make_float4(n.x, n.y, n.z, fm::rcp_nr(fm::mad_rn(a.zbp[2], z, a.zbp[3])));
        s_ao[i] = (float)__builtin_bit_cast(_Float16, v);
    }
    __syncthreads();
    if (!have) return;
    uint16_t* outp = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(a.out) + o * 2u);
    const uint16_t one = 0x3c00u;
    if (!tclass) { *outp = one; return; }   // cleared image (ray_traced_ao.cpp:1048-1055)
    if (cdepth == 1.0f) { *outp = one; return; }
    const int    ci = (ly + RADIUS * a.dy) * SW + lx + RADIUS * a.dx, stride = a.dy * SW + a.dx;
    const float4 c  = s_nz[ci];
    float total_ao = s_ao[ci], total_w = 1.0f;
#pragma unroll
    for (int i = -RADIUS; i <= RADIUS; i++)
    {
        if (i == 0) continue;
        const float4 t  = s_nz[ci + i * stride];
        const float  wZ = fm::exp2f_(-__builtin_fabsf(c.w - t.w) * 1.44269504088896341f);
        const float  wN = fm::pow32(fm::sat(c.x * t.x + c.y * t.y + c.z * t.z));
        const float  w  = s_gauss[i + RADIUS] * (fm::exp2f_((1.0f + wZ) * -1.44269504088896341f) * wN);
        total_ao += w * s_ao[ci + i * stride];
        total_w += w;
    }
    *outp = fm::half_bits(total_ao * fm::rcp(fm::fmax_(total_w, 0.0001f)));
}

// Both passes of the separable blur in ONE launch: the X result of the tile's rows + a RADIUS-row apron above
// and below never leaves LDS.  Per 32 x TH output tile: (32 + 2R) x (TH + 2R) input texels are fetched and decoded once (the
// two-launch form fetched the tile + apron twice and wrote / re-read the X image: 2 x 74 MB of counter traffic at 1080p for 33 MB of
// algorithmic bytes), the X pass runs on 32 x (TH + 2R) texels, the Y pass on 32 x TH.  The X result is rounded to fp16 exactly as
// the R16F blur image stored it, tile classes / sky texels / out-of-band rows follow the same rules texel by texel, so the output
// is BIT-IDENTICAL to kf_ao_blur run twice (tests/test_gpu_fused.py); IMG_BLUR0 is not written in this form.
template <int RADIUS, int TH>
__global__ __launch_bounds__(256) void kf_ao_blur_xy(AOBlurArgs a)
{
    const uint2 BLK = block_xy<FT_AO_BLUR_ROWS>();
    if ((int)BLK.x * 32 >= a.w) return;   // a padding column of the launch (block_map.h grid_cols)
    constexpr int SW = 32 + 2 * RADIUS, SH = TH + 2 * RADIUS;
    __shared__ float4  s_nz[SH * SW];      // unit normal, linear eye depth
    __shared__ float   s_ao[SH * SW];
    __shared__ uint8_t s_kind[SH * SW];    // 0: not resident (reads as zeros), 1: cleared tile or sky (1.0), 2: filtered
    __shared__ float   s_x[SH * 32];       // X pass, as the R16F image would hold it
    __shared__ float   s_gauss[2 * RADIUS + 1];
    const int bx0 = (int)BLK.x * 32, by0 = a.y0 + (int)BLK.y * TH;
    const int ox = bx0 - RADIUS, oy = by0 - RADIUS;
    if ((int)threadIdx.x <= 2 * RADIUS) s_gauss[threadIdx.x] = gaussian_weight_fast((float)((int)threadIdx.x - RADIUS), (float)RADIUS * (1.0f / 1.5f));
    for (int i = threadIdx.x; i < SW * SH; i += 256)
    {
        const int  sy = i / SW, sx = i - sy * SW;
        const int  px = ox + sx, py = oy + sy;
        const bool ok = !(px < 0 || py < a.y0 || px >= a.w || py >= a.y1);
        const uint32_t off = ok ? (uint32_t)(py * a.w + px) : (uint32_t)(a.y0 * a.w);
        float    z  = fm::ld<float>(a.depth.p, off * 4u);
        uint32_t g2 = fm::ld<uint32_t>(a.gb2.p, off * 8u);
        uint16_t v  = fm::ld<uint16_t>(a.in.p, off * 2u);
        const uint8_t tc = a.tile_class[ok ? (size_t)(py >> 3) * a.tiles_x + (px >> 3) : 0];
        s_kind[i] = !ok ? 0 : ((!tc || z == 1.0f) ? 1 : 2);
        if (!ok) { z = 0.0f; g2 = 0u; v = 0; }          // texel fetches outside the image read 0 (pinned rule), then decode as usual
        const f3 n = fm::oct_unit(g2);
        s_nz[i] = make_float4(n.x, n.y, n.z, fm::rcp_nr(fm::mad_rn(a.zbp[2], z, a.zbp[3])));
        s_ao[i] = (float)__builtin_bit_cast(_Float16, v);
    }
    __syncthreads();
    // X pass over the tile's columns and SH rows
    for (int j = threadIdx.x; j < SH * 32; j += 256)
    {
        const int sy = j >> 5, sx = j & 31, ci = sy * SW + sx + RADIUS;
        const uint8_t kind = s_kind[ci];
        float r = kind == 1 ? 1.0f : 0.0f;
        if (kind == 2)
        {
            const float4 c = s_nz[ci];
            float total_ao = s_ao[ci], total_w = 1.0f;
#pragma unroll
            for (int i = -RADIUS; i <= RADIUS; i++)
            {
                if (i == 0) continue;
                const float4 t  = s_nz[ci + i];
                const float  wZ = fm::exp2f_(-__builtin_fabsf(c.w - t.w) * 1.44269504088896341f);
                const float  wN = fm::pow32(fm::sat(c.x * t.x + c.y * t.y + c.z * t.z));
                const float  w  = s_gauss[i + RADIUS] * (fm::exp2f_((1.0f + wZ) * -1.44269504088896341f) * wN);
                total_ao += w * s_ao[ci + i];
                total_w += w;
            }
            r = (float)__builtin_bit_cast(_Float16, fm::half_bits(total_ao * fm::rcp(fm::fmax_(total_w, 0.0001f))));
        }
        s_x[j] = r;
    }
    __syncthreads();
    // Y pass over the tile
    for (int j = threadIdx.x; j < TH * 32; j += 256)
    {
        const int ly = j >> 5, lx = j & 31, x = bx0 + lx, y = by0 + ly;
        if (x >= a.w || y >= a.y1) continue;
        uint16_t* outp = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(a.out) + (uint32_t)(y * a.w + x) * 2u);
        const int ci = (ly + RADIUS) * SW + lx + RADIUS, xi = (ly + RADIUS) * 32 + lx;
        if (s_kind[ci] != 2) { *outp = 0x3c00u; continue; }   // cleared image (ray_traced_ao.cpp:1048-1055) / sky
        const float4 c = s_nz[ci];
        float total_ao = s_x[xi], total_w = 1.0f;
#pragma unroll
        for (int i = -RADIUS; i <= RADIUS; i++)
        {
            if (i == 0) continue;
            const float4 t  = s_nz[ci + i * SW];
            const float  wZ = fm::exp2f_(-__builtin_fabsf(c.w - t.w) * 1.44269504088896341f);
            const float  wN = fm::pow32(fm::sat(c.x * t.x + c.y * t.y + c.z * t.z));
            const float  w  = s_gauss[i + RADIUS] * (fm::exp2f_((1.0f + wZ) * -1.44269504088896341f) * wN);
            total_ao += w * s_x[xi + i * 32];
            total_w += w;
        }
        *outp = fm::half_bits(total_ao * fm::rcp(fm::fmax_(total_w, 0.0001f)));
    }
}

// run-time radius: plain gathers
__global__ __launch_bounds__(256) void kf_ao_blur_generic(AOBlurArgs a)
{
    const uint2 BLK = block_xy<0>();
    __shared__ float s_gauss[2 * 32 + 1];
    if ((int)threadIdx.x <= 2 * a.radius) s_gauss[threadIdx.x] = gaussian_weight_fast((float)((int)threadIdx.x - a.radius), (float)a.radius * (1.0f / 1.5f));
    __syncthreads();
    const int x = BLK.x * 32 + (threadIdx.x & 31), y = a.y0 + BLK.y * 8 + (threadIdx.x >> 5);
    if (x >= a.w || y >= a.y1) return;
    const uint32_t o = (uint32_t)(y * a.w + x);
    uint16_t* outp = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(a.out) + o * 2u);
    const uint16_t one = 0x3c00u;
    if (!a.tile_class[(size_t)(y >> 3) * a.tiles_x + (x >> 3)]) { *outp = one; return; }
    const float d = fm::ld<float>(a.depth.p, o * 4u);
    if (d == 1.0f) { *outp = one; return; }
    const float cd = fm::rcp_nr(fm::mad_rn(a.zbp[2], d, a.zbp[3]));
    const f3    cn = fm::oct_unit(fm::ld<uint32_t>(a.gb2.p, o * 8u));
    float total_ao = (float)__builtin_bit_cast(_Float16, fm::ld<uint16_t>(a.in.p, o * 2u)), total_w = 1.0f;
    for (int i = -a.radius; i <= a.radius; i++)
    {
        if (i == 0) continue;
        const int  px = x + a.dx * i, py = y + a.dy * i;
        const bool ok = !(px < 0 || py < a.y0 || px >= a.w || py >= a.y1);
        const uint32_t off = ok ? (uint32_t)(py * a.w + px) : o;
        float    z  = fm::ld<float>(a.depth.p, off * 4u);
        uint32_t g2 = fm::ld<uint32_t>(a.gb2.p, off * 8u);
        uint16_t v  = fm::ld<uint16_t>(a.in.p, off * 2u);
        if (!ok) { z = 0.0f; g2 = 0u; v = 0; }
        const float sd = fm::rcp_nr(fm::mad_rn(a.zbp[2], z, a.zbp[3]));
        const float wZ = fm::exp2f_(-__builtin_fabsf(cd - sd) * 1.44269504088896341f);
        const float wN = fm::pow32(fm::sat(fm::dot(cn, fm::oct_unit(g2))));
        const float w  = s_gauss[i + a.radius] * (fm::exp2f_((1.0f + wZ) * -1.44269504088896341f) * wN);
        total_ao += w * (float)__builtin_bit_cast(_Float16, v);
        total_w += w;
    }
    *outp = fm::half_bits(total_ao * fm::rcp(fm::fmax_(total_w, 0.0001f)));
}


// ------------------------------------------------------------------------------------------------------------------------
// reflections_denoise_reprojection.comp, tolerance mode.  32x8 tile per workgroup; the 48x24 input colours around it are staged
// in LDS as they are (fp16), and the 17x17 neighbourhood mean / standard deviation (:133-157 — 289 taps per pixel in the
// reference) become SEPARABLE sums: a horizontal pass (each thread forms four adjacent 17-tap row sums from one 20-texel window,
// sharing the 14-texel core) into LDS, then 17 row sums per pixel.
#define FR_TW 32
#define FR_TH 8
#define FR_R 8
#ifndef FR_EU
#define FR_EU 5        // minimum waves per SIMD the register allocator leaves room for (see FR_ALIAS)
#endif
#ifndef FR_ALIAS
#define FR_ALIAS 0     // 1: the horizontal sums overwrite the colour tile (the sums are held in registers across a barrier): 27.6 -> 18.4 KB of
                       // LDS per workgroup, 5 -> 8 workgroups per CU if the registers allow (FR_EU)
#endif
template <int GEO>
__global__ __launch_bounds__(256, FR_EU) void kf_refl_temporal(ReflTemporalArgs a)
{
    if (tile_order_rides<256>(a.sort)) return;   // the trace kernel's next launch order, as extra grid rows (tile_order.h)
    const uint2 BLK = block_xy<0>();
    constexpr int CW = FR_TW + 2 * FR_R, CH = FR_TH + 2 * FR_R;   // 48 x 24
#if FR_ALIAS
    __shared__ float4 s_raw[CH * FR_TW + CH * FR_TW / 2];          // 18.4 KB: first the colour tile (9.2 KB), then ha | hb
    uint2  (*s_col)[CW]   = reinterpret_cast<uint2 (*)[CW]>(s_raw);
    float4 (*s_ha)[FR_TW] = reinterpret_cast<float4 (*)[FR_TW]>(s_raw);
    float2 (*s_hb)[FR_TW] = reinterpret_cast<float2 (*)[FR_TW]>(s_raw + CH * FR_TW);
#else
    __shared__ uint2  s_col[CH][CW];        // rgb + ray length, fp16 as stored by the trace
    __shared__ float4 s_ha[CH][FR_TW];      // horizontal sums: sum r, g, b, sum r^2
    __shared__ float2 s_hb[CH][FR_TW];      //                  sum g^2, b^2
#endif
    __shared__ int    s_flag[4];
    const int bx0 = BLK.x * FR_TW, by0 = a.y0 + BLK.y * FR_TH;
    if (bx0 >= a.w) return;   // a padding column of the launch (block_map.h grid_cols)
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int x = bx0 + lx, y = by0 + ly;
    const bool have = x < a.w && y < a.y1;
    // memory round trips in this order: (1) centre texels + the 48x24 colour tile, under which nothing else can run; (2) the history
    // taps, issued after the horizontal LDS pass and left in flight under the vertical one
    const uint32_t o = have ? (uint32_t)(y * a.w + x) : (uint32_t)(a.y0 * a.w);
    const float d   = fm::ld<float>(a.depth.p, o * 4u);
    const uint2 cg2 = fm::ld<uint2>(a.gb2.p, o * 8u), cg3 = fm::ld<uint2>(a.gb3.p, o * 8u);
    if (threadIdx.x < 4) s_flag[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < CH * CW; i += 256)
    {
        const int cy = i / CW, cx = i - cy * CW;
        s_col[cy][cx] = a.in.raw(bx0 - FR_R + cx, by0 - FR_R + cy);
    }
    __syncthreads();
#if FR_ALIAS
    const uint2 cq = s_col[(threadIdx.x >> 5) + FR_R][(threadIdx.x & 31) + FR_R];   // the centre texel, before the tile is overwritten
    float4 ha_out[4];
    float2 hb_out[4];
#endif
    if (threadIdx.x < CH * (FR_TW / 4))
    {
        const int r = threadIdx.x >> 3, g = (threadIdx.x & 7) * 4;   // row, first of four adjacent outputs
        float c1[3] = { 0, 0, 0 }, c2[3] = { 0, 0, 0 };              // core: texels g+3 .. g+16, in every one of the four windows
        f3    e[6];                                                  // edge texels g+0, g+1, g+2, g+17, g+18, g+19
#pragma unroll
        for (int t = 0; t < 20; t++)
        {
            const uint2 q = s_col[r][g + t];
            const f3    c = mk3(fm::lo(q.x), fm::hi(q.x), fm::lo(q.y));
            if (t >= 3 && t <= 16)
            {
                c1[0] += c.x; c1[1] += c.y; c1[2] += c.z;
                c2[0] += c.x * c.x; c2[1] += c.y * c.y; c2[2] += c.z * c.z;
            }
            else e[t < 3 ? t : t - 14] = c;
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            // window j = texels g+j .. g+j+16 = core + edge texels {j, .., 2} + {3, .., 2+j}
            float s1[3] = { c1[0], c1[1], c1[2] }, s2[3] = { c2[0], c2[1], c2[2] };
#pragma unroll
            for (int k = 0; k < 6; k++)
                if ((k < 3 && k >= j) || (k >= 3 && k - 3 < j))
                {
                    s1[0] += e[k].x; s1[1] += e[k].y; s1[2] += e[k].z;
                    s2[0] += e[k].x * e[k].x; s2[1] += e[k].y * e[k].y; s2[2] += e[k].z * e[k].z;
                }
#if FR_ALIAS
            ha_out[j] = make_float4(s1[0], s1[1], s1[2], s2[0]);
            hb_out[j] = make_float2(s2[1], s2[2]);
#else
            s_ha[r][g + j] = make_float4(s1[0], s1[1], s1[2], s2[0]);
            s_hb[r][g + j] = make_float2(s2[1], s2[2]);
#endif
        }
    }
#if FR_ALIAS
    __syncthreads();   // every thread has read its part of the colour tile
    if (threadIdx.x < CH * (FR_TW / 4))
    {
        const int r = threadIdx.x >> 3, g = (threadIdx.x & 7) * 4;
#pragma unroll
        for (int j = 0; j < 4; j++) { s_ha[r][g + j] = ha_out[j]; s_hb[r][g + j] = hb_out[j]; }
    }
#endif
    __syncthreads();
    const bool  live = have && d != 1.0f;
#if !FR_ALIAS
    const uint2 cq = s_col[ly + FR_R][lx + FR_R];
#endif
    // The pixel's program from the history taps on.  It runs once; a wave in which some history tap sat on a knife edge of the validity test (Reproj::tap_valid notes
    // it) runs it a SECOND time — a cold copy, as in kf_shadows_temporal / kf_ao_temporal — in which the noted taps take the parity kernels' verdicts
    // (Reproj::exact_bits), and stores again (late round 6: tools/fuzz_tolerance.py 6361 #385 had ONE such tap flip a pixel's moments by 92 fp16 ulp, and five a-trous
    // iterations of radius 2 spread that beyond the chain's counted allowance; docs/EXPERIMENTS.md R6.14).  The separable sums are still in LDS for it.
    bool flag = false;
    auto pixel = [&](const bool redo) -> uint32_t {
    Reproj<8, true, true, GEO> rp;
    rp.M = a.vpi; rp.pgb2 = a.pgb2.p; rp.pgb3 = a.pgb3.p; rp.pdepth = a.pdepth.p; rp.hist = a.hist.p; rp.hist_moments = a.hist_moments.p; rp.hist_len = nullptr;
    rp.geo = a.geo_hist;
    rp.g = HistGeom { a.w, a.h, a.pgb2.y0, a.pgb2.y1 };
    uint32_t ovr = 0u, known = 0u;
    if (redo && live)
    {
        rp.issue(x, y, d, cg2.y, fm::lo(cg3.y), fm::hi(cg3.x), fm::oct_unit(cg2.x), mk3(a.cam[0], a.cam[1], a.cam[2]), a.pvp, fm::hi(cq.y));
        ReprojOut r0;
        rp.template resolve<true>(r0);
        if (rp.near13)
        {
            known = rp.near13;
            ovr   = rp.exact_bits(known, x, y, d, cg2.x, cg2.y);
            if ((((ovr ^ rp.valid13) & known) & 0xfu) != 0u)   // a bilinear verdict flipped: the 3x3 fallback may run where it did not (or the other way round)
            {
                const uint32_t rest = 0x1ff0u & ~known;
                ovr |= rp.exact_bits(rest, x, y, d, cg2.x, cg2.y);
                known |= rest;
            }
        }
    }
    if (live) rp.issue(x, y, d, cg2.y, fm::lo(cg3.y), fm::hi(cg3.x), fm::oct_unit(cg2.x), mk3(a.cam[0], a.cam[1], a.cam[2]), a.pvp, fm::hi(cq.y));
    if (!redo && a.apron_flag && live && y >= a.band_y0 && y < a.band_y1 && rp.apron_miss) atomicOr(a.apron_flag, 1u);   // rare
    rp.near13 = 0u;
    // vertical pass of the separable sums (LDS only) while the history taps are in flight
    float s1[3] = { 0, 0, 0 }, s2[3] = { 0, 0, 0 };
#pragma unroll 6   // in groups: fully unrolled, the 34 LDS reads are hoisted together and cost 100 VGPRs next to the 25 tap registers
    for (int dy = 0; dy <= 2 * FR_R; dy++)
    {
        const float4 ha = s_ha[ly + dy][lx];
        const float2 hb = s_hb[ly + dy][lx];
        s1[0] += ha.x; s1[1] += ha.y; s1[2] += ha.z; s2[0] += ha.w; s2[1] += hb.x; s2[2] += hb.y;
    }
    if (have)
    {
        const float roughness = fm::lo(cg3.x);
        float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f, r3 = 0.0f, m0 = 0.0f, m1 = 0.0f, hl = 0.0f;
        if (live)
        {
            const f3  color = mk3(fm::lo(cq.x), fm::hi(cq.x), fm::lo(cq.y));
            ReprojOut r;
            const bool success = rp.template resolve<true>(r, ovr, known);
            hl = fm::fmin_(32.0f, success ? r.length + 1.0f : 1.0f);
            f3 history = mk3(r.col[0], r.col[1], r.col[2]);
            if (success)
            {
                float cv[3], ext[3], cen[3], mx = 0.0f;
                const float hv[3] = { history.x, history.y, history.z };
#pragma unroll
                for (int c = 0; c < 3; c++)
                {
                    const float mean = s1[c] * (1.0f / 289.0f);
                    const float sd   = fm::sqrt1(fm::fmax_(s2[c] * (1.0f / 289.0f) - mean * mean, 0.0f));
                    // clip_aabb (:111-129) with aabb = mean -/+ sd: centre = mean, extent = sd + 0.001
                    cen[c] = mean; ext[c] = sd + 0.001f;
                    cv[c]  = hv[c] - mean;
                    mx     = fm::fmax_(mx, __builtin_fabsf(cv[c] * fm::rcp(ext[c])));
                }
                if (mx > 1.0f)
                {
                    const float im = fm::rcp(mx);
                    history = mk3(cen[0] + cv[0] * im, cen[1] + cv[1] * im, cen[2] + cv[2] * im);
                }
            }
            const float max_acc = a.moving ? 8.0f : hl;
            const float ia = fm::div_by_inrange(1.0f, div_prepare(max_acc));   // correctly rounded (1 <= max_acc <= 32): the moments' blend factor, see Reproj::resolve
            const float al = success ? fm::fmax_(a.alpha, ia) : 1.0f;
            const float am = success ? fm::fmax_(a.moments_alpha, ia) : 1.0f;
            const float lum = luminance(color);
            m0 = fm::mix_rn(r.mom[0], lum, am);
            m1 = fm::mix_rn(r.mom[1], lum * lum, am);
            r3 = fm::fmax_(0.0f, fm::var_rn(m1, m0));
            r0 = fm::mix_rn(history.x, color.x, al); r1 = fm::mix_rn(history.y, color.y, al); r2 = fm::mix_rn(history.z, color.z, al);
        }
        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(a.out_moments) + o * 8u) = make_uint2(fm::pack2(m0, m1), fm::pack2(hl, 0.0f));
        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(a.out) + o * 8u)         = make_uint2(fm::pack2(r0, r1), fm::pack2(r2, r3));
        // geometry record {oct normal, mesh id | linear z}: copies of the G-buffer's words, for this frame's a-trous taps and the next
        // frame's reprojection
        if (a.geo_out) *reinterpret_cast<uint2*>(reinterpret_cast<char*>(a.geo_out) + o * 8u) = make_uint2(cg2.x, cg3.y);
        if (d != 1.0f && roughness >= 0.05f) flag = (a.approximate_with_ddgi == 1) ? (roughness <= 0.75f) : true;
    }
    return live ? rp.near13 : 0u;
    };
    const uint32_t doubt = pixel(false);
#if HR_TAP_REDO
    if (__any(doubt != 0u)) pixel(true);   // rare: the cold copy
#endif
    // tile classification per 8x8 tile (4 tiles per workgroup): :262-272
    if (flag) atomicOr(&s_flag[lx >> 3], 1);
    __syncthreads();
    if (threadIdx.x < 4)
    {
        const int tx = (bx0 >> 3) + threadIdx.x, ty = by0 >> 3;
        if (tx < a.tiles_x) a.tile_class[(size_t)ty * a.tiles_x + tx] = s_flag[threadIdx.x] ? 1 : 0;
    }
}

// reflections_denoise_atrous.comp + copy_tiles (tile class), tolerance mode
template <int STEP, bool N32>
__global__ __launch_bounds__(256) void kf_refl_atrous(ReflAtrousArgs a)
{
    const uint2 BLK = block_xy<0>();
    const int x = BLK.x * 32 + (threadIdx.x & 31), y = a.y0 + BLK.y * 8 + (threadIdx.x >> 5);
    if (x >= a.w || y >= a.y1) return;
    const uint32_t o = (uint32_t)(y * a.w + x);
    const uint2    c = fm::ld<uint2>(a.in.p, o * 8u);
    uint2          result = c;
    if (a.tile_class[(size_t)(y >> 3) * a.tiles_x + (x >> 3)])
    {
        const int  step = STEP > 0 ? STEP : a.step, R = STEP > 0 ? 1 : a.radius;
        const int  fx0 = (int)BLK.x * 32, fy0 = a.y0 + (int)BLK.y * 8, reach = step * R > 1 ? step * R : 1;
        const bool interior = fx0 - reach >= 0 && fx0 + 31 + reach < a.w && fy0 - reach >= a.in.y0 && fy0 + 7 + reach < a.in.y1;
        const f3    cc = mk3(fm::lo(c.x), fm::hi(c.x), fm::lo(c.y));
        const float center_luma = fm::fmax_(0.299f * cc.x + 0.587f * cc.y + 0.114f * cc.z, 0.0001f);
        float var = 0.0f;
#pragma unroll
        for (int k = 0; k < 9; k++)
        {
            const int   xx = k % 3 - 1, yy = k / 3 - 1;
            const float kw = (xx == 0 ? (yy == 0 ? 0.25f : 0.125f) : (yy == 0 ? 0.125f : 0.0625f));
            const uint32_t v = interior ? fm::ld<uint32_t>(a.in.p, (uint32_t)((y + yy) * a.w + (x + xx)) * 8u + 4u) : a.in.raw(x + xx, y + yy).y;
            var += fm::hi(v) * kw;
        }
        const uint32_t g2x = fm::ld<uint32_t>(a.gb2.p, o * 8u);
        const uint2    g3  = fm::ld<uint2>(a.gb3.p, o * 8u);
        const float    d = fm::ld<float>(a.depth.p, o * 4u), roughness = fm::lo(g3.x);
        if (d == 1.0f) result = make_uint2(0u, 0u);
        else if (!(roughness < 0.05f || (a.approximate_with_ddgi == 1 && roughness > 0.75f)))
        {
            const f3    cn = fm::oct_unit(g2x);
            const float center_depth = fm::hi(g3.y);
            EdgeK ek;
            ek.kz        = 1.44269504088896341f * fm::rcp(a.sigma_depth);
            ek.phi_n     = a.phi_normal;
            ek.n32       = N32;
            ek.inv_phi_l = fm::rcp(a.phi_color * fm::sqrt1(fm::fmax_(0.0f, 1e-10f + var)));
            float sum_w = 1.0f, s0 = cc.x, s1 = cc.y, s2 = cc.z, s3 = fm::hi(c.y);
            auto tap = [&](uint2 q, uint32_t q2x, uint32_t q3y, float kk, bool ok) {
                const f3    sc = mk3(fm::lo(q.x), fm::hi(q.x), fm::lo(q.y));
                const float sl = fm::fmax_(0.299f * sc.x + 0.587f * sc.y + 0.114f * sc.z, 0.0001f);
                float wc = fm::mul_rn(edge_weight_fast(ek, center_depth, fm::hi(q3y), cn, fm::oct_unit(q2x), center_luma, sl), kk);
                if (!ok) wc = 0.0f;
                sum_w += wc;
                s0 += wc * sc.x; s1 += wc * sc.y; s2 += wc * sc.z;
                s3 += (wc * wc) * fm::hi(q.y);
            };
            if (STEP > 0)
            {
                uint2    t_in[8];
                uint32_t t_g2[8], t_g3[8];
                bool     t_ok[8];
                if (interior)
                {
#pragma unroll
                    for (int t = 0; t < 8; t++)
                    {
                        const int      k = t < 4 ? t : t + 1, xx = k % 3 - 1, yy = k / 3 - 1;
                        const uint32_t so = (uint32_t)((y + yy * STEP) * a.w + (x + xx * STEP));
                        t_ok[t] = true;
                        t_in[t] = fm::ld<uint2>(a.in.p, so * 8u);
                        if (a.geo) { const uint2 q = fm::ld<uint2>(a.geo, so * 8u); t_g2[t] = q.x; t_g3[t] = q.y; }   // uniform branch
                        else { t_g2[t] = fm::ld<uint32_t>(a.gb2.p, so * 8u); t_g3[t] = fm::ld<uint32_t>(a.gb3.p, so * 8u + 4u); }
                    }
                }
                else
                {
#pragma unroll
                    for (int t = 0; t < 8; t++)
                    {
                        const int k = t < 4 ? t : t + 1, xx = k % 3 - 1, yy = k / 3 - 1;
                        const int px = x + xx * STEP, py = y + yy * STEP;
                        t_ok[t] = px >= 0 && py >= 0 && px < a.w && py < a.h;
                        t_in[t] = a.in.raw(px, py); t_g2[t] = a.gb2.raw(px, py).x; t_g3[t] = a.gb3.raw(px, py).y;
                    }
                }
#pragma unroll
                for (int t = 0; t < 8; t++)
                {
                    const int k = t < 4 ? t : t + 1, xx = k % 3 - 1, yy = k / 3 - 1;
                    tap(t_in[t], t_g2[t], t_g3[t], (xx == 0 ? 1.0f : 2.0f / 3.0f) * (yy == 0 ? 1.0f : 2.0f / 3.0f), t_ok[t]);
                }
            }
            else
            {
                for (int yy = -R; yy <= R; yy++)
                    for (int xx = -R; xx <= R; xx++)
                    {
                        const int px = x + xx * step, py = y + yy * step;
                        if (px < 0 || py < 0 || px >= a.w || py >= a.h || (xx == 0 && yy == 0)) continue;
                        const int   axx = xx < 0 ? -xx : xx, ayy = yy < 0 ? -yy : yy;
                        const float kx = axx == 0 ? 1.0f : (axx == 1 ? 2.0f / 3.0f : 1.0f / 6.0f);
                        const float ky = ayy == 0 ? 1.0f : (ayy == 1 ? 2.0f / 3.0f : 1.0f / 6.0f);
                        tap(a.in.raw(px, py), a.gb2.raw(px, py).x, a.gb3.raw(px, py).y, kx * ky, true);
                    }
            }
            const float iw = fm::rcp(sum_w);
            result = make_uint2(fm::pack2(s0 * iw, s1 * iw), fm::pack2(s2 * iw, s3 * (iw * iw)));
        }
    }
    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(a.out) + o * 8u) = result;
    if (a.out2) *reinterpret_cast<uint2*>(reinterpret_cast<char*>(a.out2) + o * 8u) = result;
}


// Reflections a-trous iterations 0 and 1 in ONE launch (the scheme of kf_shadows_atrous01: A = tile + 4 staged once, iteration 0
// on B = tile + 3 rounded to RGBA16F as stored, iteration 1 on the tile).  Texel rules of kf_refl_atrous, texel by texel: outside
// the image -> weight 0 (staged with a zero normal); inside the image but outside the band's resident rows -> zeros that DO enter
// the sums (halo rows only), in both images; sky -> 0; copy tiles and mirror / DDGI-rough texels pass through.  Bit-identical to the
// two launches (tests/test_gpu_fused.py).
template <int TH, bool N32>
__global__ __launch_bounds__(256) void kf_refl_atrous01(ReflAtrousArgs a, uint2* out_first2)
{
    const uint2 BLK = block_xy<FT_RATROUS01_ROWS>();
    if ((int)BLK.x * 32 >= a.w) return;   // a padding column of the launch (block_map.h grid_cols)
    constexpr int AW = 38, AH = TH + 6, BW = 36, BH = TH + 4;   // A = tile + 3 all round, B = tile + 2 (round-3 advisor: one ring too many each)
    __shared__ uint2  s_in[AH * AW];
    __shared__ float4 s_nz[AH * AW];    // unit normal (0 outside the image), linear z
    __shared__ float  s_r[AH * AW];     // roughness; -1: sky texel
    __shared__ uint2  s_mid[BH * BW];
    const int bx0 = (int)BLK.x * 32, by0 = a.y0 + (int)BLK.y * TH;
    for (int i = threadIdx.x; i < AH * AW; i += 256)
    {
        const int  cy = i / AW, cx = i - cy * AW;
        const int  gx = bx0 - 3 + cx, gy = by0 - 3 + cy;
        const bool img = gx >= 0 && gx < a.w && gy >= 0 && gy < a.h;
        const bool res = img && gy >= a.in.y0 && gy < a.in.y1;
        const uint32_t so = res ? (uint32_t)(gy * a.w + gx) : (uint32_t)(a.in.y0 * a.w);
        uint2    c  = fm::ld<uint2>(a.in.p, so * 8u);
        uint32_t g2 = fm::ld<uint32_t>(a.gb2.p, so * 8u);
        uint2    g3 = fm::ld<uint2>(a.gb3.p, so * 8u);
        float    d  = fm::ld<float>(a.depth.p, so * 4u);
        if (!res) { c = make_uint2(0u, 0u); g2 = 0u; g3 = make_uint2(0u, 0u); d = 0.0f; }
        const f3 n = fm::oct_unit(g2);
        s_in[i] = c;
        s_nz[i] = img ? make_float4(n.x, n.y, n.z, fm::hi(g3.y)) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        s_r[i]  = d == 1.0f ? -1.0f : fm::lo(g3.x);
    }
    __syncthreads();
    EdgeK ek;
    ek.kz    = 1.44269504088896341f * fm::rcp(a.sigma_depth);
    ek.phi_n = a.phi_normal;
    ek.n32   = N32;
    auto filter = [&](const uint2* img, int IW, int vi, int ci, int D) -> uint2 {
        const uint2 c = img[vi];
        const float roughness = s_r[ci];
        if (roughness == -1.0f) return make_uint2(0u, 0u);
        if (roughness < 0.05f || (a.approximate_with_ddgi == 1 && roughness > 0.75f)) return c;
        float var = 0.0f;
#pragma unroll
        for (int k = 0; k < 9; k++)
        {
            const int   xx = k % 3 - 1, yy = k / 3 - 1;
            const float kw = (xx == 0 ? (yy == 0 ? 0.25f : 0.125f) : (yy == 0 ? 0.125f : 0.0625f));
            var += fm::hi(img[vi + yy * IW + xx].y) * kw;
        }
        const float4 cnz = s_nz[ci];
        const f3     cn  = mk3(cnz.x, cnz.y, cnz.z);
        const f3     cc  = mk3(fm::lo(c.x), fm::hi(c.x), fm::lo(c.y));
        const float  center_luma = fm::fmax_(0.299f * cc.x + 0.587f * cc.y + 0.114f * cc.z, 0.0001f);
        ek.inv_phi_l = fm::rcp(a.phi_color * fm::sqrt1(fm::fmax_(0.0f, 1e-10f + var)));
        float sum_w = 1.0f, s0 = cc.x, s1 = cc.y, s2 = cc.z, s3 = fm::hi(c.y);
#pragma unroll
        for (int t = 0; t < 8; t++)
        {
            const int    k = t < 4 ? t : t + 1, xx = k % 3 - 1, yy = k / 3 - 1;
            const float  kk = (xx == 0 ? 1.0f : 2.0f / 3.0f) * (yy == 0 ? 1.0f : 2.0f / 3.0f);
            const uint2  q  = img[vi + yy * D * IW + xx * D];
            const float4 tn = s_nz[ci + yy * D * AW + xx * D];
            const f3     sc = mk3(fm::lo(q.x), fm::hi(q.x), fm::lo(q.y));
            const float  sl = fm::fmax_(0.299f * sc.x + 0.587f * sc.y + 0.114f * sc.z, 0.0001f);
            const float  wc = fm::mul_rn(edge_weight_fast(ek, cnz.w, tn.w, cn, mk3(tn.x, tn.y, tn.z), center_luma, sl), kk);
            sum_w += wc;
            s0 += wc * sc.x; s1 += wc * sc.y; s2 += wc * sc.z;
            s3 += (wc * wc) * fm::hi(q.y);
        }
        const float iw = fm::rcp(sum_w);
        return make_uint2(fm::pack2(s0 * iw, s1 * iw), fm::pack2(s2 * iw, s3 * (iw * iw)));
    };
    for (int j = threadIdx.x; j < BH * BW; j += 256)
    {
        const int  cy = j / BW, cx = j - cy * BW;
        const int  gx = bx0 - 2 + cx, gy = by0 - 2 + cy;
        const bool res = gx >= 0 && gx < a.w && gy >= (a.y0 > 0 ? a.y0 : 0) && gy < (a.y1 < a.h ? a.y1 : a.h);
        const int  ci = (cy + 1) * AW + cx + 1;
        uint2      r = make_uint2(0u, 0u);
        if (res) r = a.tile_class[(size_t)(gy >> 3) * a.tiles_x + (gx >> 3)] ? filter(s_in, AW, ci, ci, 1) : s_in[ci];
        s_mid[j] = r;
        if (out_first2 && res && cx >= 2 && cx < 34 && cy >= 2 && cy < 2 + TH) out_first2[(uint32_t)(gy * a.w + gx)] = r;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < TH * 32; j += 256)
    {
        const int lx = j & 31, ly = j >> 5, x = bx0 + lx, y = by0 + ly;
        if (x >= a.w || y >= a.y1) continue;
        const uint32_t o = (uint32_t)(y * a.w + x);
        const int vi = (ly + 2) * BW + lx + 2;
        const uint2 r = a.tile_class[(size_t)(y >> 3) * a.tiles_x + (x >> 3)] ? filter(s_mid, BW, vi, (ly + 3) * AW + lx + 3, 2) : s_mid[vi];
        a.out[o] = r;
        if (a.out2) a.out2[o] = r;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// gi_sample_probe_grid.comp:75-99, tolerance mode; the 8-probe gather is ddgi_sample_fast.h (shared with the reflections' hit shading)
__global__ __launch_bounds__(256) void kf_ddgi_sample(DDGISampleArgs a)
{
    const uint2 BLK = block_xy<0>();
    const int x = BLK.x * 32 + (threadIdx.x & 31), y = a.y0 + BLK.y * 8 + (threadIdx.x >> 5);
    if (x >= a.w || y >= a.y1) return;
    const uint32_t o  = (uint32_t)(y * a.w + x);
    uint2* outp = reinterpret_cast<uint2*>(reinterpret_cast<char*>(a.out) + o * 8u);
    const float dp = fm::ld<float>(a.depth, o * 4u);
    if (dp == 1.0f) { *outp = make_uint2(0u, 0u); return; }
    const DDGIU& d = a.d;
    // (x + 0.5) / w correctly rounded (round 5, as in Reproj::issue): the clip -> world products cancel 3-4 digits, one ulp of tu moved P by up
    // to ~1e-3 (20 ulp of its coordinates) — more than the whole guard of the gather's knife-edge tests
    const float tu = fm::div_by_inrange((float)x + 0.5f, div_prepare((float)a.w)), tv = fm::div_by_inrange((float)y + 0.5f, div_prepare((float)a.h));
    const f3 P  = fm::unproject_at(fm::unproject_base(a.vpi, tu, tv), a.vpi, dp);
    const f3 N  = fm::oct_unit(fm::ld<uint32_t>(a.gb2, o * 8u));
    f3       Wo = mk3(a.cam[0] - P.x, a.cam[1] - P.y, a.cam[2] - P.z);
    const float iwo = fm::rsq(fm::dot(Wo, Wo));
    Wo = mk3(Wo.x * iwo, Wo.y * iwo, Wo.z * iwo);
    // the redo of a pixel with a probe on the Chebyshev knife edge needs k_ddgi_sample's own operands of this pixel (exact_predicates.h)
    auto exact_inputs = [&](f3& eP, f3& eN, f3& eWo) { exact::pixel_inputs(a.vpi, a.cam, x, y, a.w, a.h, dp, fm::ld<uint32_t>(a.gb2, o * 8u), eP, eN, eWo); };
#ifdef HR_DEBUG_DDGI_PIXEL   // developer build: -DHR_DEBUG_DDGI_PIXEL -DHR_DBG_X=.. -DHR_DBG_Y=.. -DHR_DBG_W=.. prints the gather's terms of one pixel
    const bool dbg = x == HR_DBG_X && y == HR_DBG_Y && a.w == HR_DBG_W;
#else
    const bool dbg = false;
#endif
    const f3 net = ddgi_fast::sample_irradiance_net<true>(d, P, N, Wo, a.irr, a.dep, exact_inputs, dbg);
    const float k = d.energy_preservation * (0.5f * HR_M_PI) * a.gi_intensity;
    *outp = make_uint2(fm::pack2(net.x * net.x * k, net.y * net.y * k), fm::pack2(net.z * net.z * k, 1.0f));
}

// ------------------------------------------------------------------------------------------------------------------------
// shadows_upsample.comp / ao_upsample.comp / reflections_upsample.comp, tolerance mode
template <int CH>
__global__ __launch_bounds__(256) void kf_upsample(UpsampleArgs a)
{
    const uint2 BLK = block_xy<0>();
    const int x = BLK.x * 32 + (threadIdx.x & 31), y = BLK.y * 8 + (threadIdx.x >> 5);
    if (x >= a.W || y >= a.H) return;
    const uint32_t o  = (uint32_t)(y * a.W + x);
    uint16_t*      op = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(a.out) + o * (2u * CH));
    const float hi_depth = fm::hi(fm::ld<uint32_t>(a.G3, o * 8u + 4u));
    if (hi_depth == -1.0f)
    {
        const uint16_t sv = fm::half_bits(a.sky_value);
#pragma unroll
        for (int c = 0; c < CH; c++) op[c] = sv;
        return;
    }
    const f3    hn = fm::oct_unit(fm::ld<uint32_t>(a.G2, o * 8u));
    // correctly rounded quotients (the texel addressing below is a discrete decision); the divisors are launch constants
    const float tu = fm::div_by_inrange((float)x + 0.5f, div_prepare((float)a.W)), tv = fm::div_by_inrange((float)y + 0.5f, div_prepare((float)a.H));
    const float tsx = fm::div_by_inrange(1.0f, div_prepare((float)a.w)), tsy = fm::div_by_inrange(1.0f, div_prepare((float)a.h));
    float up[CH], total_w = 0.0f;
#pragma unroll
    for (int c = 0; c < CH; c++) up[c] = 0.0f;
    uint32_t t3[4], t2[4];
    uint16_t tin[4][CH];
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const float kx = (i == 1) ? 1.0f : (i == 2 ? -1.0f : 0.0f), ky = (i == 0) ? 1.0f : (i == 3 ? -1.0f : 0.0f);
        int sx = fm::tap_texel(tu, kx, tsx, (float)a.w), sy = fm::tap_texel(tv, ky, tsy, (float)a.h);   // fast_math.h: the reference's rounding
        sx = clampi(sx, 0, a.w - 1); sy = clampi(sy, 0, a.h - 1);
        const uint32_t so = (uint32_t)(sy * a.w + sx);
        t3[i] = fm::ld<uint32_t>(a.g3, so * 8u + 4u); t2[i] = fm::ld<uint32_t>(a.g2, so * 8u);
        const uint16_t* ip = reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(a.in) + so * (uint32_t)(2 * a.in_channels));
#pragma unroll
        for (int c = 0; c < CH; c++) tin[i][c] = ip[c];
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const float cd = fm::hi(t3[i]);
        // compute_edge_stopping_weight, NORMAL weight only: wL = 1 (edge_stopping.glsl:53-59)
        const float wZ = fm::exp2f_(-__builtin_fabsf(hi_depth - cd) * 1.44269504088896341f);
        const float wN = fm::pow32(fm::sat(fm::dot(hn, fm::oct_unit(t2[i]))));
        const float wt = (cd == -1.0f) ? 0.0f : fm::exp2f_((1.0f + wZ) * -1.44269504088896341f) * wN;
#pragma unroll
        for (int c = 0; c < CH; c++) up[c] += (float)__builtin_bit_cast(_Float16, tin[i][c]) * wt;
        total_w += wt;
    }
    const float iw = fm::rcp(fm::fmax_(total_w, 0.00000001f));
#pragma unroll
    for (int c = 0; c < CH; c++)
    {
        float r = up[c] * iw;
        if (a.power != 0.0f) r = fm::powf_(r, a.power);
        op[c] = fm::half_bits(r);
    }
}

} // namespace

namespace hr {

// a riding sort (TileOrder::ride) goes behind the `rows` grid rows of a temporal launch of `grid_x` workgroups per row: the grid with it
static dim3 grid_with_sort(TileSortArgs& t, int grid_x, int rows)
{
    t.row0 = rows;
    return dim3(grid_x, rows + (t.groups ? cdiv(t.groups, grid_x) : 0));
}
// ... and the history apron's records (GeoApronArgs) behind the sort: ~4 texels per thread
static dim3 grid_with_apron(dim3 grid, GeoApronArgs& g, int threads)
{
    const int total = g.out ? ((g.a1 - g.a0) + (g.b1 - g.b0)) * g.w : 0;
    if (total <= 0) { g.row0 = -1; return grid; }
    g.row0 = (int)grid.y;
    return dim3(grid.x, grid.y + (unsigned)cdiv(total, (int)grid.x * threads * 4));
}

void launch_shadows_temporal_fast(const TemporalArgs& a_, int n_tiles, hipStream_t st)
{
    TemporalArgs a = a_;
    const dim3 grid = grid_with_apron(grid_with_sort(a.sort, grid_cols(cdiv(a.tiles_x, FT_WAVES), 4), a.tiles_y), a.apron, 64 * FT_WAVES);
    if (a.geo_hist) hipLaunchKernelGGL(kf_shadows_temporal<1>, grid, dim3(64 * FT_WAVES), 0, st, a);
    else hipLaunchKernelGGL(kf_shadows_temporal<0>, grid, dim3(64 * FT_WAVES), 0, st, a);
}

void launch_shadows_atrous_fast(const AtrousArgs& a, hipStream_t st)
{
    const dim3 grid(grid_cols(cdiv(a.w, 32), (a.step >= 4 || !FT_ATROUS_LDS) ? 0 : 7), cdiv(a.y1 - a.y0, 8));
    const bool n32 = a.phi_normal == 32.0f;
#define HR_LAUNCH_ATROUS(K, S) \
    do { if (n32) hipLaunchKernelGGL((K<S, true>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((K<S, false>), grid, dim3(256), 0, st, a); } while (0)
    // tap distances 1 and 2 through LDS (apron 1-2 texels: 14.4 / 14.6 µs against 15.8 / 15.9 for per-pixel gathers at 1080p, event
    // times); at 4 and 8 the apron is as large as the tile and the gathers win (15.7 / 18.7 against 16.1 / 15.9)
    if (a.radius == 1 && a.step == 1 && FT_ATROUS_LDS) HR_LAUNCH_ATROUS(kf_shadows_atrous_lds, 1);
    else if (a.radius == 1 && a.step == 2 && FT_ATROUS_LDS) HR_LAUNCH_ATROUS(kf_shadows_atrous_lds, 2);
    else if (a.radius == 1 && a.step == 1) HR_LAUNCH_ATROUS(kf_shadows_atrous, 1);
    else if (a.radius == 1 && a.step == 2) HR_LAUNCH_ATROUS(kf_shadows_atrous, 2);
    else if (a.radius == 1 && a.step == 4) HR_LAUNCH_ATROUS(kf_shadows_atrous, 4);
    else if (a.radius == 1 && a.step == 8) HR_LAUNCH_ATROUS(kf_shadows_atrous, 8);
    else HR_LAUNCH_ATROUS(kf_shadows_atrous, 0);
#undef HR_LAUNCH_ATROUS
}

#ifndef FT_ATROUS01_TH
#define FT_ATROUS01_TH 16
#endif
// iterations 0 + 1 in one launch; `a` describes iteration 0 (a.in = temporal output, a.step == 1), a.out = the image iteration 1
// writes, a.out2 / out_first2 = the feedback copy of iteration 1 / 0, power1 = iteration 1's power.  false: not available
bool launch_shadows_atrous01_fast(const AtrousArgs& a, uint32_t* out_first2, float power1, hipStream_t st)
{
    if (a.radius != 1 || a.step != 1) return false;
    const dim3 grid(grid_cols(cdiv(a.w, 32), 7), cdiv(a.y1 - a.y0, FT_ATROUS01_TH));
    if (a.phi_normal == 32.0f) hipLaunchKernelGGL((kf_shadows_atrous01<FT_ATROUS01_TH, true>), grid, dim3(256), 0, st, a, out_first2, power1);
    else hipLaunchKernelGGL((kf_shadows_atrous01<FT_ATROUS01_TH, false>), grid, dim3(256), 0, st, a, out_first2, power1);
    return true;
}

void launch_ao_temporal_fast(const AOTemporalArgs& a_, int n_tiles, hipStream_t st)
{
    AOTemporalArgs a = a_;
    const dim3 grid = grid_with_sort(a.sort, grid_cols(cdiv(a.tiles_x, FT_WAVES), 5), a.tiles_y);
    if (a.geo_hist && a.geo_band)
    {
        if (a.spp > 1) hipLaunchKernelGGL((kf_ao_temporal<true, 1>), grid, dim3(64 * FT_WAVES), 0, st, a);
        else hipLaunchKernelGGL((kf_ao_temporal<false, 1>), grid, dim3(64 * FT_WAVES), 0, st, a);
    }
    else if (a.geo_hist)
    {
        if (a.spp > 1) hipLaunchKernelGGL((kf_ao_temporal<true, 2>), grid, dim3(64 * FT_WAVES), 0, st, a);
        else hipLaunchKernelGGL((kf_ao_temporal<false, 2>), grid, dim3(64 * FT_WAVES), 0, st, a);
    }
    else if (a.spp > 1) hipLaunchKernelGGL((kf_ao_temporal<true, 0>), grid, dim3(64 * FT_WAVES), 0, st, a);
    else hipLaunchKernelGGL((kf_ao_temporal<false, 0>), grid, dim3(64 * FT_WAVES), 0, st, a);
}

void launch_ao_blur_fast(const AOBlurArgs& a, hipStream_t st)
{
    const dim3 grid(cdiv(a.w, 32), cdiv(a.y1 - a.y0, 8));
    if (a.radius == 4) hipLaunchKernelGGL(kf_ao_blur<4>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(kf_ao_blur_generic, grid, dim3(256), 0, st, a);
}

#ifndef FT_AO_BLUR_TH
#define FT_AO_BLUR_TH 16   // tile height: 16 against 8 -> 24.8 against 26.5 us; 16 against 32 -> 1080p 23.5 against 25.7 us, 4K 66.5 against 70.5 (32: 1.56x instead of 1.88x apron reads, but 38.7 KB of LDS per workgroup)
#endif
bool launch_ao_blur_xy_fast(const AOBlurArgs& a, hipStream_t st)
{
    if (a.radius != 4) return false;   // other radii keep the two-launch form
    hipLaunchKernelGGL((kf_ao_blur_xy<4, FT_AO_BLUR_TH>), dim3(grid_cols(cdiv(a.w, 32), 8), cdiv(a.y1 - a.y0, FT_AO_BLUR_TH)), dim3(256), 0, st, a);
    return true;
}

void launch_refl_temporal_fast(const ReflTemporalArgs& a_, hipStream_t st)
{
    ReflTemporalArgs a = a_;
    const dim3 grid = grid_with_sort(a.sort, grid_cols(cdiv(a.w, FR_TW), 6), cdiv(a.y1 - a.y0, FR_TH));
    if (a.geo_hist) hipLaunchKernelGGL(kf_refl_temporal<1>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(kf_refl_temporal<0>, grid, dim3(256), 0, st, a);
}

void launch_refl_atrous_fast(const ReflAtrousArgs& a, hipStream_t st)
{
    const dim3 grid(grid_cols(cdiv(a.w, 32), 1), cdiv(a.y1 - a.y0, 8));
    const bool n32 = a.phi_normal == 32.0f;
#define HR_LAUNCH_RATROUS(S) \
    do { if (n32) hipLaunchKernelGGL((kf_refl_atrous<S, true>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((kf_refl_atrous<S, false>), grid, dim3(256), 0, st, a); } while (0)
    if (a.radius == 1 && a.step == 1) HR_LAUNCH_RATROUS(1);
    else if (a.radius == 1 && a.step == 2) HR_LAUNCH_RATROUS(2);
    else if (a.radius == 1 && a.step == 4) HR_LAUNCH_RATROUS(4);
    else if (a.radius == 1 && a.step == 8) HR_LAUNCH_RATROUS(8);
    else HR_LAUNCH_RATROUS(0);
#undef HR_LAUNCH_RATROUS
}

#ifndef FT_RATROUS01_TH
#define FT_RATROUS01_TH 16
#endif
bool launch_refl_atrous01_fast(const ReflAtrousArgs& a, uint2* out_first2, hipStream_t st)
{
    if (a.radius != 1 || a.step != 1) return false;
    const dim3 grid(grid_cols(cdiv(a.w, 32), 9), cdiv(a.y1 - a.y0, FT_RATROUS01_TH));
    if (a.phi_normal == 32.0f) hipLaunchKernelGGL((kf_refl_atrous01<FT_RATROUS01_TH, true>), grid, dim3(256), 0, st, a, out_first2);
    else hipLaunchKernelGGL((kf_refl_atrous01<FT_RATROUS01_TH, false>), grid, dim3(256), 0, st, a, out_first2);
    return true;
}

void launch_ddgi_sample_fast(const DDGISampleArgs& a, hipStream_t st)
{
    hipLaunchKernelGGL(kf_ddgi_sample, dim3(grid_cols(cdiv(a.w, 32), 3), cdiv(a.y1 - a.y0, 8)), dim3(256), 0, st, a);
}

void launch_upsample_fast(const UpsampleArgs& a, hipStream_t st)
{
    const dim3 grid(grid_cols(cdiv(a.W, 32), 2), cdiv(a.H, 8));
    if (a.channels == 4) hipLaunchKernelGGL(kf_upsample<4>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(kf_upsample<1>, grid, dim3(256), 0, st, a);
}

} // namespace hr
