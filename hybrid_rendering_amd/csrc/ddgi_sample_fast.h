// Tolerance-mode restatement of sample_irradiance (gi_common.glsl:188-320) — used by the per-pixel probe-grid sample (kf_ddgi_sample,
// denoise_fast.hip).  (Rounds 4-5 also used it for the reflections' hit shading in tolerance mode; round 6 returned those gathers to the
// parity arithmetic — one fp16 ulp in the trace image's colours did not survive the next frame's variance, docs/EXPERIMENTS.md R6.1.)
// hardware rcp / rsq / sqrt, contracted FMAs, re-associated sums; the ill-conditioned sub-expressions (atlas coordinates, bilinear
// weights of the depth moments, the Chebyshev variance) keep the reference's operation order (fast_math.h *_rn helpers).
// Hoisted out of the 8-probe loop: the octahedral texel offset of the surface normal (identical for every probe), the bias vector,
// 1 / grid_step; the probe's atlas cell needs no division (the atlas is nx * ny probes wide by construction, ddgi.cpp:197-201, checked
// in hr_ddgi_create); directions are octahedrally encoded without normalising them first (the L1 projection is scale free).
#pragma once
#include "fast_math.h"
#include "shading.h"
#include "exact_predicates.h"

#pragma clang fp contract(fast)

namespace hr {
namespace ddgi_fast {

HR_DEV void oct_encode_any(f3 v, float& ox, float& oy)
{
    const float inv = fm::rcp(__builtin_fabsf(v.x) + __builtin_fabsf(v.y) + __builtin_fabsf(v.z));
    float rx = v.x * inv, ry = v.y * inv;
    if (v.z < 0.0f)
    {
        const float nx = (1.0f - __builtin_fabsf(ry)) * (rx >= 0.0f ? 1.0f : -1.0f);
        const float ny = (1.0f - __builtin_fabsf(rx)) * (ry >= 0.0f ? 1.0f : -1.0f);
        rx = nx; ry = ny;
    }
    ox = rx; oy = ry;
}

// Bilinear footprint in an atlas.  A probe cell is framed by a one-texel border and the atlas by a one-texel margin
// (ddgi.cpp:197-201), so the 2x2 footprint of an in-cell coordinate never leaves the image; the base texel is still clamped
// (one v_med3 per axis) so that a NaN coordinate cannot produce a wild address.
struct Bilin { uint32_t o00; float fx, fy; };
HR_DEV Bilin bilin_setup(float x, float y, int w, int h)   // x, y in texel units, already minus the half texel
{
    const float fx0 = __builtin_floorf(x), fy0 = __builtin_floorf(y);
    Bilin b;
    b.fx = x - fx0; b.fy = y - fy0;
    const int x0 = clampi((int)fx0, 0, w - 2), y0 = clampi((int)fy0, 0, h - 2);
    b.o00 = (uint32_t)(y0 * w + x0);
    return b;
}

// N, Wo: unit vectors.  Returns `net`: the weighted sqrt-space irradiance BEFORE squaring and the energy / pi / 2 scaling
// (gi_common.glsl:299-320), NaN components replaced by 0.5 as the shader does.
// UNROLL: the eight probes fully unrolled (their fetches overlap; 102 VGPRs — right for the stand-alone sample kernel) or as a rolled loop
// (for the hit shading inside a trace kernel, whose occupancy the unrolled form would halve: shading.h sample_irradiance has the same note).
//
// WHERE THE FAST FORM IS NOT TRUSTED (round 5; tools/ddgi_pixel_terms.py and tools/ddgi_conditioning.py show the two cases on the CPU).  The
// gather is a weighted mean whose weights span 17 orders of magnitude (an occluded probe is crushed to 2.5e-17), and two things in it are not
// continuous:
//   (a) the trilinear factor of a probe plane the shading point lies ON is exactly 0 in the parity arithmetic ((P - base) / step == 0: walls on
//       the planes of a grid fitted to the scene's bounds) and ~1e-7 here (P, the quotient and the clamp each differ by an ulp).  When every
//       probe that does count is crushed, that 1e-7 of an unoccluded probe on the "zero" plane outweighs them all: the pixel shows ANOTHER
//       probe's irradiance.  Test: leak = 2e-7 x (largest weight before its trilinear factor) against 1e-3 x sum_w.
//   (b) INFINITE second moments.  The depth atlas is RG16F (ddgi.cpp:197-201) and holds the squared hit distance: beyond 256 units it
//       overflows to inf, and it stays inf (the hysteresis blend keeps it).  With var = inf the Chebyshev visibility
//       `dist <= mean ? 1 : (var / (var + (dist - mean)^2))^3` (gi_common.glsl:262-273) is a STEP at dist == mean (inf / inf = NaN -> 0 through
//       the max), and the bilinear fetch itself is discontinuous at every texel boundary next to an inf texel (mix(inf, x, t) is inf for
//       t < 1 and x at t == 1): an ulp of dist, mean or of the atlas coordinate decides whether a probe weighs 1 or 1e-6.  Only possible
//       when max_distance^2 overflows fp16 — a wave-uniform test on the uniforms, false for any grid whose probes are less than ~170 units
//       apart (the bench grid: 47-75) — so the per-probe tests sit behind a uniform branch: |dist - mean| <= g with var not finite
//       (g = 4e-6 (|P|_1 + |grid|_1 + dist + mean) + 4e-5 (max - min of the footprint's four mean texels) >= 4x the error of the fast
//       dist - mean), or the footprint within 2e-4 texels of a
//       texel boundary (the fast coordinates are good to a few ulp of ~150).  In that regime a third of the waves take the redo: it is the
//       regime in which the reference's own arithmetic is NaN-driven, and exactness is what can still be compared.
//       (Measured and dropped: the same note wherever var < 2e-4 mean^2, "steep" — +19 % on the 1080p bench frame for no outlier it
//       explained; every diagnosed outlier was (a) or an inf.)
//   (c) the shading point ON a probe (late round 6; tools/fuzz_tolerance.py 6351 #150, frame 1: the corner of the Cornell room — the grid is fitted to the scene's
//       bounds, so its corner probes sit on the bounding box's corners).  normalize(probe - P) is 0 * inf = NaN in the parity arithmetic, the NaN passes through
//       max() into the probe's weight and the sums, and the shader's NaN replacement makes the pixel net = 0.5; here v_max_f32 drops the NaN and the pixel
//       gets a finite, different value (532 fp16 ulp apart).  Test: |probe - P|^2 <= (1e-6 x (|P|_1 + |grid|_1))^2 — a hundred times the distance by which
//       the two arithmetics' probe positions (fused against unfused g0 + step * c) can disagree about "on".
// A pixel that meets any of them is REDONE with the parity kernels' own gather (shading.h sample_irradiance_net on their operands: exact_inputs) —
// a handful of pixels per frame on the bench scene (0.005 % of the pixels have sum_w < 1e-3), so the hot loop carries only the tests.
// exact_inputs(P, N, Wo): fills in the PARITY kernels' operands of this shading point (only called by the redo); a caller whose P, N, Wo
// already are those values passes them back (SameInputs).
struct SameInputs
{
    f3 P, N, Wo;
    HR_DEV void operator()(f3& p, f3& n, f3& wo) const { p = P; n = N; wo = Wo; }
};
template <bool UNROLL>
HR_DEV f3 sample_irradiance_pass(const DDGIU& d, f3 P, f3 N, f3 Wo, const AtlasRGBA& irr, const AtlasRG& dep, bool& redo, bool dbg = false)
{
    const f3 gs = mk3(d.grid_step[0], d.grid_step[1], d.grid_step[2]), g0 = mk3(d.grid_start_position[0], d.grid_start_position[1], d.grid_start_position[2]);
    const f3 igs = mk3(fm::rcp(gs.x), fm::rcp(gs.y), fm::rcp(gs.z));
    const int nx = d.probe_counts[0], ny = d.probe_counts[1], nz = d.probe_counts[2];
    const int bx = clampi((int)((P.x - g0.x) * igs.x), 0, nx - 1), by = clampi((int)((P.y - g0.y) * igs.y), 0, ny - 1), bz = clampi((int)((P.z - g0.z) * igs.z), 0, nz - 1);
    const f3 base = mk3(g0.x + gs.x * (float)bx, g0.y + gs.y * (float)by, g0.z + gs.z * (float)bz);
    const f3 alpha = mk3(fm::sat((P.x - base.x) * igs.x), fm::sat((P.y - base.y) * igs.y), fm::sat((P.z - base.z) * igs.z));
    const f3 nb = mk3((N.x + 3.0f * Wo.x) * d.normal_bias, (N.y + 3.0f * Wo.y) * d.normal_bias, (N.z + 3.0f * Wo.z) * d.normal_bias);
    // octahedral texel offset of N inside a probe's irradiance cell: the same for all eight probes
    float nox, noy;
    oct_encode_any(N, nox, noy);
    const int   is = d.irradiance_probe_side_length, ds = d.depth_probe_side_length;
    const float ncx = (nox + 1.0f) * 0.5f * (float)is + 1.5f, ncy = (noy + 1.0f) * 0.5f * (float)is + 1.5f;   // + 2 (cell origin) - 0.5 (texel centre)
    // the cell origins are integers: floor / fract of the in-cell coordinate serve all eight probes (four texel weights)
    const float nfx0 = __builtin_floorf(ncx), nfy0 = __builtin_floorf(ncy);
    const float nfx = ncx - nfx0, nfy = ncy - nfy0;
    const int   nix = clampi((int)nfx0, 1, is + 1), niy = clampi((int)nfy0, 1, is + 1);
    const float w00 = (1.0f - nfx) * (1.0f - nfy), w10 = nfx * (1.0f - nfy), w01 = (1.0f - nfx) * nfy, w11 = nfx * nfy;
    const uint32_t irr_row = (uint32_t)irr.w * 8u, dep_row = (uint32_t)dep.w * 4u;
    const DivBy    Ddw = div_prepare((float)dep.w), Ddh = div_prepare((float)dep.h);
    f3    sum = mk3(0.0f, 0.0f, 0.0f);
    float sum_w = 0.0f;
    // (b): the part of the guard that does not depend on the probe (probe positions lie inside the grid: |pp|_1 <= |grid|_1)
    const float G = 4e-6f * (__builtin_fabsf(P.x) + __builtin_fabsf(P.y) + __builtin_fabsf(P.z)
                             + fm::fmax_(__builtin_fabsf(g0.x), __builtin_fabsf(g0.x + gs.x * (float)(nx - 1))) + fm::fmax_(__builtin_fabsf(g0.y), __builtin_fabsf(g0.y + gs.y * (float)(ny - 1)))
                             + fm::fmax_(__builtin_fabsf(g0.z), __builtin_fabsf(g0.z + gs.z * (float)(nz - 1))));
    const bool wild_possible = !(d.max_distance * d.max_distance < 60000.0f);
    const float near2 = G * G * 0.0625f;   // (c): (1e-6 x (|P|_1 + |grid|_1))^2 — G is 4e-6 x that sum
    bool  noted  = false;
    float max_nt = 0.0f;   // (a): the largest weight before its trilinear factor
    auto probe = [&](const int i) {
        const int ox = i & 1, oy = (i >> 1) & 1, oz = (i >> 2) & 1;
        const int cx = clampi(bx + ox, 0, nx - 1), cy = clampi(by + oy, 0, ny - 1), cz = clampi(bz + oz, 0, nz - 1);
        const int col = cx + cy * nx;   // probe p = col + cz * (nx * ny) sits in atlas cell (col, cz)
        const f3  pp  = mk3(g0.x + gs.x * (float)cx, g0.y + gs.y * (float)cy, g0.z + gs.z * (float)cz);
        const f3  ptp = mk3(P.x - pp.x + nb.x, P.y - pp.y + nb.y, P.z - pp.z + nb.z);
        const f3  tri = mk3(ox ? alpha.x : 1.0f - alpha.x, oy ? alpha.y : 1.0f - alpha.y, oz ? alpha.z : 1.0f - alpha.z);
        f3 tdp = mk3(pp.x - P.x, pp.y - P.y, pp.z - P.z);
        const float tl2 = fm::dot(tdp, tdp);
        noted = noted || !(tl2 > near2);   // (c): the shading point ON a probe (see above)
        const float t = fm::fmax_(0.0001f, (fm::dot(tdp, N) * fm::rsq(tl2) + 1.0f) * 0.5f);
        float weight = t * t + 0.2f;
        if (d.visibility_test == 1)
        {
            float dox, doy;
            oct_encode_any(ptp, dox, doy);
            const float l2 = fm::dot(ptp, ptp), dist = l2 * fm::rsq(l2);
            float ax, ay;
            fm::atlas_coord_rn(dox, doy, col, cz, ds, (float)dep.w, (float)dep.h, Ddw, Ddh, ax, ay);
            const Bilin b = bilin_setup(ax, ay, dep.w, dep.h);
            const uint32_t bo = b.o00 * 4u;
            const uint32_t t00 = fm::ld<uint32_t>(dep.p, bo), t10 = fm::ld<uint32_t>(dep.p, bo + 4u), t01 = fm::ld<uint32_t>(dep.p, bo + dep_row), t11 = fm::ld<uint32_t>(dep.p, bo + dep_row + 4u);
            const float mean = fm::bilerp_rn(fm::lo(t00), fm::lo(t10), fm::lo(t01), fm::lo(t11), b.fx, b.fy);
            const float m2   = fm::bilerp_rn(fm::hi(t00), fm::hi(t10), fm::hi(t01), fm::hi(t11), b.fx, b.fy);
            const float variance = fm::cheb_variance_rn(mean, m2);
            const float dmr = dist - mean, dm = fm::fmax_(dmr, 0.0f);
            float che = variance * fm::rcp(variance + dm * dm);
            che       = fm::fmax_(che * che * che, 0.0f);
            const float vis = (dist <= mean) ? 1.0f : che;
            if (wild_possible)   // wave-uniform (see (b) above)
            {
                // ... + the error of `mean` itself: the fast atlas coordinate is good to ~1e-5 texels, and the four texels of a footprint can
                // differ by hundreds of units (a probe that sees a near wall beside the far end of the hall)
                const float m00 = fm::lo(t00), m10 = fm::lo(t10), m01 = fm::lo(t01), m11 = fm::lo(t11);
                const float spread = fm::fmax_(fm::fmax_(m00, m10), fm::fmax_(m01, m11)) - fm::fmin_(fm::fmin_(m00, m10), fm::fmin_(m01, m11));
                const float g = G + 4e-6f * (dist + mean) + 4e-5f * spread;
                const bool  edge = (b.fx < 2e-4f) | (b.fx > 1.0f - 2e-4f) | (b.fy < 2e-4f) | (b.fy > 1.0f - 2e-4f);
                noted = noted || edge || (!(variance < 3.0e38f) && __builtin_fabsf(dmr) <= g);
            }
            weight *= vis;
        }
        weight = fm::fmax_(0.000001f, weight);
        const uint32_t io = (uint32_t)((cz * (is + 2) + niy) * irr.w + col * (is + 2) + nix) * 8u;
        const uint2 q00 = fm::ld<uint2>(irr.p, io), q10 = fm::ld<uint2>(irr.p, io + 8u), q01 = fm::ld<uint2>(irr.p, io + irr_row), q11 = fm::ld<uint2>(irr.p, io + irr_row + 8u);
        const float ir = w00 * fm::lo(q00.x) + w10 * fm::lo(q10.x) + w01 * fm::lo(q01.x) + w11 * fm::lo(q11.x);
        const float ig = w00 * fm::hi(q00.x) + w10 * fm::hi(q10.x) + w01 * fm::hi(q01.x) + w11 * fm::hi(q11.x);
        const float ib = w00 * fm::lo(q00.y) + w10 * fm::lo(q10.y) + w01 * fm::lo(q01.y) + w11 * fm::lo(q11.y);
        if (weight < 0.2f) weight *= weight * weight * 25.0f;   // crush tiny weights (1 / 0.2^2)
        max_nt = fm::fmax_(max_nt, weight);
        weight *= tri.x * tri.y * tri.z;
        sum.x += fm::sqrt1(ir) * weight; sum.y += fm::sqrt1(ig) * weight; sum.z += fm::sqrt1(ib) * weight;   // sqrt-space blending (LINEAR_BLENDING undefined)
        sum_w += weight;
#ifdef HR_DEBUG_DDGI_PIXEL
        if (dbg) printf("[ddgi dbg] probe %d cell (%d %d %d) tri %.9g %.9g %.9g weight %.9g sum_w %.9g max_nt %.9g noted %d ir %.9g\n", i, cx, cy, cz, tri.x, tri.y, tri.z, weight, sum_w, max_nt, (int)noted, ir);
#endif
        };
    if constexpr (UNROLL)
    {
#pragma unroll
        for (int i = 0; i < 8; i++) probe(i);
    }
    else
    {
#pragma unroll 1
        for (int i = 0; i < 8; i++) probe(i);
    }
    const float iw = fm::rcp(sum_w);
    f3 net = mk3(sum.x * iw, sum.y * iw, sum.z * iw);
    net.x = (net.x != net.x) ? 0.5f : net.x; net.y = (net.y != net.y) ? 0.5f : net.y; net.z = (net.z != net.z) ? 0.5f : net.z;
    redo = noted || (2e-7f * max_nt > 1e-3f * sum_w);
#ifdef HR_DEBUG_DDGI_PIXEL
    if (dbg) printf("[ddgi dbg] P %.9g %.9g %.9g base %d %d %d alpha %.9g %.9g %.9g net %.9g %.9g %.9g redo %d\n", P.x, P.y, P.z, bx, by, bz, alpha.x, alpha.y, alpha.z, net.x, net.y, net.z, (int)redo);
#endif
    return net;
}

template <bool UNROLL, typename ExactInputs>
HR_DEV f3 sample_irradiance_net(const DDGIU& d, f3 P, f3 N, f3 Wo, const AtlasRGBA& irr, const AtlasRG& dep, const ExactInputs& exact_inputs, bool dbg = false)
{
    bool redo;
    f3   net = sample_irradiance_pass<UNROLL>(d, P, N, Wo, irr, dep, redo, dbg);
    if (redo)   // rare: the parity kernels' own gather for this shading point
    {
        f3 eP, eN, eWo;
        exact_inputs(eP, eN, eWo);
        net = hr::sample_irradiance_net(d, eP, eN, eWo, irr, dep);
#ifdef HR_DEBUG_DDGI_PIXEL
        if (dbg) printf("[ddgi dbg] redone: eP %.9g %.9g %.9g net %.9g %.9g %.9g\n", eP.x, eP.y, eP.z, net.x, net.y, net.z);
#endif
    }
    return net;
}

// the value gi_common.glsl's sample_irradiance returns: net^2 * energy_preservation * pi / 2
// P, N, Wo: the parity kernels' operands (the hit shading computes them with the parity arithmetic in both modes)
template <bool UNROLL>
HR_DEV f3 sample_irradiance(const DDGIU& d, f3 P, f3 N, f3 Wo, const AtlasRGBA& irr, const AtlasRG& dep)
{
    const f3    net = sample_irradiance_net<UNROLL>(d, P, N, Wo, irr, dep, SameInputs { P, N, Wo });
    const float k   = d.energy_preservation * (0.5f * HR_M_PI);
    return mk3(net.x * net.x * k, net.y * net.y * k, net.z * net.z * k);
}

} // namespace ddgi_fast
} // namespace hr

#pragma clang fp contract(off)   // the build's default (-ffp-contract=off): whatever follows this header is compiled as before
