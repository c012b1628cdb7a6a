// Tolerance-mode restatement of sample_irradiance (gi_common.glsl:188-320) — shared by the per-pixel probe-grid sample
// (kf_ddgi_sample, denoise_fast.hip) and by the reflections' hit shading in tolerance mode (k_refl_trace<.., FAST>, reflections.hip):
// hardware rcp / rsq / sqrt, contracted FMAs, re-associated sums; the ill-conditioned sub-expressions (atlas coordinates, bilinear
// weights of the depth moments, the Chebyshev variance) keep the reference's operation order (fast_math.h *_rn helpers).
// Hoisted out of the 8-probe loop: the octahedral texel offset of the surface normal (identical for every probe), the bias vector,
// 1 / grid_step; the probe's atlas cell needs no division (the atlas is nx * ny probes wide by construction, ddgi.cpp:197-201, checked
// in hr_ddgi_create); directions are octahedrally encoded without normalising them first (the L1 projection is scale free).
#pragma once
#include "fast_math.h"
#include "shading.h"

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
template <bool UNROLL>
HR_DEV f3 sample_irradiance_net(const DDGIU& d, f3 P, f3 N, f3 Wo, const AtlasRGBA& irr, const AtlasRG& dep)
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
    auto probe = [&](const int i) {
        const int ox = i & 1, oy = (i >> 1) & 1, oz = (i >> 2) & 1;
        const int cx = clampi(bx + ox, 0, nx - 1), cy = clampi(by + oy, 0, ny - 1), cz = clampi(bz + oz, 0, nz - 1);
        const int col = cx + cy * nx;   // probe p = col + cz * (nx * ny) sits in atlas cell (col, cz)
        const f3  pp  = mk3(g0.x + gs.x * (float)cx, g0.y + gs.y * (float)cy, g0.z + gs.z * (float)cz);
        const f3  ptp = mk3(P.x - pp.x + nb.x, P.y - pp.y + nb.y, P.z - pp.z + nb.z);
        const f3  tri = mk3(ox ? alpha.x : 1.0f - alpha.x, oy ? alpha.y : 1.0f - alpha.y, oz ? alpha.z : 1.0f - alpha.z);
        f3 tdp = mk3(pp.x - P.x, pp.y - P.y, pp.z - P.z);
        const float t = fm::fmax_(0.0001f, (fm::dot(tdp, N) * fm::rsq(fm::dot(tdp, tdp)) + 1.0f) * 0.5f);
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
            const float dm  = fm::fmax_(dist - mean, 0.0f);
            float che = variance * fm::rcp(variance + dm * dm);
            che       = fm::fmax_(che * che * che, 0.0f);
            weight *= (dist <= mean) ? 1.0f : che;
        }
        weight = fm::fmax_(0.000001f, weight);
        const uint32_t io = (uint32_t)((cz * (is + 2) + niy) * irr.w + col * (is + 2) + nix) * 8u;
        const uint2 q00 = fm::ld<uint2>(irr.p, io), q10 = fm::ld<uint2>(irr.p, io + 8u), q01 = fm::ld<uint2>(irr.p, io + irr_row), q11 = fm::ld<uint2>(irr.p, io + irr_row + 8u);
        const float ir = w00 * fm::lo(q00.x) + w10 * fm::lo(q10.x) + w01 * fm::lo(q01.x) + w11 * fm::lo(q11.x);
        const float ig = w00 * fm::hi(q00.x) + w10 * fm::hi(q10.x) + w01 * fm::hi(q01.x) + w11 * fm::hi(q11.x);
        const float ib = w00 * fm::lo(q00.y) + w10 * fm::lo(q10.y) + w01 * fm::lo(q01.y) + w11 * fm::lo(q11.y);
        if (weight < 0.2f) weight *= weight * weight * 25.0f;   // crush tiny weights (1 / 0.2^2)
        weight *= tri.x * tri.y * tri.z;
        sum.x += fm::sqrt1(ir) * weight; sum.y += fm::sqrt1(ig) * weight; sum.z += fm::sqrt1(ib) * weight;   // sqrt-space blending (LINEAR_BLENDING undefined)
        sum_w += weight;
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
    return net;
}

// the value gi_common.glsl's sample_irradiance returns: net^2 * energy_preservation * pi / 2
template <bool UNROLL>
HR_DEV f3 sample_irradiance(const DDGIU& d, f3 P, f3 N, f3 Wo, const AtlasRGBA& irr, const AtlasRG& dep)
{
    const f3    net = sample_irradiance_net<UNROLL>(d, P, N, Wo, irr, dep);
    const float k   = d.energy_preservation * (0.5f * HR_M_PI);
    return mk3(net.x * net.x * k, net.y * net.y * k, net.z * net.z * k);
}

} // namespace ddgi_fast
} // namespace hr

#pragma clang fp contract(off)   // the build's default (-ffp-contract=off): whatever follows this header is compiled as before
