// Parity-arithmetic operands for the REDO path of the tolerance-mode DDGI gather (ddgi_sample_fast.h, round 5).
//
// The tolerance mode computes with v_rcp / v_rsq / contracted FMAs.  sample_irradiance (gi_common.glsl:188-320) is a weighted mean whose
// weights span 17 orders of magnitude and which is not continuous in two places (a trilinear factor that is exactly 0 on a probe plane; the
// Chebyshev step at dist == mean where the depth variance is infinite or zero): there an fp32 ulp of difference against the oracle moved a
// texel by far more than the image tolerance (rounds 2-4 answered with a counted "outlier" allowance in tests/test_gpu_tolerance.py).
// ddgi_sample_fast.h now detects such shading points and redoes them with the parity kernels' own gather (shading.h) on the parity
// kernels' operands — computed HERE for a pixel of the probe-grid sample, with exactly the operations of k_ddgi_sample (ddgi.hip):
// correctly rounded / and sqrt, no contraction (this header is included before any `#pragma clang fp contract(fast)`; contraction is a
// per-instruction flag that inlining keeps).
// The same header holds the parity kernels' verdict on a reprojection tap (tap_valid) for the shadow / AO temporal kernels.
// (First attempt, measured and dropped: the exact re-evaluation INLINED into Reproj::resolve cost 12-100 % of the temporal kernels — 77 -> 96
// VGPRs, scratch; as non-inlined calls 300-800 B of scratch per call site.  What ships keeps the hot path free of it: docs/EXPERIMENTS.md R5.1.)
#pragma once
#include "device_math.h"
#include "reproject.h"
#include "shading.h"

namespace hr {
namespace exact {

// The parity kernels' verdict on one history tap (reproject.h:176-181 = is_reprojection_valid, reprojection.glsl:52-67): plane distance > 5,
// (n . n')^2 > 0.1 — a history tap is in or out, an fp32 ulp decides on the knife edge (one pixel in ~50 M in the shadow pass; the history
// is then re-weighted over other texels).  The tolerance-mode temporal kernels of the shadow and AO passes keep their fast test, note taps
// inside a guard band around either threshold and, for the (rare) wave that has one, ask THIS function and run the pixel again with its
// verdicts (denoise_fast.hip Reproj::exact_bits).  (x, y): the pixel; depth / c2x / c2y: its depth and the two words of its GB2 texel as the
// kernel uses them (edge threads: zeros); q2x / q3y / d: the tap's normal word, mesh-id word and depth (out-of-image taps: zeros).
HR_DEV bool tap_valid(const float* __restrict__ M, int x, int y, int w, int h, float depth, uint32_t c2x, uint32_t c2y, float cur_id, int hcx, int hcy,
                      uint32_t q2x, uint32_t q3y, float d)
{
    const float fw = (float)w, fh = (float)h;
    const float tu = __fdiv_rn((float)x + 0.5f, fw), tv = __fdiv_rn((float)y + 0.5f, fh);
    const f3    cur_n   = oct_decode(h2f_lo(c2x), h2f_hi(c2x));
    const f3    cur_pos = world_pos_from_depth(tu, tv, depth, M);
    const float htu = tu + h2f_lo(c2y), htv = tv + h2f_hi(c2y);
    const f3    hn = oct_decode(h2f_lo(q2x), h2f_hi(q2x));
    const f3    hp = world_pos_from_depth(htu, htv, d, M);
    return reprojection_valid(hcx, hcy, cur_pos, hp, cur_n, hn, cur_id, h2f_lo(q3y), w, h);
}

// virtual_point_reprojection (reprojection.glsl:71-111, reproject.h:115-129) with the parity kernels' operations: the history coordinate of a mirror-like
// pixel (ray length > 0 on a flat surface).  The tolerance-mode reflections temporal kernel used rsq / rcp here; the coordinate then differed from the
// oracle's by a few ulp, the bilinear weights of the four history taps by ~1e-4 — enough to move an interpolated fp16 MOMENT across a rounding boundary
// in one of ~10 such texels, and the variance m2 - m1^2 of two stored moments is what the a-trous chain is sensitive to (docs/EXPERIMENTS.md R5.8,
// tools/refl_atrous_conditioning.py).  With this coordinate the weights, and so the history colour and moments, are the parity kernels' wherever
// the history images are.
HR_DEV void virtual_point(const float* __restrict__ M, const float* __restrict__ prev_vp, f3 cam_pos, int x, int y, int w, int h, float depth, float ray_length,
                          float& rx, float& ry)
{
    const float fw = (float)w, fh = (float)h;
    const float vu = __fdiv_rn((float)x, fw), vv = __fdiv_rn((float)y, fh);   // NB without the half-pixel offset, as the reference
    const f3    ro = world_pos_from_depth(vu, vv, depth, M);
    f3          cr = sub3(ro, cam_pos);
    const float crl = len3(cr);
    cr = normalize3(cr);
    const f3 hp = add3(cam_pos, scale3(cr, crl + ray_length));
    const f4 rp = mul_m4(prev_vp, hp.x, hp.y, hp.z, 1.0f);
    const float px = __fdiv_rn(rp.x, rp.w), py = __fdiv_rn(rp.y, rp.w);
    rx = (px * 0.5f + 0.5f) * fw;
    ry = (py * 0.5f + 0.5f) * fh;
}

// the operands k_ddgi_sample (ddgi.hip) derives for pixel (x, y): world position, normal, direction to the camera
HR_DEV void pixel_inputs(const float* __restrict__ vpi, const float* __restrict__ cam, int x, int y, int w, int h, float depth, uint32_t g2x, f3& P, f3& N, f3& Wo)
{
    const float tu = __fdiv_rn((float)x + 0.5f, (float)w), tv = __fdiv_rn((float)y + 0.5f, (float)h);
    P  = world_pos_from_depth(tu, tv, depth, vpi);
    N  = oct_decode(h2f_lo(g2x), h2f_hi(g2x));
    Wo = normalize3(sub3(mk3(cam[0], cam[1], cam[2]), P));
}

} // namespace exact
} // namespace hr
