// Tolerance-mode arithmetic (hr_*_params.exact == 0) — include ONLY from denoise_fast.hip.
//
// The exact kernels pin every GLSL built-in to one correctly rounded fp32 definition so that all stage images can be compared
// bit for bit with the oracle.  That contract costs 10-20 VALU instructions per division / square root / exp
// and forbids fused multiply-adds; it made every denoise kernel VALU-bound at 10-35% of the HBM roofline.
// north_star asks for bit-exactness only of the visibility masks and a stated fp32 tolerance for the fp16 images, so the
// shipping mode computes with the hardware's transcendental unit:
//   x / y      -> x * v_rcp_f32(y)           (1 ulp)          sqrt -> v_sqrt_f32 (1 ulp), 1/sqrt -> v_rsq_f32 (1 ulp)
//   exp(x)     -> v_exp_f32(x * log2 e)      (1 ulp)          pow(x, p) -> v_exp_f32(p * v_log_f32(x))
//   a * b + c  -> v_fma_f32 (the file is compiled with fp-contract fast), sums re-associated where it saves work
// Accuracy: a few fp32 ulp per stage — three orders of magnitude below the fp16 ulp (2^-11) the images are stored with;
// tests/test_gpu_tolerance.py holds the stated bound (rel-L2 <= 1e-3 per image, <= 2 fp16 ulp on >= 99.9 % of the texels —
// the rest are discrete decisions, e.g. a reprojection validity test on a knife edge, flipped by an fp32 ulp).
#pragma once
#include "device_math.h"

namespace hr {
namespace fm {

HR_DEV float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
HR_DEV float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
HR_DEV float sqrt1(float x) { return __builtin_amdgcn_sqrtf(x); }
HR_DEV float exp2f_(float x) { return __builtin_amdgcn_exp2f(x); }
HR_DEV float log2f_(float x) { return __builtin_amdgcn_logf(x); }
HR_DEV float expf_(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
// pow: 0 for x <= 0, as the exact mode's det_pow (the reference's pow() is only ever fed non-negative bases)
HR_DEV float powf_(float x, float p) { return x > 0.0f ? __builtin_amdgcn_exp2f(p * __builtin_amdgcn_logf(x)) : 0.0f; }
HR_DEV float pow32(float x) { float b = x * x; b = b * b; b = b * b; b = b * b; return b * b; }
HR_DEV float sat(float x) { return __builtin_fminf(__builtin_fmaxf(x, 0.0f), 1.0f); }   // folds into the clamp output modifier
HR_DEV float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
HR_DEV float fmin_(float a, float b) { return __builtin_fminf(a, b); }
// GLSL mix = a * (1 - t) + b * t, NOT a + t * (b - a): the former returns b exactly at t == 1 (first frames, disocclusions:
// alpha = 1), which the tile classification `ao < 1.0` depends on
HR_DEV float mix(float a, float b, float t) { return __builtin_fmaf(b, t, a * (1.0f - t)); }
// a * b + c with TWO roundings (this header is compiled with fp-contract off; only __builtin_fmaf fuses): for denominators the
// oracle forms that way and whose reciprocals are then subtracted from one another
HR_DEV float mad_rn(float a, float b, float c) { return a * b + c; }
// a * b that never fuses with a following add (a tap weight that is summed AND multiplied further: the fused and unfused forms of
// a filter must round it alike to stay bit-identical to each other)
HR_DEV float mul_rn(float a, float b) { return a * b; }
// reciprocal refined by one Newton step (~0.5 ulp): for quantities whose DIFFERENCES are used (linear eye depth of neighbouring
// texels cancels 3-4 digits in the bilateral depth weight)
HR_DEV float rcp_nr(float x) { const float r = __builtin_amdgcn_rcpf(x); return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r); }
HR_DEV float dot(f3 a, f3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }

// fp16 pair -> packed word, round to nearest even (the back end may fuse a preceding multiply: allowed here)
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef float    f2_t __attribute__((ext_vector_type(2)));
HR_DEV uint32_t pack2(float a, float b)
{
    f2_t v = { a, b };
    h2_t h = __builtin_convertvector(v, h2_t);
    return __builtin_bit_cast(uint32_t, h);
}
HR_DEV float lo(uint32_t p) { return (float)__builtin_bit_cast(h2_t, p).x; }
HR_DEV float hi(uint32_t p) { return (float)__builtin_bit_cast(h2_t, p).y; }
HR_DEV uint16_t half_bits(float a) { return __builtin_bit_cast(uint16_t, (_Float16)a); }

// octahedral decode WITHOUT the normalisation (common.glsl:150-156): callers that only need the direction of the normal
// (a cosine against another unit vector) divide by the length once, where it is cheapest
HR_DEV f3 oct_raw(uint32_t packed_xy)
{
    const float ex = lo(packed_xy), ey = hi(packed_xy);
    const float z  = 1.0f - __builtin_fabsf(ex) - __builtin_fabsf(ey);
    const float t  = __builtin_fmaxf(-z, 0.0f);
    return mk3(ex + (ex >= 0.0f ? -t : t), ey + (ey >= 0.0f ? -t : t), z);
}
HR_DEV f3 oct_unit(uint32_t packed_xy)
{
    const f3    v = oct_raw(packed_xy);
    const float s = rsq(dot(v, v));
    return mk3(v.x * s, v.y * s, v.z * s);
}

// load through a uniform base + a 32-bit byte offset (global_load ... saddr): no 64-bit address arithmetic per tap
template <typename T>
HR_DEV T ld(const void* __restrict__ base, uint32_t byte_off) { return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off); }

// world_position_from_depth (common.glsl:169-184), split so that taps sharing (u, v) share the matrix work.
// The clip -> world product is ILL-CONDITIONED: w = M[11] * d + M[15] cancels three to four digits for depths near 1, so two
// different roundings of the same product move a world position by ~1e-4 of its distance — enough to flip the 5-unit plane
// distance test of the reprojection at silhouettes a few times per frame (measured).  The product therefore keeps the oracle's
// operation order ((m0 x + m1 y) + m2 d) + m3, each operation rounded (this header is compiled with fp-contract off); that order
// happens to share (m0 x + m1 y) between all taps of a pixel.  Only the perspective divide is fast (refined reciprocal).
struct Unproj { float tx, ty, tz, tw; };
HR_DEV Unproj unproject_base(const float* __restrict__ M, float u, float v)
{
    const float x = u * 2.0f - 1.0f, y = v * 2.0f - 1.0f;
    Unproj b;
    b.tx = M[0] * x + M[4] * y;
    b.ty = M[1] * x + M[5] * y;
    b.tz = M[2] * x + M[6] * y;
    b.tw = M[3] * x + M[7] * y;
    return b;
}
HR_DEV f3 unproject_at(const Unproj& b, const float* __restrict__ M, float d)
{
    const float wx = (b.tx + M[8] * d) + M[12], wy = (b.ty + M[9] * d) + M[13], wz = (b.tz + M[10] * d) + M[14], ww = (b.tw + M[11] * d) + M[15];
    const float inv = rcp_nr(ww);
    return mk3(wx * inv, wy * inv, wz * inv);
}
// GLSL mix with every operation rounded, and m2 - m1^2 likewise: the last steps of the temporal accumulation.  Their results are
// compared against thresholds (ao < 1, visibility > 0 => tile class) and differenced (variance), so they follow the oracle's order
HR_DEV float mix_rn(float a, float b, float t) { return a * (1.0f - t) + b * t; }
HR_DEV float var_rn(float m2, float m1) { return m2 - m1 * m1; }

// Texel coordinates (minus the half texel) of an octahedral direction (ox, oy in [-1, 1]) inside atlas cell (col, row) of probes
// `side` texels wide, in the oracle's operation order (gi_common.glsl texture_coord_from_direction + the bilinear set-up of
// shading.h): u = tl / tw + (z * side) / tw, x = u * tw - 0.5.  The two divisions are correctly rounded through the shared
// denominators Dw / Dh.  Needed for the DEPTH atlas only: its Chebyshev term |mean^2 - m2| cancels three digits (fp16 moments),
// so 1e-5 texels of disagreement in the bilinear fractions show up as 1e-2 in a probe's weight.
using hr::div_by_inrange;   // device_math.h: div_by without its range tests — the callers here guarantee 1e-6 <= d <= 1e6 and n == 0 or 1e-12 <= |n| <= 3e5
HR_DEV void atlas_coord_rn(float ox, float oy, int col, int row, int side, float tw, float th, const DivBy& Dw, const DivBy& Dh, float& x, float& y)
{
    const float zx = (ox + 1.0f) * 0.5f, zy = (oy + 1.0f) * 0.5f;
    const float pwb = (float)side + 2.0f;
    // numerators: cell origins 2 .. atlas size, z * side in {0} u [3e-8, 16]; denominators: atlas extents — all inside div_by's fast range
    const float u = div_by_inrange((float)col * pwb + 2.0f, Dw) + div_by_inrange(zx * (float)side, Dw);
    const float v = div_by_inrange((float)row * pwb + 2.0f, Dh) + div_by_inrange(zy * (float)side, Dh);
    x = u * tw - 0.5f;
    y = v * th - 0.5f;
}
HR_DEV float bilerp_rn(float t00, float t10, float t01, float t11, float fx, float fy) { return mix_rn(mix_rn(t00, t10, fx), mix_rn(t01, t11, fx), fy); }
HR_DEV float cheb_variance_rn(float mean, float m2) { return __builtin_fabsf(mean * mean - m2); }

// Texel addressing of the upsample kernels (*_upsample.comp:70-84): uv = (pixel + 0.5) / size and the tap's texel
// floor((uv + k / size_low) * size_low) are DISCRETE decisions that hit integers exactly for whole columns / rows of an odd-sized
// image ((96 + 0.5) / 193 = 0.5): they keep the reference's operations — correctly rounded divisions, separate multiply and add.
HR_DEV float div_rn(float n, float d) { return __fdiv_rn(n, d); }
HR_DEV int   tap_texel(float uv, float k, float texel_size, float extent) { return (int)__builtin_floorf((uv + k * texel_size) * extent); }

} // namespace fm
} // namespace hr
