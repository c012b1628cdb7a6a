// Device-side fp32 arithmetic of the hot path (gfx950).
//
// The GLSL built-ins the reference shaders use are pinned to one IEEE-754 binary32 definition
// (DESIGN.md §3 "numerical contract") so that integer / bit outputs — the packed visibility
// masks of shadows_ray_trace.comp / ao_ray_trace.comp — are reproducible bit for bit:
//   * this translation unit is compiled with -ffp-contract=off: every + - * / sqrt below is one
//     correctly rounded operation (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt);
//     fused multiply-adds appear only where written explicitly (hr_fma, box tests);
//   * dot = (x*x' + y*y') + z*z',  normalize(v) = v * (1/sqrt(dot(v,v))),  mix = a*(1-t) + b*t;
//   * sin/cos/exp/log are fixed polynomial kernels (Cephes single-precision coefficients);
//   * fp16 stores round to nearest even (v_cvt_f16_f32), fp16 denormals kept.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HR_DEV __device__ __forceinline__

namespace hr {

struct f3 { float x, y, z; };

// correctly rounded sqrt (llvm.sqrt.f32 under -fhip-fp32-correctly-rounded-divide-sqrt).  NB: HIP's
// __fsqrt_rn() is the NATIVE (1 ulp) v_sqrt_f32 unless OCML_BASIC_ROUNDED_OPERATIONS is defined.
HR_DEV float hr_sqrt(float x) { return __builtin_sqrtf(x); }
HR_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
HR_DEV f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
HR_DEV f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
HR_DEV f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
HR_DEV f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }
HR_DEV float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
HR_DEV f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
HR_DEV float len3(f3 a) { return hr_sqrt(dot3(a, a)); }
HR_DEV f3 normalize3(f3 a)
{
    float inv = __fdiv_rn(1.0f, hr_sqrt(dot3(a, a)));
    return scale3(a, inv);
}
HR_DEV float min2(float a, float b) { return a < b ? a : b; }
HR_DEV float max2(float a, float b) { return a > b ? a : b; }
HR_DEV float clamp1(float x, float lo, float hi) { return min2(max2(x, lo), hi); }
HR_DEV float mix1(float a, float b, float t) { return a * (1.0f - t) + b * t; }
HR_DEV float fract1(float x) { return x - floorf(x); }
HR_DEV float step1(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
HR_DEV float smoothstep1(float e0, float e1, float x)
{
    float t = clamp1(__fdiv_rn(x - e0, e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
HR_DEV float hr_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// column-major mat4 * vec4, rows summed left to right
struct f4 { float x, y, z, w; };
HR_DEV f4 mul_m4(const float* __restrict__ M, float x, float y, float z, float w)
{
    f4 r;
    r.x = ((M[0] * x + M[4] * y) + M[8] * z) + M[12] * w;
    r.y = ((M[1] * x + M[5] * y) + M[9] * z) + M[13] * w;
    r.z = ((M[2] * x + M[6] * y) + M[10] * z) + M[14] * w;
    r.w = ((M[3] * x + M[7] * y) + M[11] * z) + M[15] * w;
    return r;
}

// ---- fp16 ------------------------------------------------------------------------------------
HR_DEV float    h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
// fp32 -> fp16, round to nearest even, as its own instruction.  A plain (_Float16) cast lets the AMDGPU back end fold a
// preceding fp32 multiply into v_fma_mixlo_f16 (ONE rounding of the exact product instead of fp32 then fp16) and pair two
// casts into v_cvt_pk_f16_f32 — found by tools/fuzz_gpu.py as 1-ulp fp16 differences in the DDGI sample pass whenever
// gi_intensity != 1.  The contract rounds every fp32 op first, as the CPU and the reference's SPIR-V do.
HR_DEV uint16_t f2h(float f)
{
    uint32_t r;
    asm("v_cvt_f16_f32 %0, %1" : "=v"(r) : "v"(f));
    return (uint16_t)r;
}
HR_DEV float    h2f_lo(uint32_t packed) { return h2f((uint16_t)(packed & 0xffffu)); }
HR_DEV float    h2f_hi(uint32_t packed) { return h2f((uint16_t)(packed >> 16)); }
HR_DEV uint32_t pack_h2(float a, float b) { return (uint32_t)f2h(a) | ((uint32_t)f2h(b) << 16); }

// ---- fixed transcendental kernels ---------------------------------------------------------------
HR_DEV void det_sincos(float x, float& s, float& c)
{
    float kf = floorf(x * 0.636619772367581f + 0.5f);
    int   k  = (int)kf;
    float r  = ((x - kf * 1.5703125f) - kf * 4.837512969970703125e-4f) - kf * 7.54978995489188e-8f;
    float z  = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    int   q  = k & 3;
    float sv = (q & 1) ? cp : sp;
    float cv = (q & 1) ? sp : cp;
    s = (q & 2) ? -sv : sv;
    c = (q == 1 || q == 2) ? -cv : cv;
}

HR_DEV float det_exp(float x)
{
    if (x > 88.0f) x = 88.0f;
    if (x < -87.0f) return 0.0f;   // keep the early return: in the a-trous kernels whole waves take it (tiny phi => huge -wL);
                                   // a branch-free version measured 18% slower there (24.7 vs 20.9 us per iteration)
    float nf = floorf(x * 1.44269504088896341f + 0.5f);
    int   n  = (int)nf;
    float r  = (x - nf * 0.693359375f) - nf * -2.12194440e-4f;
    float z  = r * r;
    float p  = (((((1.9875691500e-4f * r + 1.3981999507e-3f) * r + 8.3334519073e-3f) * r + 4.1665795894e-2f) * r + 1.6666665459e-1f) * r + 5.0000001201e-1f) * z + r + 1.0f;
    return p * __uint_as_float((uint32_t)(n + 127) << 23);
}

// Correctly rounded n / d for MANY numerators over ONE denominator: the denominator half of the IEEE sequence
// (v_rcp_f32 + one Newton step) is done once, each quotient then costs five FMAs — the very operations the compiler's
// expansion of `/` performs when v_div_scale does not rescale.  That holds for 1e-6 <= d <= 1e6 and n == 0 or
// 1e-12 <= |n| <= 3e5 (the callers' numerators are differences of fp16 values); anything else takes the generic path.
// Measured on the shadow a-trous kernel (8 taps per pixel): 25.3 -> 22.8 us per iteration, bit-identical images.
// (Evaluating the two exp() of a tap pair with packed v_pk_mul/add_f32 was tried as well: slower, 26.6 us.)
struct DivBy
{
    float d, r1;
    bool  fast;
};
HR_DEV DivBy div_prepare(float d)
{
    DivBy D;
    D.d = d;
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e0 = hr_fma(-d, r0, 1.0f);
    D.r1   = hr_fma(e0, r0, r0);
    D.fast = d >= 1e-6f && d <= 1e6f;
    return D;
}
HR_DEV float div_by(float n, const DivBy& D)
{
    const float an = fabsf(n);
    if (D.fast && (n == 0.0f || (an >= 1e-12f && an <= 3e5f)))
    {
        const float q0 = n * D.r1;
        const float e1 = hr_fma(-D.d, q0, n);
        const float q1 = hr_fma(e1, D.r1, q0);
        const float e2 = hr_fma(-D.d, q1, n);
        return hr_fma(e2, D.r1, q1);
    }
    return __fdiv_rn(n, D.d);
}
// div_by without its range tests: the CALLER guarantees D.fast and n == 0 or 1e-12 <= |n| <= 3e5 (a NaN comes back a NaN either way)
HR_DEV float div_by_inrange(float n, const DivBy& D)
{
    const float q0 = n * D.r1;
    const float e1 = hr_fma(-D.d, q0, n);
    const float q1 = hr_fma(e1, D.r1, q0);
    const float e2 = hr_fma(-D.d, q1, n);
    return hr_fma(e2, D.r1, q1);
}
// ... behind a flag the caller established once for all its numerators (wave-uniform where the denominator is): no per-quotient range branches
HR_DEV float div_by_if(bool inrange, float n, const DivBy& D) { return inrange ? div_by_inrange(n, D) : __fdiv_rn(n, D.d); }

HR_DEV float det_log(float x)
{
    if (x <= 0.0f) return -1.0e30f;
    uint32_t u = __float_as_uint(x);
    int      e = (int)(u >> 23) - 126;
    float    m = __uint_as_float((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f)
    {
        e -= 1;
        m = m + m - 1.0f;
    }
    else
        m = m - 1.0f;
    float z  = m * m;
    float y  = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m + 1.4249322787e-1f) * m - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m + 3.3333331174e-1f) * m * z;
    float fe = (float)e;
    y        = y + fe * -2.12194440e-4f;
    y        = y - 0.5f * z;
    float r  = m + y;
    r        = r + fe * 0.693359375f;
    return r;
}

HR_DEV float det_pow(float x, float y)
{
    if (x <= 0.0f) return 0.0f;
    return det_exp(y * det_log(x));
}

HR_DEV float det_powi(float x, int n)
{
    float r = 1.0f, b = x;
    while (n > 0)
    {
        if (n & 1) r = r * b;
        b = b * b;
        n >>= 1;
    }
    return r;
}

// integer exponents 1..64 by squaring, everything else exp(y*log(x))
HR_DEV float det_pow_auto(float x, float p)
{
    float pf = floorf(p);
    if (pf == p && p >= 1.0f && p <= 64.0f) return det_powi(x, (int)p);
    return det_pow(x, p);
}

// ---- common.glsl ------------------------------------------------------------------------------
#define HR_M_PI 3.14159265359f

// common.glsl:150-156 octohedral_to_direction
HR_DEV f3 oct_decode(float ex, float ey)
{
    f3 v = mk3(ex, ey, 1.0f - fabsf(ex) - fabsf(ey));
    if (v.z < 0.0f)
    {
        float nx = (1.0f - fabsf(v.y)) * (step1(0.0f, v.x) * 2.0f - 1.0f);
        float ny = (1.0f - fabsf(v.x)) * (step1(0.0f, v.y) * 2.0f - 1.0f);
        v.x = nx;
        v.y = ny;
    }
    return normalize3(v);
}

// g_buffer.frag:47-51 direction_to_octohedral
HR_DEV void oct_encode(f3 n, float& ox, float& oy)
{
    float inv = __fdiv_rn(1.0f, (fabsf(n.x) + fabsf(n.y)) + fabsf(n.z));
    float px = n.x * inv, py = n.y * inv;
    if (n.z > 0.0f) { ox = px; oy = py; }
    else
    {
        ox = (1.0f - fabsf(py)) * (step1(0.0f, px) * 2.0f - 1.0f);
        oy = (1.0f - fabsf(px)) * (step1(0.0f, py) * 2.0f - 1.0f);
    }
}

// common.glsl:169-184 world_position_from_depth
// Threads of a ray-trace dispatch.  The reference launches ceil(w/8) x ceil(h/4) groups of 8x4 threads WITHOUT a bounds
// check (shadows_ray_trace.comp:89-132, ao_ray_trace.comp:90-126): a thread past the right / bottom edge of a ragged
// image fetches depth 0 and normal (0,0) (pinned rule: texel fetches outside an image return 0), so it traces a ray
// and contributes its mask bit — and the 17x17 neighbourhood mean of the denoiser reads those bits.
//   0: no such thread (past the last 8x4 group, or rows another band owns)   1: image pixel   2: edge thread
HR_DEV int trace_lane_kind(int x, int y, int w, int h, int y0, int y1)
{
    const int gy = (y >> 2) << 2; // first row of the thread's 8x4 group
    if (gy >= h) return 0;
    if (y < h) return (y < y0 || y >= y1) ? 0 : (x < w ? 1 : 2);
    return (gy >= y0 && gy < y1) ? 2 : 0;
}

HR_DEV f3 world_pos_from_depth(float u, float v, float ndc_depth, const float* __restrict__ view_proj_inverse)
{
    f4 wp = mul_m4(view_proj_inverse, u * 2.0f - 1.0f, v * 2.0f - 1.0f, ndc_depth, 1.0f);
    return mk3(__fdiv_rn(wp.x, wp.w), __fdiv_rn(wp.y, wp.w), __fdiv_rn(wp.z, wp.w));
}

// common.glsl:141-144
HR_DEV float luminance(f3 rgb) { return max2(dot3(rgb, mk3(0.299f, 0.587f, 0.114f)), 0.0001f); }

// bnd_sampler.glsl:4-24.  int(clamp(unorm8 * 256, 0, 255)) is the identity on 0..255
// (tests/test_oracle_kat.py::test_unorm8_identity), so the byte is used directly.
// the scrambling / ranking texel of a pixel depends on (x, y) only: kernels fetch it up front, next to the G-buffer texels
HR_DEV uint32_t blue_noise_texel(int cx, int cy, const uint8_t* __restrict__ sr) { return *(const uint32_t*)(sr + (((cy & 127) * 128 + (cx & 127)) << 2)); }
HR_DEV float sample_blue_noise_t(uint32_t t, int sample_index, int dim, const uint8_t* __restrict__ sobol)
{
    sample_index &= 255;
    dim &= 3;
    int ranked = sample_index ^ (int)((t >> 16) & 0xffu);
    int value  = (int)sobol[ranked * 4 + dim];
    value ^= (int)((t >> ((dim & 1) * 8)) & 0xffu);
    return (0.5f + (float)value) * 0.00390625f;   // / 256: a power of two, so the product is the correctly rounded quotient
}
HR_DEV float sample_blue_noise(int cx, int cy, int sample_index, int dim, const uint8_t* __restrict__ sobol, const uint8_t* __restrict__ sr)
{
    return sample_blue_noise_t(blue_noise_texel(cx, cy, sr), sample_index, dim, sobol);
}

} // namespace hr
