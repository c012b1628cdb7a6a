// ORACLE — TEST INFRASTRUCTURE ONLY.
// CPU restatement of the GLSL arithmetic used by the reference's ray-trace + denoise
// shaders (/root/reference/src/shaders).  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this; the product (hybrid_rendering_amd/) never does.
//
// PARITY STATUS: pinned.  The reference ships no tests / golden vectors and cannot be built here (Vulkan), so the pin is
// the reference's OWN SHADER SOURCES executed on the CPU: oracle/refshim translates them (read where they lie under
// /root/reference/src/shaders, output only into oracle/_ref/) into C++ over a GLSL runtime, and tests/test_ref_shaders.py
// requires every stage of this restatement — and every committed golden fixture — to equal them bit for bit.
// What stays a CONTRACT rather than reference behaviour is exactly what GLSL / Vulkan leave implementation-defined:
// the fp32 built-ins below, sampler filtering, and the BVH traversal (the Vulkan driver).  The shim
// binds the shaders' built-ins to the definitions in this file; the HIP kernels implement them independently.
//
// Numerical contract:
//   * every fp32 op is an individually rounded IEEE-754 binary32 op (compile with
//     -ffp-contract=off, no fast-math); + - * / sqrt are correctly rounded.
//   * dot(a,b)        = (a.x*b.x + a.y*b.y) + a.z*b.z           (left to right)
//   * normalize(v)    = v * (1.0f / sqrt(dot(v,v)))
//   * mix(a,b,t)      = a*(1-t) + b*t                            (GLSL spec form)
//   * mat4*vec4       = ((m0*x + m1*y) + m2*z) + m3*w per row, column-major storage
//   * sin/cos/exp/log = the polynomial kernels below (Cephes single-precision
//     coefficients), pow(x,y) = exp(y*log(x)), integer powers by repeated squaring
//     where the shader passes a literal integer exponent.
//   * fp16 stores round to nearest even; fp16 subnormals are kept.
#pragma once
#include <cstdint>
#include <cstring>
#include <cmath>

namespace orc {

// ----------------------------------------------------------------------------- bits
static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float    u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// fp32 -> fp16 bits, round-to-nearest-even, subnormals preserved.
static inline uint16_t f32_to_f16(float f)
{
    uint32_t x    = f2u(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));
    uint32_t e = x >> 23;
    if (e >= 143) return (uint16_t)(sign | 0x7c00u);
    if (e >= 113)
    {
        uint32_t m   = x & 0x7fffffu;
        uint32_t h   = ((e - 112) << 10) | (m >> 13);
        uint32_t rem = m & 0x1fffu;
        if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
        return (uint16_t)(sign | h);
    }
    if (e < 102) return (uint16_t)sign;
    uint32_t m     = (x & 0x7fffffu) | 0x800000u;
    uint32_t shift = 126 - e;
    uint32_t h     = m >> shift;
    uint32_t rem   = m & ((1u << shift) - 1u);
    uint32_t half  = 1u << (shift - 1);
    if (rem > half || (rem == half && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}

static inline float f16_to_f32(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e    = (h >> 10) & 0x1fu;
    uint32_t m    = h & 0x3ffu;
    if (e == 0)
    {
        if (m == 0) return u2f(sign);
        // subnormal: m * 2^-24
        float v = (float)m * 5.9604644775390625e-8f;
        return u2f(f2u(v) | sign);
    }
    if (e == 31) return u2f(sign | 0x7f800000u | (m << 13));
    return u2f(sign | ((e + 112) << 23) | (m << 13));
}

// ----------------------------------------------------------------------------- vectors
struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };
struct ivec2 { int x, y; };
struct mat4 { float m[16]; }; // column-major: m[c*4 + r]

static inline vec3 v3(float x, float y, float z) { return vec3 { x, y, z }; }
static inline vec3 operator+(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 operator-(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 operator*(vec3 a, vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline vec3 operator*(vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline vec3 operator*(float s, vec3 a) { return v3(a.x * s, a.y * s, a.z * s); }
static inline vec3 operator/(vec3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
static inline vec3 operator-(vec3 a) { return v3(-a.x, -a.y, -a.z); }
static inline float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline vec3  cross(vec3 a, vec3 b)
{
    return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
static inline vec3  normalize(vec3 a)
{
    float inv = 1.0f / std::sqrt(dot(a, a));
    return a * inv;
}
static inline float fmin2(float a, float b) { return a < b ? a : b; }
static inline float fmax2(float a, float b) { return a > b ? a : b; }
static inline float clampf(float x, float lo, float hi) { return fmin2(fmax2(x, lo), hi); }
static inline float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
static inline vec3  mix3(vec3 a, vec3 b, float t) { return a * (1.0f - t) + b * t; }
static inline float fractf(float x) { return x - std::floor(x); }
static inline float stepf(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
static inline float smoothstepf(float e0, float e1, float x)
{
    float t = clampf((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}

static inline vec4 mul(const mat4& M, vec4 v)
{
    vec4 r;
    r.x = ((M.m[0] * v.x + M.m[4] * v.y) + M.m[8] * v.z) + M.m[12] * v.w;
    r.y = ((M.m[1] * v.x + M.m[5] * v.y) + M.m[9] * v.z) + M.m[13] * v.w;
    r.z = ((M.m[2] * v.x + M.m[6] * v.y) + M.m[10] * v.z) + M.m[14] * v.w;
    r.w = ((M.m[3] * v.x + M.m[7] * v.y) + M.m[11] * v.z) + M.m[15] * v.w;
    return r;
}

// ----------------------------------------------------------------------------- detmath
// Cephes single-precision kernels, every op individually rounded.
static inline void det_sincos(float x, float* s_out, float* c_out)
{
    // quadrant reduction: k = round(x * 2/pi)
    float kf = std::floor(x * 0.636619772367581f + 0.5f);
    int   k  = (int)kf;
    // Cody-Waite with pi/2 split in three parts
    float r = ((x - kf * 1.5703125f) - kf * 4.837512969970703125e-4f) - kf * 7.54978995489188e-8f;
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    float s, c;
    switch (k & 3)
    {
        case 0: s = sp; c = cp; break;
        case 1: s = cp; c = -sp; break;
        case 2: s = -sp; c = -cp; break;
        default: s = -cp; c = sp; break;
    }
    *s_out = s;
    *c_out = c;
}

static inline float det_exp(float x)
{
    if (x > 88.0f) x = 88.0f;
    if (x < -87.0f) return 0.0f;
    float nf = std::floor(x * 1.44269504088896341f + 0.5f);
    int   n  = (int)nf;
    float r  = (x - nf * 0.693359375f) - nf * -2.12194440e-4f;
    float z  = r * r;
    float p  = (((((1.9875691500e-4f * r + 1.3981999507e-3f) * r + 8.3334519073e-3f) * r + 4.1665795894e-2f) * r + 1.6666665459e-1f) * r + 5.0000001201e-1f) * z + r + 1.0f;
    return p * u2f((uint32_t)(n + 127) << 23);
}

static inline float det_log(float x)
{
    // x > 0, normal.  Returns natural log.
    if (x <= 0.0f) return -1.0e30f;
    uint32_t u = f2u(x);
    int      e = (int)(u >> 23) - 126;             // x = m * 2^e, m in [0.5,1)
    float    m = u2f((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f)
    {
        e -= 1;
        m = m + m - 1.0f;
    }
    else
        m = m - 1.0f;
    float z = m * m;
    float y = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m + 1.4249322787e-1f) * m - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m + 3.3333331174e-1f) * m * z;
    float fe = (float)e;
    y = y + fe * -2.12194440e-4f;
    y = y - 0.5f * z;
    float r = m + y;
    r = r + fe * 0.693359375f;
    return r;
}

static inline float det_pow(float x, float y)
{
    if (x <= 0.0f) return 0.0f;
    return det_exp(y * det_log(x));
}

// pow with a small positive integer exponent: repeated squaring, MSB first.
static inline float det_powi(float x, int n)
{
    float r = 1.0f;
    float b = x;
    while (n > 0)
    {
        if (n & 1) r = r * b;
        b = b * b;
        n >>= 1;
    }
    return r;
}

// pow(x, p) as the shaders use it for edge-stopping / gamma: integer exponents
// (1..64) use det_powi, everything else det_pow.
static inline float det_pow_auto(float x, float p)
{
    float pf = std::floor(p);
    if (pf == p && p >= 1.0f && p <= 64.0f) return det_powi(x, (int)p);
    return det_pow(x, p);
}

// ----------------------------------------------------------------------------- common.glsl
#define ORC_M_PI 3.14159265359f

// common.glsl:150-156
static inline vec3 octohedral_to_direction(float ex, float ey)
{
    vec3 v = v3(ex, ey, 1.0f - std::fabs(ex) - std::fabs(ey));
    if (v.z < 0.0f)
    {
        float nx = (1.0f - std::fabs(v.y)) * (stepf(0.0f, v.x) * 2.0f - 1.0f);
        float ny = (1.0f - std::fabs(v.x)) * (stepf(0.0f, v.y) * 2.0f - 1.0f);
        v.x      = nx;
        v.y      = ny;
    }
    return normalize(v);
}

// g_buffer.frag:47-51
static inline vec2 direction_to_octohedral(vec3 n)
{
    float inv = 1.0f / ((std::fabs(n.x) + std::fabs(n.y)) + std::fabs(n.z));
    float px = n.x * inv, py = n.y * inv;
    if (n.z > 0.0f) return vec2 { px, py };
    return vec2 { (1.0f - std::fabs(py)) * (stepf(0.0f, px) * 2.0f - 1.0f), (1.0f - std::fabs(px)) * (stepf(0.0f, py) * 2.0f - 1.0f) };
}

// common.glsl:169-184
static inline vec3 world_position_from_depth(float u, float v, float ndc_depth, const mat4& view_proj_inverse)
{
    vec4 ndc = vec4 { u * 2.0f - 1.0f, v * 2.0f - 1.0f, ndc_depth, 1.0f };
    vec4 wp  = mul(view_proj_inverse, ndc);
    return v3(wp.x / wp.w, wp.y / wp.w, wp.z / wp.w);
}

// common.glsl:141-144
static inline float luminance(vec3 rgb)
{
    return fmax2(dot(rgb, v3(0.299f, 0.587f, 0.114f)), 0.0001f);
}

// common.glsl:160-165
static inline float gaussian_weight(float offset, float deviation)
{
    float weight = 1.0f / std::sqrt(2.0f * ORC_M_PI * deviation * deviation);
    weight       = weight * det_exp(-(offset * offset) / (2.0f * deviation * deviation));
    return weight;
}

// common.glsl:188-191
static inline float linear_eye_depth(float z, const float zbp[4])
{
    return 1.0f / (zbp[2] * z + zbp[3]);
}

} // namespace orc
