// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).  Pinned against the reference's own shaders: tests/test_ref_shaders.py.
// Data contract shared by the restated shaders: UBO / Light structs, image views with the
// pinned out-of-bounds rule (texelFetch outside the image returns 0 — SURVEY.md §8a quirk 3),
// the blue-noise sampler and the scene handle.
#pragma once
#include "orc_math.h"
#include <vector>

namespace orc {

// common.h:106-158 / common.glsl:75-139
struct Light
{
    float data0[4]; // xyz = direction TO the light, w = intensity
    float data1[4]; // xyz = position, w = radius
    float data2[4]; // xyz = colour
    float data3[4]; // x = type (0 dir, 1 point, 2 spot), y = cos outer, z = cos inner
};

// common.h:161-179 (416 bytes)
struct UBO
{
    mat4  view_inverse;
    mat4  proj_inverse;
    mat4  view_proj_inverse;
    mat4  prev_view_proj;
    mat4  view_proj;
    float cam_pos[4];
    float current_prev_jitter[4];
    Light light;
};
static_assert(sizeof(UBO) == 416, "UBO layout");

// ---- image views: tightly packed row-major [h][w][C] --------------------------------
template <int C>
struct ImgH // fp16 channels
{
    const uint16_t* p;
    int             w, h;
    inline bool inside(int x, int y) const { return x >= 0 && y >= 0 && x < w && y < h; }
    inline float fetch(int x, int y, int c) const
    {
        if (!inside(x, y)) return 0.0f;
        return f16_to_f32(p[((size_t)y * w + x) * C + c]);
    }
};
template <int C>
struct ImgHW
{
    uint16_t* p;
    int       w, h;
    inline void store(int x, int y, int c, float v)
    {
        if (x < 0 || y < 0 || x >= w || y >= h) return;
        p[((size_t)y * w + x) * C + c] = f32_to_f16(v);
    }
};
struct ImgF
{
    const float* p;
    int          w, h;
    inline float fetch(int x, int y) const
    {
        if (x < 0 || y < 0 || x >= w || y >= h) return 0.0f;
        return p[(size_t)y * w + x];
    }
};
struct ImgU
{
    const uint32_t* p;
    int             w, h;
    inline uint32_t fetch(int x, int y, uint32_t oob = 0u) const
    {
        if (x < 0 || y < 0 || x >= w || y >= h) return oob;
        return p[(size_t)y * w + x];
    }
};

// bnd_sampler.glsl:4-24.  sobol: 256x1 RGBA8, scrambling_ranking: 128x128 RGBA8.
// texelFetch of an UNORM8 texel returns b/255; the shader multiplies by 256 and clamps
// to [0,255] before the int() truncation:  int(clamp(b/255*256, 0, 255)).
struct BlueNoise
{
    const uint8_t* sobol;              // [256][4]
    const uint8_t* scrambling_ranking; // [128][128][4]
};
static inline int unorm8_to_int256(uint8_t b)
{
    float v = ((float)b / 255.0f) * 256.0f;
    v       = clampf(v, 0.0f, 255.0f);
    return (int)v;
}
static inline float sample_blue_noise(int cx, int cy, int sample_index, int sample_dimension, const BlueNoise& bn)
{
    cx               = cx % 128;
    cy               = cy % 128;
    sample_index     = sample_index % 256;
    sample_dimension = sample_dimension % 4;
    const uint8_t* sr = bn.scrambling_ranking + ((size_t)cy * 128 + cx) * 4;
    int ranked = sample_index ^ unorm8_to_int256(sr[2]);
    int value  = unorm8_to_int256(bn.sobol[(size_t)ranked * 4 + sample_dimension]);
    value      = value ^ unorm8_to_int256(sr[sample_dimension % 2]);
    return (0.5f + (float)value) / 256.0f;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

} // namespace orc
