// Edge-aware 4-tap bilateral upsample shared by the three passes — device restatement of
// shadows_upsample.comp:62-109, ao_upsample.comp:63-112, reflections_upsample.comp:62-109.
// All pass samplers are nearest (g_buffer.cpp:328-340, ray_traced_shadows.cpp:363...), so
// textureLod(uv) is a point fetch at floor(uv * size) clamped to the edge.
#pragma once
#include "device_math.h"

namespace hr {

struct UpsampleArgs
{
    int          W, H, w, h;       // full-res and low-res extents
    const uint2* G2;               // full-res GB2 / GB3 (RGBA16F)
    const uint2* G3;
    const uint2* g2;               // low-res (mip) GB2 / GB3
    const uint2* g3;
    const void*  in;               // low-res input, fp16, in_channels per texel
    int          in_channels;
    int          channels;         // 1 (shadows, AO) or 4 (reflections)
    void*        out;              // full-res fp16, `channels` per texel
    float        sky_value;        // 0 (shadows / reflections), 1 (AO)
    float        power;            // 0 = none, AO: 1.2
};

template <int CH>
__global__ __launch_bounds__(256) void k_upsample(UpsampleArgs a)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= a.W || y >= a.H) return;
    const size_t o  = (size_t)y * a.W + x;
    uint16_t*    op = (uint16_t*)a.out + o * CH;
    const uint2  G3 = a.G3[o];
    const float  hi_depth = h2f_hi(G3.y);
    if (hi_depth == -1.0f)
    {
        const uint16_t sv = f2h(a.sky_value);
#pragma unroll
        for (int c = 0; c < CH; c++) op[c] = sv;
        return;
    }
    const uint2 G2 = a.G2[o];
    const f3    hn = oct_decode(h2f_lo(G2.x), h2f_hi(G2.x));
    const float tu = __fdiv_rn((float)x + 0.5f, (float)a.W), tv = __fdiv_rn((float)y + 0.5f, (float)a.H);
    const float tsx = __fdiv_rn(1.0f, (float)a.w), tsy = __fdiv_rn(1.0f, (float)a.h);
    float up[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) up[c] = 0.0f;
    float total_w = 0.0f;
    // all twelve texels of the four low-resolution taps first (clamped addresses: always valid), then the weights
    uint2    t3[4], t2[4];
    uint16_t tin[4][CH];
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const float kx = (i == 1) ? 1.0f : (i == 2 ? -1.0f : 0.0f), ky = (i == 0) ? 1.0f : (i == 3 ? -1.0f : 0.0f);
        const float cu = tu + kx * tsx, cv = tv + ky * tsy;
        int sx = (int)floorf(cu * (float)a.w), sy = (int)floorf(cv * (float)a.h);
        sx = sx < 0 ? 0 : (sx > a.w - 1 ? a.w - 1 : sx);
        sy = sy < 0 ? 0 : (sy > a.h - 1 ? a.h - 1 : sy);
        const size_t so = (size_t)sy * a.w + sx;
        t3[i] = a.g3[so]; t2[i] = a.g2[so];
        const uint16_t* ip = (const uint16_t*)a.in + so * a.in_channels;
#pragma unroll
        for (int c = 0; c < CH; c++) tin[i][c] = ip[c];
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const float cd = h2f_hi(t3[i].y);
        if (cd == -1.0f) continue;
        const f3    cn = oct_decode(h2f_lo(t2[i].x), h2f_hi(t2[i].x));
        // compute_edge_stopping_weight, NORMAL weight only: wL = 1.0 (edge_stopping.glsl:53-59)
        const float wZ = det_exp(__fdiv_rn(-fabsf(hi_depth - cd), 1.0f));
        const float wN = det_pow_auto(clamp1(dot3(hn, cn), 0.0f, 1.0f), 32.0f);
        const float wt = det_exp((0.0f - 1.0f) - max2(wZ, 0.0f)) * wN;
#pragma unroll
        for (int c = 0; c < CH; c++) up[c] += h2f(tin[i][c]) * wt;
        total_w += wt;
    }
#pragma unroll
    for (int c = 0; c < CH; c++)
    {
        float r = __fdiv_rn(up[c], max2(total_w, 0.00000001f));
        if (a.power != 0.0f) r = det_pow_auto(r, a.power);
        op[c] = f2h(r);
    }
}

inline void launch_upsample(const UpsampleArgs& a, hipStream_t st)
{
    dim3 grid((a.W + 31) / 32, (a.H + 7) / 8), block(256);
    if (a.channels == 4) hipLaunchKernelGGL(k_upsample<4>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(k_upsample<1>, grid, block, 0, st, a);
}

} // namespace hr
