// 17x17 box sums over the packed 8x4 visibility masks of the trace kernels — integer work shared by both arithmetic modes
// (k_ao_temporal in ao.hip, kf_shadows_temporal / kf_ao_temporal in denoise_fast.hip): the counts are exact either way.
#pragma once
#include "device_math.h"

namespace hr {

// 17x17 box sum over packed 8x4 visibility masks (shadows_denoise_reprojection.comp:157-190, ao_...:152-185), 1..4 sample planes.
// Lanes 0..23 of a wave assemble the 24-bit row patterns of the 24 cached pixel rows (3 mask columns x 6 mask rows); with
// several planes the per-pixel sample count (0..4) is kept BIT-SLICED (three 24-bit words per row: bits 0, 1, 2 of the count),
// so the window sum costs three bfe + bcnt pairs per row whatever the sample count.
struct MaskRows { uint32_t b0[24], b1[24], b2[24]; };

template <bool AO>
HR_DEV void build_mask_rows(MaskRows& R, uint32_t (*s_mask)[18], const uint32_t* __restrict__ mask, int spp, int mw, int mh, int tx, int ty, int y0, int y1, int lane, bool tile_ok)
{
    // populate_cache: 3x6 mask words per plane; outside the mask image shadows read 0, AO reads all-ones (the shaders' guards)
    for (int s = 0; s < spp; s++)
        if (lane < 18)
        {
            const int cx = tx - 1 + lane % 3, cy = ty * 2 - 2 + lane / 3;
            uint32_t  v  = AO ? 0xFFFFFFFFu : 0u;
            const bool in = tile_ok && cx >= 0 && cy >= 0 && cx < mw && cy < mh && (AO || (cy * 4 >= y0 - 8 && cy * 4 < y1 + 8));
            if (in) v = mask[((size_t)s * mh + cy) * mw + cx];
            s_mask[s][lane] = v;
        }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the LDS stores above are visible to the wave
    if (lane < 24)
    {
        const int m = lane >> 2, br = (lane & 3) * 8;
        uint32_t  p[4] = { 0u, 0u, 0u, 0u };
        for (int s = 0; s < spp; s++)
            p[s] = ((s_mask[s][m * 3 + 0] >> br) & 0xffu) | (((s_mask[s][m * 3 + 1] >> br) & 0xffu) << 8) | (((s_mask[s][m * 3 + 2] >> br) & 0xffu) << 16);
        // count = p0 + p1 + p2 + p3 per bit position, bit-sliced
        const uint32_t s0 = p[0] ^ p[1], c0 = p[0] & p[1], s1 = p[2] ^ p[3], c1 = p[2] & p[3];
        const uint32_t carry = s0 & s1;
        R.b0[lane] = s0 ^ s1;
        R.b1[lane] = carry | (c0 ^ c1);   // carry excludes c0 and c1 (carry => p0 != p1 and p2 != p3)
        R.b2[lane] = c0 & c1;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
}

// window sum of lane (lx, ly) and the lane's own sample count
template <bool MULTI>
HR_DEV void mask_window(const MaskRows& R, int lx, int ly, int& sum, int& own)
{
    uint32_t a0 = 0, a1 = 0, a2 = 0;
    // several planes: rolled in groups so that the 51 row words are not all live at once (102 VGPRs fully unrolled)
#pragma clang loop unroll_count(MULTI ? 6 : 17)
    for (int yy = 0; yy <= 16; yy++)
    {
        a0 = __builtin_popcount(__builtin_amdgcn_ubfe(R.b0[ly + yy], (uint32_t)lx, 17u)) + a0;
        if (MULTI)
        {
            a1 = __builtin_popcount(__builtin_amdgcn_ubfe(R.b1[ly + yy], (uint32_t)lx, 17u)) + a1;
            a2 = __builtin_popcount(__builtin_amdgcn_ubfe(R.b2[ly + yy], (uint32_t)lx, 17u)) + a2;
        }
    }
    sum = (int)(a0 + 2u * a1 + 4u * a2);
    const uint32_t sh = (uint32_t)lx + 8u;
    own = (int)((R.b0[ly + 8] >> sh) & 1u);
    if (MULTI) own += (int)(((R.b1[ly + 8] >> sh) & 1u) * 2u + ((R.b2[ly + 8] >> sh) & 1u) * 4u);
}


} // namespace hr
