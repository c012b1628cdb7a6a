// Workgroup -> tile mapping for the image kernels (denoise, blur, upsample, probe-grid sample).
//
// The dispatcher deals the workgroups of a launch round-robin over the 8 XCDs in linear order (x fastest), and every XCD has
// its own L2: with the identity mapping a tile's apron (the a-trous rings, the blur's reach, the reprojection footprint) is
// fetched by up to eight L2s, which the FETCH_SIZE counters show as 1.5-1.9x the algorithmic bytes (docs/EXPERIMENTS.md §4.4).
// block_xy<R>() hands XCD k runs of R whole tile rows instead (dealt round-robin down the screen, so every XCD still samples the
// whole image), so horizontally — and for R > 1 vertically — neighbouring tiles share one L2.
//
// Measured (round 3, docs/EXPERIMENTS.md §4.4): it pays for exactly two kernels, the fused AO blur (-12 % at 1080p) and the fused reflections
// a-trous 0 + 1 (-7..12 %); every other image kernel is FASTER with the identity (shadows a-trous 2 / 3 +12 %, upsample +10 %,
// probe-grid sample +3 %: the fine interleave spreads the concurrently running tiles over more DRAM channels, and the kernels
// with a per-tile early-out lose their load balance), and one contiguous eighth per XCD (R = -1) is 30-60 % slower for those.
// The trace kernels need the round-robin deal as their load balancer (docs/EXPERIMENTS.md §4.2).
#pragma once
#include "device_math.h"

#ifndef HR_XCD_ROWS_ALL
#define HR_XCD_ROWS_ALL -2     // A/B switch: force one mapping on every image kernel (-2: each kernel's own choice)
#endif

// Grid width of the identity-mapped kernels (docs/EXPERIMENTS.md R5.7).  HR_COLS8: bit mask of kernel families (0 shadows a-trous 4 / 8, 1 reflections
// a-trous, 2 upsample, 3 probe-grid sample, 4 shadows temporal, 5 AO temporal, 6 reflections temporal, 7 shadows a-trous 0+1 / LDS);
// HR_COLS_MODE 2: every width is made odd; 1: a width that is a multiple of 8 workgroups gets HR_COLS_SKEW more columns; 0: every width is
// padded up to a multiple of 8.
#ifndef HR_COLS8
#define HR_COLS8 ((1 << 0) | (1 << 4) | (1 << 7))   // the shadow pass's kernels: the only ones it pays for
#endif
#ifndef HR_COLS_MODE
#define HR_COLS_MODE 2
#endif
#ifndef HR_COLS_SKEW
#define HR_COLS_SKEW 1
#endif

namespace hr {

// Grid width and the XCDs.  With the identity mapping workgroup (bx, by) runs on XCD (by * gridDim.x + bx) mod 8.  When gridDim.x is a multiple of 8
// (3840 wide: 120 columns of 32) that is bx mod 8: XCD k only ever touches tile columns k, k + 8, ... — a tile and the tiles above and below it share
// an L2 (the 0.45x counter traffic of the 4K shadow a-trous, against 1.6-2.3x for the same kernels on 960 / 1920 wide images), but every XCD's
// requests then fall on the same eighth of the address residues.  Measured both ways (R5.7): ALIGNING the widths that are not (30 -> 32, 60 -> 64
// columns) makes every kernel slower (a-trous +20-30 %, temporal +3-10 %); DE-ALIGNING the ones that are (120 -> 121 columns at 3840 wide) makes
// the shadow pass's kernels 4-8 % faster and leaves the others where they were; an ODD width (60 -> 61 at 1920 wide: the XCD of a column then
// rotates through all eight from tile row to tile row instead of alternating between two) is worth another 1-4 % to the shadow pass's kernels
// at 1080p.  The extra workgroups find no pixel and leave.
static inline int grid_cols(int gx, int bit)
{
    if (!((HR_COLS8 >> bit) & 1)) return gx;
    if (HR_COLS_MODE == 0) return (gx + 7) & ~7;
    if (HR_COLS_MODE == 2) return gx | 1;   // every width odd: the XCD of a column rotates by an odd step from tile row to tile row
    return (gx & 7) == 0 ? gx + HR_COLS_SKEW : gx;
}

// ROWS = 0: identity; ROWS > 0: runs of ROWS tile rows per XCD; ROWS = -1: one contiguous eighth of the tiles per XCD
template <int ROWS_>
HR_DEV uint2 block_xy()
{
    constexpr int ROWS = HR_XCD_ROWS_ALL == -2 ? ROWS_ : HR_XCD_ROWS_ALL;
    if constexpr (ROWS == 0) {
        return make_uint2(blockIdx.x, blockIdx.y);
    } else {
        const uint32_t gx = gridDim.x, n = gx * gridDim.y;
        const uint32_t lin = blockIdx.y * gx + blockIdx.x;
        uint32_t       l2;
        if constexpr (ROWS < 0) {
            // XCD k owns the linear ids k, k + 8, ...: q + (k < r) of them; its eighth starts after the eighths of XCDs 0..k-1
            const uint32_t q = n >> 3, r = n & 7u, xcd = lin & 7u;
            l2 = xcd * q + (xcd < r ? xcd : r) + (lin >> 3);
        } else {
            // groups of 8 R tile rows: inside a group XCD k (linear ids = k mod 8) walks rows [k R, k R + R); the ragged tail keeps the identity
            const uint32_t C = gx * ROWS, S = 8u * C, g = lin / S, rem = lin - g * S;
            l2 = (g + 1) * S <= n ? g * S + (rem & 7u) * C + (rem >> 3) : lin;
        }
        const uint32_t y = l2 / gx;
        return make_uint2(l2 - y * gx, y);
    }
}

}  // namespace hr
