// Heaviest-first launch order for the one-wave-per-tile trace kernels.
//
// A trace launch ends with a long tail: the dispatcher hands out tiles in blockIdx order, the expensive tiles (long rays through the
// dense parts of the scene) sit wherever the camera puts them, and the ones that start late run on an almost empty GPU — at 1080p every
// shadow tile has STARTED after 56 us of a 92 us launch, and the slowest tile alone takes 54 us (tools/timeline.py, round 4).  Tile
// costs barely change from one frame to the next (static scene, camera moving a fraction of a pixel), so every wave stores how long it
// lived (100 MHz ticks, 16 bits), a small kernel turns last frame's costs into a launch order (64 quarter-octave buckets, heaviest
// bucket first — a counting sort, no comparison sort), and the next launch maps blockIdx through it: the long tiles start first and
// the short ones fill the tail.  Every tile is still computed exactly once by the same code: results do not depend on the order.
#pragma once
#include "hr_internal.h"

namespace hr {

#define HR_ORDER_BUCKETS 64

// quarter-octave bucket of a cost: 4 * floor(log2 c) + the next two bits — 19 % relative resolution over the whole 16-bit range
__device__ __forceinline__ uint32_t order_bucket(uint32_t c)
{
    c |= 4u;                                  // costs below 4 ticks share the lowest buckets
    const uint32_t lz = 31u - (uint32_t)__clz((int)c);
    return lz * 4u + ((c >> (lz - 2u)) & 3u);   // lz <= 15 -> at most 63
}

// Counting sort, heaviest bucket first.  G workgroups (a multiple of 8): group g sorts the tiles of residue class g (tile % G == g, a
// representative sample of the image) and writes them to the launch slots of the same class — workgroups are dealt round-robin over
// the 8 XCDs, so every XCD walks G / 8 interleaved heaviest-first lists.  Per wave a private histogram (plain LDS atomics: most tiles are
// cheap and share a bucket — one LDS address per wave, not one for the workgroup), then an exclusive prefix in (bucket descending,
// wave) order (a shuffle scan), then the scatter with returning LDS atomics.
#define HR_ORDER_GROUPS_MIN 8
#define HR_ORDER_GROUPS_MAX 64
static __global__ __launch_bounds__(1024) void k_tile_order(const uint16_t* __restrict__ cost, uint32_t* __restrict__ order, int n)
{
    __shared__ uint32_t s_cnt[16][HR_ORDER_BUCKETS];
    __shared__ uint32_t s_tot[HR_ORDER_BUCKETS];
    const int g = blockIdx.x, G = gridDim.x;
    const int tid = threadIdx.x, wave = tid >> 6;
    const int n_g = (n - g + G - 1) / G;
    (&s_cnt[0][0])[tid] = 0u;
    __syncthreads();
    for (int k = tid; k < n_g; k += 1024) atomicAdd(&s_cnt[wave][order_bucket(cost[(size_t)k * G + g])], 1u);
    __syncthreads();
    {
        // exclusive prefix over the 16 x 64 counters in (bucket descending, wave ascending) order: thread i owns element i of that order
        const int b = HR_ORDER_BUCKETS - 1 - (tid >> 4), w = tid & 15, lane = tid & 63;
        const uint32_t v = s_cnt[w][b];
        uint32_t inc = v;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, o); if (lane >= o) inc += t; }
        if (lane == 63) s_tot[wave] = inc;
        __syncthreads();
        uint32_t before = 0u;
        for (int k = 0; k < wave; k++) before += s_tot[k];
        s_cnt[w][b] = before + inc - v;
    }
    __syncthreads();
    for (int k = tid; k < n_g; k += 1024)
    {
        const uint32_t tile = (uint32_t)k * (uint32_t)G + (uint32_t)g;
        const uint32_t pos  = atomicAdd(&s_cnt[wave][order_bucket(cost[tile])], 1u);
        order[(size_t)pos * G + g] = tile;
    }
}

// Host side of one pass: the two side buffers and whether last frame's order may be used.  The sort follows the trace kernel on the
// caller's stream (capturable into a hipGraph like everything else).  Running it on a side stream beside the denoise kernels was
// measured and is slower: two cross-stream event waits per frame cost more than the ~4 us kernel (docs/EXPERIMENTS.md R4.4).
struct TileOrder
{
    DevBuf cost, order;
    int    n       = 0;
    bool   enabled = true;    // developer A/B switch HR_TILE_ORDER=0 (read once at create)
    bool   valid   = false;   // `order` holds a permutation of 0..n-1 built from a launch over the same tiles
    hr_status init(int n_tiles)
    {
        n = n_tiles;
        valid = false;
        if (!enabled) return HR_OK;
        hr_status s = cost.alloc((size_t)n_tiles * 2);
        if (s != HR_OK) return s;
        return order.alloc((size_t)n_tiles * 4);
    }
    bool active(int n_tiles) const { return enabled && n_tiles == n; }   // a pass asked to trace a different region keeps blockIdx order
    // arguments of the trace launch (nullptr: blockIdx order / no cost record)
    const uint32_t* order_arg(int n_tiles) const { return active(n_tiles) && valid ? (const uint32_t*)order.p : nullptr; }
    uint16_t*       cost_arg(int n_tiles) const { return active(n_tiles) ? (uint16_t*)cost.p : nullptr; }
    // after the trace launch, on the same stream: next frame's order
    hr_status update(int n_tiles, hipStream_t st)
    {
        if (!active(n_tiles)) return HR_OK;
        int groups = (n / 2048) & ~7;    // ~2000 tiles per workgroup: two rounds of its 1024 threads
        groups = groups < HR_ORDER_GROUPS_MIN ? HR_ORDER_GROUPS_MIN : (groups > HR_ORDER_GROUPS_MAX ? HR_ORDER_GROUPS_MAX : groups);
        hipLaunchKernelGGL(k_tile_order, dim3(groups), dim3(1024), 0, st, (const uint16_t*)cost.p, (uint32_t*)order.p, n);
        HR_HIP(hipGetLastError());
        valid = true;
        return HR_OK;
    }
};

} // namespace hr
