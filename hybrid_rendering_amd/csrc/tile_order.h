// Heaviest-first launch order for the one-wave-per-tile trace kernels.
//
// A trace launch ends with a long tail: the dispatcher hands out tiles in blockIdx order, the expensive tiles (long rays through the
// dense parts of the scene) sit wherever the camera puts them, and the ones that start late run on an almost empty GPU — at 1080p every
// shadow tile has STARTED after 56 us of a 92 us launch, and the slowest tile alone takes 54 us (tools/timeline.py, round 4).  Tile
// costs barely change from one frame to the next (static scene, camera moving a fraction of a pixel), so every wave stores how long it
// lived (100 MHz ticks, 16 bits), a counting sort (64 quarter-octave buckets, heaviest bucket first; a few workgroups riding along with
// the pass's next launch, see TileOrder) turns last frame's costs into a launch order, and the next launch maps blockIdx through it:
// the long tiles start first and the short ones fill the tail.  Every tile is still computed exactly once by the same code: results
// do not depend on the order.  Measured: shadow / AO / reflections trace -17 / -14 / -12 % at 1080p (docs/EXPERIMENTS.md R4.4).
#pragma once
#include "hr_internal.h"
#include "pass_args.h"
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

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
// wave) order (a shuffle scan: one element per thread), then the scatter with returning LDS atomics.
// THREADS = workgroup size of the kernel this runs in: 1024 alone (k_tile_order), 256 as extra workgroups of a pass's temporal kernel.
#define HR_ORDER_GROUPS_MIN 8
// min_spread (shadow pass: HR_ORDER_COHERENT_SPREAD; 0 = always sort): a group whose costs are NOT spread widely — the 90th percentile
// less than min_spread quarter-octaves above the median — keeps the image order.  Shadow rays towards one light are coherent, neighbouring
// tiles walk the same BVH nodes, and a sorted launch gives that up: measured on the shadow trace, p90 / median 9.2 (bench scene) -17 %,
// 9.6 (4x the triangles) -10 %, 1.5 (same scene, a light most pixels face) +1 %, 1.9 (hard tier) +8 %.  The incoherent AO and
// reflection rays gain 12-14 % at a spread of 1.5-1.9 and always sort.
#define HR_ORDER_COHERENT_SPREAD 6   // quarter-octaves: p90 >= 2.8 x median
template <int THREADS>
__device__ __forceinline__ void tile_order_block(const uint16_t* __restrict__ cost, uint32_t* __restrict__ order, int n, int g, int G, int min_spread)
{
    constexpr int WAVES = THREADS / 64;
    __shared__ uint32_t s_cnt[WAVES][HR_ORDER_BUCKETS];
    __shared__ uint32_t s_tot[WAVES];
    __shared__ uint32_t s_keep;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n_g = (n - g + G - 1) / G;
    (&s_cnt[0][0])[tid] = 0u;
    __syncthreads();
    for (int k = tid; k < n_g; k += THREADS) atomicAdd(&s_cnt[wave][order_bucket(cost[(size_t)k * G + g])], 1u);
    __syncthreads();
    {
        // thread i owns element i of the (bucket descending, wave ascending) order
        const int b = HR_ORDER_BUCKETS - 1 - tid / WAVES, w = tid % WAVES;
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
    if (wave == 0)
    {
        // s_cnt[0][b] now = tiles in the buckets above b: lane b + 1 asks whether its bucket and those above hold a tenth / a half of the tiles
        const uint32_t at_least = lane == 0 ? (uint32_t)n_g : s_cnt[0][lane - 1];
        const unsigned long long b90 = __ballot(at_least * 10u >= (uint32_t)n_g), b50 = __ballot(at_least * 2u >= (uint32_t)n_g);
        const int p90 = 63 - __clzll((long long)b90), p50 = 63 - __clzll((long long)b50);   // lane 0 always votes: never empty
        if (lane == 0) s_keep = (min_spread > 0 && p90 - p50 < min_spread) ? 1u : 0u;
    }
    __syncthreads();
    const bool keep = s_keep != 0u;
    for (int k = tid; k < n_g; k += THREADS)
    {
        const uint32_t tile = (uint32_t)k * (uint32_t)G + (uint32_t)g;
        const uint32_t pos  = keep ? (uint32_t)k : atomicAdd(&s_cnt[wave][order_bucket(cost[tile])], 1u);
        order[(size_t)pos * G + g] = tile;
    }
}

static __global__ __launch_bounds__(1024) void k_tile_order(const uint16_t* __restrict__ cost, uint32_t* __restrict__ order, int n, int min_spread)
{
    tile_order_block<1024>(cost, order, n, (int)blockIdx.x, (int)gridDim.x, min_spread);
}

// Host side of one pass: the two side buffers, whether last frame's order may be used, and who runs the sort.  A launch of its own
// costs ~6 us of stream time for ~1 us of work (launch, cold loads, kernel boundary), so the sort rides along as a few extra
// workgroups of the pass's NEXT launch, the tolerance-mode temporal kernel (TileSortArgs in its argument block: the temporal stage
// does not depend on it, the trace launch before it has finished, and the next trace launch must follow the temporal stage anyway —
// it overwrites the mask that stage reads).  When no such launch comes (denoiser off, parity mode) the next trace call runs
// k_tile_order first.  Running the sort on a side stream beside the denoise kernels was measured and is slower: two cross-stream
// event waits per frame cost more than the kernel (docs/EXPERIMENTS.md R4.4).
struct TileOrder
{
    DevBuf cost, order;
    int    n       = 0;
    bool   enabled = true;    // developer A/B switch HR_TILE_ORDER=0 (read once at create)
    bool   ride_along = true; // developer A/B switch HR_TILE_ORDER_FUSED=0: always a launch of its own
    bool   valid   = false;   // a sort has been enqueued since creation (informational: `order` is a permutation either way, see identity())
    bool   pending = false;   // costs of a trace launch are waiting to be sorted
    int    min_spread = 0;    // > 0 (coherent rays: the shadow pass): sort only widely spread costs, see tile_order_block (developer switch HR_TILE_ORDER_SPREAD)
    std::string dump_path;    // developer switch HR_DEBUG_TILE_COSTS=<prefix>: every launch's costs go to <prefix>.<tag> (synchronises: never in a timed run)
    const char* tag = "pass";
    void dump(hipStream_t st)
    {
        if (dump_path.empty() || !cost.p || n <= 0) return;
        std::vector<uint16_t> host((size_t)n);
        if (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(host.data(), cost.p, host.size() * 2, hipMemcpyDeviceToHost) != hipSuccess) return;
        if (FILE* f = fopen((dump_path + "." + tag).c_str(), "wb")) { fwrite(host.data(), 2, host.size(), f); fclose(f); }
    }
    hr_status init(int n_tiles)
    {
        n = n_tiles;
        valid = pending = false;
        if (const char* e = getenv("HR_DEBUG_TILE_COSTS")) dump_path = e;
        if (const char* e = getenv("HR_TILE_ORDER_FUSED")) ride_along = atoi(e) != 0;
        if (const char* e = getenv("HR_TILE_ORDER_SPREAD")) min_spread = min_spread ? atoi(e) : 0;
        if (!enabled) return HR_OK;
        hr_status s = cost.alloc((size_t)n_tiles * 2);
        if (s != HR_OK) return s;
        if ((s = order.alloc((size_t)n_tiles * 4)) != HR_OK) return s;
        return identity();
    }
    // `order` ALWAYS holds a permutation of 0..n-1: the identity from creation until the first sort lands, a sorted list afterwards (a sort
    // writes every slot of its residue class).  The trace launch therefore always takes the list — its arguments are the same from frame 0 on,
    // so a hipGraph captured on the FIRST frame (whose temporal launch already carries the riding sort) replays with the launch order like
    // one captured later, and a sort that never ran (a captured frame that was never launched, a failed launch) leaves a
    // valid list behind instead of the zeros of a fresh buffer.  Synchronous: creation / reset, never inside a frame.
    hr_status identity()
    {
        if (!enabled || !order.p || n <= 0) return HR_OK;
        std::vector<uint32_t> id((size_t)n);
        for (int i = 0; i < n; i++) id[(size_t)i] = (uint32_t)i;
        HR_HIP(hipMemcpy(order.p, id.data(), id.size() * 4, hipMemcpyHostToDevice));
        valid = false;
        return HR_OK;
    }
    // hr_*_launch_order (introspection): the list as the next trace launch will read it
    hr_status read(uint32_t* out, int32_t* n_tiles, hipStream_t st)
    {
        const bool on = enabled && order.p && n > 0;
        if (n_tiles) *n_tiles = on ? n : 0;
        if (!out || !on) return HR_OK;
        // the DEVICE, not `st`: under HR_FRAME_GRAPH the pass's last stream is the internal capture stream, while the riding sort that writes
        // `order` runs on whatever stream the graph was launched on
        (void)st;
        HR_HIP(hipDeviceSynchronize());
        HR_HIP(hipMemcpy(out, order.p, (size_t)n * 4, hipMemcpyDeviceToHost));
        return HR_OK;
    }
    // hr_*_reset_history: forget costs nobody has sorted yet (they may belong to a frame that never ran).  The list itself stays — it is a
    // permutation whatever happened — unless the caller asks for the image order back (device idle: reset is not called inside a frame)
    void invalidate() { pending = false; }
    bool active(int n_tiles) const { return enabled && n_tiles == n; }   // a pass asked to trace a different region keeps blockIdx order
    static int groups_for(int n, int threads)
    {
        int g = (n / (2 * threads)) & ~7;    // two rounds of the workgroup's threads
        return g < HR_ORDER_GROUPS_MIN ? HR_ORDER_GROUPS_MIN : (g > 256 ? 256 : g);
    }
    // First thing in the pass's trace call: costs nobody has sorted yet get a launch of their own.
    hr_status flush(hipStream_t st)
    {
        if (!pending) return HR_OK;
        pending = false;
        hipLaunchKernelGGL(k_tile_order, dim3(groups_for(n, 1024)), dim3(1024), 0, st, (const uint16_t*)cost.p, (uint32_t*)order.p, n, min_spread);
        HR_HIP(hipGetLastError());
        valid = true;
        return HR_OK;
    }
    // arguments of the trace launch (nullptr: blockIdx order / no cost record)
    const uint32_t* order_arg(int n_tiles) const { return active(n_tiles) ? (const uint32_t*)order.p : nullptr; }
    uint16_t*       cost_arg(int n_tiles) const { return active(n_tiles) ? (uint16_t*)cost.p : nullptr; }
    // after a trace launch that recorded costs
    hr_status traced(int n_tiles, hipStream_t st)
    {
        if (!active(n_tiles)) return HR_OK;
        dump(st);
        pending = true;
        return ride_along ? HR_OK : flush(st);
    }
    // For the pass's tolerance-mode temporal launch (256-thread workgroups): the sort as extra grid rows behind the launch's own (the
    // launcher appends them and fills in row0).  groups == 0 when there is nothing to sort.
    TileSortArgs ride()
    {
        TileSortArgs t { nullptr, nullptr, 0, 0, 0, 0 };
        if (!pending || !enabled) return t;
        pending = false;
        valid   = true;
        t.cost = (const uint16_t*)cost.p; t.order = (uint32_t*)order.p; t.n = n; t.groups = groups_for(n, 256); t.min_spread = min_spread;
        return t;
    }
};

// at the top of a kernel whose argument block carries a TileSortArgs: true = this workgroup was one of the sort's (or a spare one)
template <int THREADS>
__device__ __forceinline__ bool tile_order_rides(const TileSortArgs& t)
{
    if (t.groups == 0 || (int)blockIdx.y < t.row0) return false;
    const int g = ((int)blockIdx.y - t.row0) * (int)gridDim.x + (int)blockIdx.x;
    if (g < t.groups) tile_order_block<THREADS>(t.cost, t.order, t.n, g, t.groups, t.min_spread);
    return true;
}

} // namespace hr
