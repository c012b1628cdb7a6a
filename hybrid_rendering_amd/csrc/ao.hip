// RayTracedAO on MI355X — HIP replacement for src/ray_traced_ao.{h,cpp} and src/shaders/ao/*.
//   ray_trace()              ray_traced_ao.cpp:863-903,  ao_ray_trace.comp:90-126            -> k_ao_trace
//   temporal_accumulation()  :983-1028,                  ao_denoise_reprojection.comp:191-260 -> k_ao_temporal
//   bilateral_blur()         :1032-1137 (dir (1,0) then (0,1)), ao_denoise_bilateral_blur.comp:75-139 -> k_ao_blur
//   upsample()               :918-955,                   ao_upsample.comp:63-112             -> k_upsample (upsample.h)
// Extension (BASELINE.json configs[2]): spp samples per pixel (one mask plane per sample,
// sample index = spp*num_frames + s); spp = 1 is the reference behaviour.
#include "hr_internal.h"
#include "reproject.h"
#include "traverse.h"
#include "mask_window.h"
#include "upsample.h"
#include "pass_args.h"
#include "tile_order.h"

using namespace hr;

// brdf.glsl:8-32 sample_cosine_lobe + make_rotation_matrix
HR_DEV f3 sample_cosine_lobe(f3 n, float rx, float ry)
{
    rx = max2(0.00001f, rx);
    ry = max2(0.00001f, ry);
    const float phi = 2.0f * HR_M_PI * ry;
    const float ct = hr_sqrt(rx), st = hr_sqrt(1.0f - rx);
    float s, c;
    det_sincos(phi, s, c);
    const f3 t   = mk3(st * c, st * s, ct);
    const f3 ref = fabsf(dot3(n, mk3(0.0f, 1.0f, 0.0f))) > 0.99f ? mk3(0.0f, 0.0f, 1.0f) : mk3(0.0f, 1.0f, 0.0f);
    const f3 x   = normalize3(cross3(ref, n));
    const f3 y   = cross3(n, x);
    return normalize3(mk3((x.x * t.x + y.x * t.y) + n.x * t.z, (x.y * t.x + y.y * t.y) + n.y * t.z, (x.z * t.x + y.z * t.y) + n.z * t.z));
}

struct AOTraceArgs
{
    float              vpi[16];
    const float*       depth;
    const uint2*       gb2;
    const uint8_t*     sobol;
    const uint8_t*     sr;
    uint32_t*          mask;       // [spp][mh][mw]
    uint32_t*          ray_slots;  // rays per 8x8 tile (a single shared atomic counter serialises the waves)
    const Node8*       nodes;
    const TriGPU*      tris;
    unsigned long long* stats;
    int                w, h, y0, y1, mw, mh;
    int                tiles_x, tiles_y, tile_y0;
    float              bias, ray_length;
    uint32_t           num_frames;
    int                spp;
    const uint32_t*    order;      // nullable: launch slot -> tile, last frame's heaviest tiles first (tile_order.h)
    uint16_t*          cost;       // nullable: per tile, how long its wave lived (100 MHz ticks)
    // nullable: table of traversal entry nodes over a grid of cells of edge 1 / grid_inv_c from grid_lo (AoEntryGrid below)
    const uint32_t*    grid;
    float              grid_lo[3], grid_inv_c;
    int                grid_n[3];
};

// Entry-node table (round 5).  Every sample ray of a pixel stays within ray_length of its origin, so the traversals may start at the deepest
// BVH node that holds all the geometry of that ball (entry_node_for_box) — until round 4 found per pixel by a descent from the root: 3.5
// dependent node tests per pixel, 5 for the slowest lane of a wave, ~18 % of the kernel.  The descent only depends on WHERE the ball is, so it
// is done once per CELL of a grid over the scene (cell box grown by the ray length: a superset of the ball of every origin in the cell — any
// node whose subtree holds all geometry touching a superset is a valid entry, so the hit set and the masks are unchanged) and a pixel looks
// its cell up: one load instead of the descent, for an entry node that is on average 0.2-0.3 node steps per ray shallower
// (tools/bvh_eval.cpp AO_ENTRY_CELL study: cells of 3.5 / 7 / 14 units at ray length 7: +0.17 / +0.30 / +0.48 node steps per ray against
// 3.45 descent tests per pixel saved).  The table belongs to the pass, is rebuilt (one thread per cell, in-stream) when the scene or the ray
// length changes, and has a fixed capacity so that nothing is allocated inside a frame (a frame may be under hipGraph capture).
#ifndef HR_AO_GRID_CELLS
#define HR_AO_GRID_CELLS (1 << 22)      // 16 MB (developer A/B: -DHR_AO_GRID_CELLS=...)
#endif
constexpr int kAoGridCells = HR_AO_GRID_CELLS;
__global__ __launch_bounds__(256) void k_ao_entry_grid(const Node8* __restrict__ nodes, uint32_t* __restrict__ cells, float lox, float loy, float loz, float c, int nx, int ny, int nz, float grow)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nx * ny * nz) return;
    const int ix = i % nx, iy = (i / nx) % ny, iz = i / (nx * ny);
    // the cell's box in the arithmetic the lookup uses (lo + index * c), grown by the ray length and by a sliver of the cell for the
    // rounding of the lookup's (o - lo) / c
    const float g  = grow + 1e-3f * c;
    const f3    bl = mk3(lox + (float)ix * c - g, loy + (float)iy * c - g, loz + (float)iz * c - g);
    const f3    bh = mk3(lox + (float)(ix + 1) * c + g, loy + (float)(iy + 1) * c + g, loz + (float)(iz + 1) * c + g);
    cells[i] = entry_node_for_box(nodes, bl, bh);
}

#ifdef HR_TRACE_DIVERGENCE
static __device__ unsigned long long g_div_ao[8];
extern "C" int hr_debug_divergence_ao(uint64_t* out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_div_ao), sizeof(g_div_ao)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_div_ao), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif
#ifndef AO_TRACE_WAVES
#define AO_TRACE_WAVES 1   // waves (8x8 tiles) per workgroup, see k_shadows_trace: finished waves' slots back-fill at once
#endif
#ifndef AO_SEQ
#define AO_SEQ 0    // N: the sample rays of a pixel, N at a time, walked back to back per lane inside ONE wave loop (traverse.h trace_any_seq);
                    // 0: one wave-level traversal per sample.  Measured (round 3, 4 spp, bit-identical masks): 1080p 395 us (0) / 468 (2) / 498 (4),
                    // 4K 1303 / 1540 / 1563 — every iteration of the wave pays for the ray switch of whichever lane just finished, and the
                    // triangle tests lose the wave-cooperative path; kept as a measured A/B path only
#endif
#ifndef AO_ORDER
#define AO_ORDER HR_ORDER_SLOTS   // the short AO rays (97 % of them miss) gain nothing from a visiting order (tools/bvh_eval.cpp): plain slot order, no rev logic
#endif
#ifndef AO_COOP
#define AO_COOP 1   // wave-cooperative triangle tests (traverse.h trace_coop) for the AO rays: 0.418 -> 0.399 ms at 1080p, 4 spp
#endif
template <bool STATS>
#ifndef AO_TRACE_EU
#define AO_TRACE_EU 6   // minimum waves per SIMD the register allocator must leave room for.  Round 4 (new tree): 1 / 4 / 5 / 6 / 7 / 8 -> 372 / 373 / 372 / 361-368 / 374 / 372 us at 1080p, 1208 -> 1156 at 4K for 6;
                        // round 3: 1 / 6 / 8 -> 393 / 389 / 396
#endif
__global__ __launch_bounds__(64 * AO_TRACE_WAVES, AO_TRACE_EU) void k_ao_trace(AOTraceArgs a)
{
    __shared__ uint32_t s_stack[AO_TRACE_WAVES][HR_STACK_ENTRIES * 64];
#if AO_COOP && !AO_SEQ
    __shared__ CoopWave s_coop[AO_TRACE_WAVES];
#endif
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int launch_slot = blockIdx.x * AO_TRACE_WAVES + wave;
    if (launch_slot >= a.tiles_x * a.tiles_y) return;
    const int tile = a.order ? (int)a.order[launch_slot] : launch_slot;
    const unsigned long long t_begin = a.cost ? wall_clock64() : 0ull;
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x + a.tile_y0;
    const int x = tx * 8 + (lane & 7), y = ty * 8 + (lane >> 3);
    bool  active = false;
    f3    ro = mk3(0, 0, 0), N = mk3(0, 0, 1);
    const int kind = trace_lane_kind(x, y, a.w, a.h, a.y0, a.y1);   // 2: edge thread of a ragged image (device_math.h)
    if (kind)
    {
        const float d = kind == 1 ? a.depth[(size_t)y * a.w + x] : 0.0f;
        if (d != 1.0f)
        {
            const float tu = __fdiv_rn((float)x + 0.5f, (float)a.w), tv = __fdiv_rn((float)y + 0.5f, (float)a.h);
            const f3    P  = world_pos_from_depth(tu, tv, d, a.vpi);
            const uint2 g2 = kind == 1 ? a.gb2[(size_t)y * a.w + x] : make_uint2(0u, 0u);
            N      = oct_decode(h2f_lo(g2.x), h2f_hi(g2.x));
            ro     = add3(P, scale3(N, a.bias));
            active = true;
        }
    }
    uint32_t nn = 0, nt = 0;
    HR_DIV(DivCounters dv = {};)
    const unsigned long long fired = __ballot(active);
    // every sample ray of the pixel stays within ray_length of its origin: find, once, the deepest BVH node that holds all
    // the geometry of that ball and start the traversals there instead of at the root (traverse.h: entry_node_for_box)
    uint32_t entry = 0u;
    if (active)
    {
        const float r = a.ray_length * 1.0001f + 1e-4f;
        bool looked_up = false;
        if (a.grid)   // wave-uniform
        {
            const float gx = (ro.x - a.grid_lo[0]) * a.grid_inv_c, gy = (ro.y - a.grid_lo[1]) * a.grid_inv_c, gz = (ro.z - a.grid_lo[2]) * a.grid_inv_c;
            const int   ix = (int)floorf(gx), iy = (int)floorf(gy), iz = (int)floorf(gz);
            if (ix >= 0 && iy >= 0 && iz >= 0 && ix < a.grid_n[0] && iy < a.grid_n[1] && iz < a.grid_n[2])   // an origin outside the scene's box (bias): the descent
            {
                entry     = a.grid[((size_t)iz * a.grid_n[1] + iy) * a.grid_n[0] + ix];
                looked_up = true;
            }
        }
        if (!looked_up) entry = entry_node_for_box(a.nodes, mk3(ro.x - r, ro.y - r, ro.z - r), mk3(ro.x + r, ro.y + r, ro.z + r));
    }
#if AO_SEQ && !defined(HR_DEV_PATHS)
#error "-DAO_SEQ=N is an A/B path: build with -DHR_DEV_PATHS"
#endif
#if AO_SEQ
    if (!STATS)
    {
        // the sample rays of a pixel, AO_SEQ at a time, back to back inside one wave loop (traverse.h trace_any_seq)
        for (int s0 = 0; s0 < a.spp; s0 += AO_SEQ)
        {
            f3 dir[AO_SEQ];
            const int n = a.spp - s0 < AO_SEQ ? a.spp - s0 : AO_SEQ;
#pragma unroll
            for (int k = 0; k < AO_SEQ; k++)
            {
                dir[k] = mk3(0.0f, 0.0f, 1.0f);
                if (active && k < n)
                {
                    const int   idx = (int)a.num_frames * a.spp + s0 + k;
                    const float r0  = sample_blue_noise(x, y, idx, 0, a.sobol, a.sr), r1 = sample_blue_noise(x, y, idx, 1, a.sobol, a.sr);
                    dir[k] = sample_cosine_lobe(N, r0, r1);
                }
            }
            float tm[AO_SEQ];
#pragma unroll
            for (int k = 0; k < AO_SEQ; k++) tm[k] = a.ray_length;
            const uint32_t occ = trace_any_seq<AO_SEQ, AO_ORDER>(active, n, a.nodes, a.tris, ro, dir, 0.01f, tm, s_stack[wave], lane, entry HR_DIV(, &dv));
            for (int k = 0; k < n; k++)
            {
                const unsigned long long bits = __ballot(active && !((occ >> k) & 1u));
                if (lane == 0)
                {
                    const int my = ty * 2;
                    uint32_t* m  = a.mask + (size_t)(s0 + k) * a.mh * a.mw;
                    if (my * 4 >= a.y0 && my * 4 < a.y1) m[(size_t)my * a.mw + tx] = (uint32_t)(bits & 0xffffffffull);
                    if ((my + 1) * 4 < a.y1 && (my + 1) * 4 < a.h) m[(size_t)(my + 1) * a.mw + tx] = (uint32_t)(bits >> 32);
                }
            }
        }
    }
    else
#endif
    for (int s = 0; s < a.spp; s++)
    {
        bool visible = false;
#if AO_COOP && !AO_SEQ
        if (!STATS)
        {
            f3 dir = mk3(0.0f, 0.0f, 1.0f);
            if (active)
            {
                const int   idx = (int)a.num_frames * a.spp + s;
                const float r0  = sample_blue_noise(x, y, idx, 0, a.sobol, a.sr), r1 = sample_blue_noise(x, y, idx, 1, a.sobol, a.sr);
                dir = sample_cosine_lobe(N, r0, r1);
            }
            visible = trace_coop<true, AO_ORDER>(active, a.nodes, a.tris, ro, dir, 0.01f, a.ray_length, s_stack[wave], s_coop[wave], lane, entry).prim != 0 && active;
        }
        else
#endif
        if (active)
        {
            const int   idx = (int)a.num_frames * a.spp + s;
            const float r0  = sample_blue_noise(x, y, idx, 0, a.sobol, a.sr), r1 = sample_blue_noise(x, y, idx, 1, a.sobol, a.sr);
            const f3    dir = sample_cosine_lobe(N, r0, r1);
            visible         = !trace_any<STATS, AO_ORDER>(a.nodes, a.tris, ro, dir, 0.01f, a.ray_length, s_stack[wave], lane, nn, nt, entry HR_DIV(, &dv));
        }
        const unsigned long long bits = __ballot(visible);
        if (lane == 0)
        {
            const int my = ty * 2;
            uint32_t* m  = a.mask + (size_t)s * a.mh * a.mw;
            if (my * 4 >= a.y0 && my * 4 < a.y1) m[(size_t)my * a.mw + tx] = (uint32_t)(bits & 0xffffffffull);
            if ((my + 1) * 4 < a.y1 && (my + 1) * 4 < a.h) m[(size_t)(my + 1) * a.mw + tx] = (uint32_t)(bits >> 32);
        }
    }
    HR_DIV(div_flush(dv, g_div_ao);)
    if (STATS)
        for (int o = 32; o > 0; o >>= 1) { nn += __shfl_down(nn, o); nt += __shfl_down(nt, o); }
    if (lane == 0)
    {
        a.ray_slots[(size_t)ty * a.tiles_x + tx] = (uint32_t)__popcll(fired) * (uint32_t)a.spp;
        if (a.cost)
        {
            const unsigned long long ticks = wall_clock64() - t_begin;
            a.cost[tile] = (uint16_t)(ticks > 65535ull ? 65535ull : ticks);
        }
        if (STATS && a.stats)
        {
            atomicAdd(a.stats + 0, (unsigned long long)nn);
            atomicAdd(a.stats + 1, (unsigned long long)nt);
        }
    }
}

// ------------------------------------------------------------------------------------------------

template <bool MULTI>
__global__ __launch_bounds__(256) void k_ao_temporal(AOTemporalArgs a)
{
    __shared__ uint32_t s_mask[4][4][18];
    __shared__ MaskRows s_rows[4];
    __shared__ float    s_vpi[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wave;
    const bool tile_ok = tile < a.tiles_x * a.tiles_y;
    const int tx = tile_ok ? tile % a.tiles_x : 0, ty = (tile_ok ? tile / a.tiles_x : 0) + a.tile_y0;
    const int lx = lane & 7, ly = lane >> 3;
    const int x = tx * 8 + lx, y = ty * 8 + ly;
    if (threadIdx.x < 16) s_vpi[threadIdx.x] = a.vpi[threadIdx.x];
    // the window sums are integer work: the bit-sliced row patterns of the tolerance-mode kernel (mask_window.h) give the same counts as
    // one plane at a time (round-2 review item 7: 0-ulp structural change)
    build_mask_rows<true>(s_rows[wave], s_mask[wave], a.mask, MULTI ? a.spp : 1, a.mw, a.mh, tx, ty, a.y0, a.y1, lane, tile_ok);
    __syncthreads();   // s_vpi
    int sum, own;
    mask_window<MULTI>(s_rows[wave], lx, ly, sum, own);
    if (!tile_ok) return;
    const float mean = __fdiv_rn((float)sum, 289.0f * (float)a.spp);
    const bool  in_image = x < a.w && y < a.h && y >= a.y0 && y < a.y1;
    const bool  edge = x >= a.w || y >= a.h;   // no pixel, but the unguarded shader thread (:191-260) still votes (see k_shadows_temporal)
    bool        flag = false, apron_miss = false;
    if (in_image || edge)
    {
        const float d = edge ? 0.0f : a.depth.p[(size_t)y * a.w + x];
        float out = 1.0f, hlen = 0.0f;
        if (d != 1.0f)
        {
            const float ao = __fdiv_rn((float)own, (float)a.spp);
            ReprojIn in;
            in.x = x; in.y = y; in.depth = d; in.vpi = s_vpi;
            in.gb2 = a.gb2; in.gb3 = a.gb3; in.pgb2 = a.pgb2; in.pgb3 = a.pgb3; in.pdepth = a.pdepth;
            in.w = a.w; in.h = a.h;
            float hao, dummy[2];
            ImgRGBA16F none { nullptr, 0, 0, 0 };
            const bool success = reproject<true, false, false, ImgR16F>(in, a.hist, none, a.hist_len, &hao, dummy, hlen);
            apron_miss = in.apron_miss && in_image && y >= a.band_y0 && y < a.band_y1;
            hlen = min2(32.0f, success ? hlen + 1.0f : 1.0f);
            if (success)
            {
                float sv = max2(mean - mean * mean, 0.0f);
                float sd = hr_sqrt(sv);
                hao      = clamp1(hao, mean - 0.5f * sd, mean + 0.5f * sd);
            }
            const float al = success ? max2(a.alpha, __fdiv_rn(1.0f, hlen)) : 1.0f;
            out = mix1(hao, ao, al);
        }
        if (in_image)
        {
            a.out[(size_t)y * a.w + x]     = f2h(out);
            a.out_len[(size_t)y * a.w + x] = f2h(hlen);
        }
        flag = out < 1.0f;
    }
    if (a.apron_flag && __ballot(apron_miss) && lane == 0) atomicOr(a.apron_flag, 1u);
    const unsigned long long any = __ballot(flag);
    if (lane == 0) a.tile_class[(size_t)ty * a.tiles_x + tx] = any ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------

HR_DEV float linear_eye_depth(float z, const float* zbp) { return __fdiv_rn(1.0f, zbp[2] * z + zbp[3]); }

// common.glsl:160-165
HR_DEV float gaussian_weight(float offset, float deviation)
{
    float w = __fdiv_rn(1.0f, hr_sqrt(2.0f * HR_M_PI * deviation * deviation));
    return w * det_exp(__fdiv_rn(-(offset * offset), 2.0f * deviation * deviation));
}

#define AO_MAX_BLUR_RADIUS 32
// RADIUS == 4 (the reference default, ray_traced_ao.h): unrolled, loads up front, weights computed back to back;
// RADIUS < 0: run-time radius.
template <int RADIUS>
__global__ __launch_bounds__(256) void k_ao_blur(AOBlurArgs a)
{
    // gaussian_weight(i, radius / 1.5) depends on the tap index only: one lane per tap evaluates it once per workgroup
    // (two divisions, a square root and an exp) instead of every pixel re-deriving all of them
    __shared__ float s_gauss[2 * AO_MAX_BLUR_RADIUS + 1];
    {
        const int i = (int)threadIdx.x - a.radius;
        if ((int)threadIdx.x <= 2 * a.radius) s_gauss[threadIdx.x] = gaussian_weight((float)i, __fdiv_rn((float)a.radius, 1.5f));
    }
    __syncthreads();
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = a.y0 + blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= a.w || y >= a.y1) return;
    const size_t o = (size_t)y * a.w + x;
    const uint16_t one = f2h(1.0f);
    if (!a.tile_class[(size_t)(y >> 3) * a.tiles_x + (x >> 3)]) { a.out[o] = one; return; } // cleared image (:1048-1055)
    const float d = a.depth.p[o];
    if (d == 1.0f) { a.out[o] = one; return; }
    float total_ao = h2f(a.in.p[o]), total_w = 1.0f;
    const float cd = linear_eye_depth(d, a.zbp);
    const uint2 g2 = a.gb2.p[o];
    const f3    cn = oct_decode(h2f_lo(g2.x), h2f_hi(g2.x));
    if constexpr (RADIUS > 0)
    {
        constexpr int N = 2 * RADIUS;
        float t_d[N], t_a[N];
        uint2 t_g[N];
#pragma unroll
        for (int t = 0; t < N; t++)
        {
            const int i = t < RADIUS ? t - RADIUS : t - RADIUS + 1;
            const int sx = x + a.dx * i, sy = y + a.dy * i;
            t_d[t] = a.depth.fetch(sx, sy); t_a[t] = a.in.fetch(sx, sy); t_g[t] = a.gb2.raw(sx, sy);
        }
        float w8[N];
#pragma unroll
        for (int t = 0; t < N; t++)
        {
            const int   i  = t < RADIUS ? t - RADIUS : t - RADIUS + 1;
            const float sd = linear_eye_depth(t_d[t], a.zbp);
            const f3    sn = oct_decode(h2f_lo(t_g[t].x), h2f_hi(t_g[t].x));
            const float wZ = det_exp(__fdiv_rn(-fabsf(cd - sd), 1.0f));
            const float wN = det_pow_auto(clamp1(dot3(cn, sn), 0.0f, 1.0f), 32.0f);
            w8[t] = s_gauss[i + RADIUS] * (det_exp((0.0f - 1.0f) - max2(wZ, 0.0f)) * wN);
        }
#pragma unroll
        for (int t = 0; t < N; t++)
        {
            total_ao += w8[t] * t_a[t];
            total_w += w8[t];
        }
    }
    else
    {
        for (int i = -a.radius; i <= a.radius; i++)
        {
            if (i == 0) continue;
            const int   sx = x + a.dx * i, sy = y + a.dy * i;
            const float sd = linear_eye_depth(a.depth.fetch(sx, sy), a.zbp);
            const float sa = a.in.fetch(sx, sy);
            const uint2 s2 = a.gb2.raw(sx, sy);
            const f3    sn = oct_decode(h2f_lo(s2.x), h2f_hi(s2.x));
            float w = s_gauss[i + a.radius];
            const float wZ = det_exp(__fdiv_rn(-fabsf(cd - sd), 1.0f));
            const float wN = det_pow_auto(clamp1(dot3(cn, sn), 0.0f, 1.0f), 32.0f);
            w = w * (det_exp((0.0f - 1.0f) - max2(wZ, 0.0f)) * wN);
            total_ao += w * sa;
            total_w += w;
        }
    }
    a.out[o] = f2h(__fdiv_rn(total_ao, max2(total_w, 0.0001f)));
}

// ------------------------------------------------------------------------------------------------
struct hr_ao
{
    hr_ctx* ctx = nullptr;
    int     full_w = 0, full_h = 0, w = 0, h = 0, scale = 0, y0 = 0, y1 = 0, band_y0 = 0, band_y1 = 0;
    int     mw = 0, mh = 0, tiles_x = 0, tiles_y = 0, max_spp = 4;
    DevBuf  mask, color[2], length[2], blur[2], upsample, tile_class, counters, ray_slots, geo[2];
    bool    first_frame = true, last_denoise = true, want_stats = false;
    bool    fuse = true;   // tolerance mode: both blur passes in one launch (developer A/B switch HR_FUSE=0, read once at create)
    int     last_pp = 0;
    StageProfiler prof;
    hipStream_t   last_stream = nullptr;
    // Tolerance mode: the temporal kernel also writes an 8-byte record per pixel {oct normal, mesh id | AO} — copies of the current
    // G-buffer's words and of its own output — into geo[geo_parity].  When the caller hands back, as in->prev, the images it passed as
    // in->cur in the previous call (the reference's ping-pong, g_buffer.cpp:208-211) and alternates ping_pong, those records ARE the
    // previous G-buffer + the AO history, and the next frame reprojects from them: 2 images + the history length, 9 gathers per pixel
    // instead of 5 images, 17 gathers with half of every G-buffer line unused.
    bool          geo_history = true;   // developer A/B switch HR_GEO_HISTORY=0 (read once at create)
    bool          geo_valid = false;
    bool          dbg_require_geo = false;   // HR_DEBUG_REQUIRE_GEO (tests)
    int           geo_parity = 0, geo_pp = -1;
    const void*   geo_gb2 = nullptr;
    const void*   geo_gb3 = nullptr;
    TileOrder     tile_order;           // heaviest-first launch order of the trace kernel (tile_order.h)
    // entry-node table of the trace kernel (AoEntryGrid above): built for (scene, ray_length), rebuilt in-stream when either changes
    DevBuf        entry_grid;
    bool          grid_enabled = true;  // developer A/B switch HR_AO_ENTRY_GRID=0 (read once at create)
    uint64_t      grid_scene = 0;       // hr_scene::uid the table was built for (0: none)
    uint64_t      grid_epoch = 0;       // ... and its geometry_epoch (an instanced scene moves: hr_scene_update_instances)
    float         grid_ray_length = -1.0f, grid_lo[3] = { 0, 0, 0 }, grid_c = 0.0f;
    int           grid_n[3] = { 0, 0, 0 };
};

bool hr::profiling_enabled(const hr_ao* p) { return p && p->prof.enabled; }

extern "C" {

void hr_ao_default_params(hr_ao_params* p)
{
    p->denoise = 1; p->ray_length = 7.0f; p->bias = 0.3f; p->alpha = 0.01f; p->blur_radius = 4; p->power = 1.2f; p->spp = 1; p->exact = 1;
}

hr_status hr_ao_create(hr_ctx* ctx, int32_t full_width, int32_t full_height, hr_scale scale, const hr_band* band, hr_ao** out)
{
    HR_CHECK_ARG(ctx && out && full_width > 0 && full_height > 0 && (int)scale >= 0 && (int)scale <= 2);
    HR_HIP(hipSetDevice(ctx->device));
    hr_ao* p = new hr_ao();
    p->ctx = ctx; p->full_w = full_width; p->full_h = full_height; p->scale = (int)scale;
    if (const char* e = getenv("HR_FUSE")) p->fuse = atoi(e) != 0;
    if (const char* e = getenv("HR_GEO_HISTORY")) p->geo_history = atoi(e) != 0;
    if (const char* e = getenv("HR_DEBUG_REQUIRE_GEO")) p->dbg_require_geo = atoi(e) != 0;   // test switch, see hr_shadows
    if (const char* e = getenv("HR_AO_ENTRY_GRID")) p->grid_enabled = atoi(e) != 0;
    if (const char* e = getenv("HR_TILE_ORDER")) p->tile_order.enabled = atoi(e) != 0;
    p->tile_order.tag = "ao";
    p->w = full_width >> (int)scale; p->h = full_height >> (int)scale;
    p->y0 = 0; p->y1 = p->h;
    if (band && band->band_y1 > band->band_y0)
    {
        p->band_y0 = band->band_y0; p->band_y1 = band->band_y1;
        p->y0 = band->band_y0 - band->halo < 0 ? 0 : band->band_y0 - band->halo;
        p->y1 = band->band_y1 + band->halo > p->h ? p->h : band->band_y1 + band->halo;
        if ((p->y0 & 7) || ((p->y1 & 7) && p->y1 != p->h)) { set_last_error("band rows must be multiples of 8"); delete p; return HR_ERR_INVALID_ARG; }
    }
    p->mw = cdiv(p->w, 8); p->mh = cdiv(p->h, 4); p->tiles_x = cdiv(p->w, 8); p->tiles_y = cdiv(p->h, 8);
    const size_t px = (size_t)p->w * p->h;
    hr_status s;
#define A(buf, n) if ((s = p->buf.alloc(n)) != HR_OK) { delete p; return s; }
    A(mask, (size_t)p->max_spp * p->mw * p->mh * 4)
    A(color[0], px * 2) A(color[1], px * 2) A(length[0], px * 2) A(length[1], px * 2)
    A(blur[0], px * 2) A(blur[1], px * 2)
    A(upsample, (size_t)full_width * full_height * 2)
    A(tile_class, (size_t)p->tiles_x * p->tiles_y)
    A(counters, 64)
    A(ray_slots, (size_t)p->tiles_x * p->tiles_y * 4)
    if (p->geo_history) { A(geo[0], px * 8) A(geo[1], px * 8) }   // (round 5: bands too — see hr_ao_temporal)
    if (p->grid_enabled) { A(entry_grid, (size_t)kAoGridCells * 4) }
#undef A
    if ((s = p->tile_order.init(p->tiles_x * (cdiv(p->y1, 8) - p->y0 / 8))) != HR_OK) { delete p; return s; }
    HR_HIP(hipMemset(p->counters.p, 0, 64));
    HR_HIP(hipMemset(p->ray_slots.p, 0, p->ray_slots.bytes));
    HR_HIP(hipMemset(p->mask.p, 0, p->mask.bytes));
    *out = p;
    return HR_OK;
}

hr_status hr_ao_destroy(hr_ao* p)
{
    if (!p) return HR_OK;
    (void)hipSetDevice(p->ctx->device);
    (void)hipDeviceSynchronize();
    delete p;
    return HR_OK;
}
// Also forgets the entry-node table: it is marked built when its kernel is ENQUEUED, and this is the path a frame whose capture / launch
// failed takes (frame.hip reset_after_failed_frame) — the table may never have been written.
hr_status hr_ao_reset_history(hr_ao* p)
{
    HR_CHECK_ARG(p);
    p->first_frame = true; p->geo_valid = false; p->tile_order.invalidate();
    p->grid_scene = 0; p->grid_ray_length = -1.0f;
    return HR_OK;
}
hr_status hr_ao_history_apron_exceeded(hr_ao* p, int32_t* exceeded)   // see hr_shadows_history_apron_exceeded
{
    HR_CHECK_ARG(p && exceeded);
    uint32_t v = 0;
    HR_HIP(hipStreamSynchronize(p->last_stream));
    HR_HIP(hipMemcpy(&v, (char*)p->counters.p + 48, 4, hipMemcpyDeviceToHost));
    if (v) HR_HIP(hipMemset((char*)p->counters.p + 48, 0, 4));
    *exceeded = v ? 1 : 0;
    return HR_OK;
}

hr_status hr_ao_set_profiling(hr_ao* p, int32_t e) { HR_CHECK_ARG(p); p->prof.enabled = e != 0; return HR_OK; }
hr_status hr_ao_get_stage_times(hr_ao* p, hr_stage_times* out) { HR_CHECK_ARG(p && out); p->prof.collect(out); return HR_OK; }
hr_status hr_ao_ray_count(hr_ao* p, uint64_t* rays)
{
    HR_CHECK_ARG(p && rays);
    HR_HIP(hipStreamSynchronize(p->last_stream));
    std::vector<uint32_t> slots((size_t)p->tiles_x * p->tiles_y);
    HR_HIP(hipMemcpy(slots.data(), p->ray_slots.p, slots.size() * 4, hipMemcpyDeviceToHost));
    uint64_t total = 0;
    for (uint32_t v : slots) total += v;
    *rays = total;
    return HR_OK;
}

hr_status hr_ao_ray_trace(hr_ao* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_ao_params* prm, void* stream_)
{
    HR_CHECK_ARG(p && scene && in && prm && prm->spp >= 1 && prm->spp <= p->max_spp);
    HR_CHECK_ARG(in->cur.depth && in->cur.gb2 && in->cur.gb3 && in->cur.width == p->w && in->cur.height == p->h && in->sobol && in->scrambling_ranking);
    hipStream_t st = (hipStream_t)stream_;
    p->last_stream = st;
    if (p->first_frame)
    {
        // clear_images() (ray_traced_ao.cpp:828-860): the history slots that will be read start at 0
        HR_HIP(hipMemsetAsync(p->color[!in->ping_pong].p, 0, p->color[0].bytes, st));
        HR_HIP(hipMemsetAsync(p->length[!in->ping_pong].p, 0, p->length[0].bytes, st));
        p->first_frame = false;
    }
    if (p->want_stats) HR_HIP(hipMemsetAsync(p->counters.p, 0, 32, st));   // only the instrumented build accumulates into it
    AOTraceArgs a;
    for (int i = 0; i < 16; i++) a.vpi[i] = in->ubo.view_proj_inverse[i];
    a.depth = in->cur.depth; a.gb2 = (const uint2*)in->cur.gb2; a.sobol = in->sobol; a.sr = in->scrambling_ranking;
    a.mask = (uint32_t*)p->mask.p; a.ray_slots = (uint32_t*)p->ray_slots.p;
    a.nodes = (const Node8*)scene->nodes.p; a.tris = (const TriGPU*)scene->tris.p;
    a.stats = p->want_stats ? (unsigned long long*)((char*)p->counters.p + 16) : nullptr;
    a.w = p->w; a.h = p->h; a.y0 = p->y0; a.y1 = p->y1; a.mw = p->mw; a.mh = p->mh;
    a.tile_y0 = p->y0 / 8; a.tiles_x = p->tiles_x; a.tiles_y = cdiv(p->y1, 8) - a.tile_y0;
    a.bias = prm->bias; a.ray_length = prm->ray_length; a.num_frames = in->num_frames; a.spp = prm->spp;
    const int n_tiles = a.tiles_x * a.tiles_y;
    const uint64_t px = (uint64_t)p->w * (p->y1 - p->y0);
    { const hr_status fs = p->tile_order.flush(st); if (fs != HR_OK) return fs; }   // last launch's costs, if no temporal stage took them along
    a.order = p->tile_order.order_arg(n_tiles); a.cost = p->want_stats ? nullptr : p->tile_order.cost_arg(n_tiles);
    a.grid = nullptr;
    if (p->grid_enabled && p->entry_grid.p && prm->ray_length > 0.0f)
    {
        bool enqueued = false;
        if (p->grid_scene != scene->uid || p->grid_epoch != scene->geometry_epoch || p->grid_ray_length != prm->ray_length)
        {
            // cells of half the ray length, no finer than 1/256 of the longest extent, coarsened until the table fits its fixed capacity
            const float* lo = scene->grid_lo; const float* hi = scene->grid_hi;   // no read-back: conservative bounds for an instanced scene
            const float ext[3] = { hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2] };
            const float longest = ext[0] > ext[1] ? (ext[0] > ext[2] ? ext[0] : ext[2]) : (ext[1] > ext[2] ? ext[1] : ext[2]);
            float c = 0.5f * prm->ray_length;
            if (c < longest / 256.0f) c = longest / 256.0f;
            if (!(c > 0.0f) || !(longest < 1e30f)) c = 0.0f;
            int n[3] = { 0, 0, 0 };
            for (int it = 0; it < 64 && c > 0.0f; it++)
            {
                bool fits = true;
                for (int k = 0; k < 3; k++) { const double q = std::floor((double)ext[k] / c) + 1.0; n[k] = q < 1.0 ? 1 : (q > 4096.0 ? 4096 : (int)q); if (q > 4096.0) fits = false; }
                if (fits && (long long)n[0] * n[1] * n[2] <= (long long)kAoGridCells) break;
                c *= 1.26f;
            }
            p->grid_scene = 0;
            if (c > 0.0f && (long long)n[0] * n[1] * n[2] <= (long long)kAoGridCells)
            {
                const int cells = n[0] * n[1] * n[2];
                hipLaunchKernelGGL(k_ao_entry_grid, dim3(cdiv(cells, 256)), dim3(256), 0, st, (const Node8*)scene->nodes.p, (uint32_t*)p->entry_grid.p,
                                   lo[0], lo[1], lo[2], c, n[0], n[1], n[2], prm->ray_length * 1.0001f + 1e-4f);
                HR_HIP(hipGetLastError());
                enqueued = true;
                p->grid_scene = scene->uid; p->grid_epoch = scene->geometry_epoch; p->grid_ray_length = prm->ray_length; p->grid_c = c;
                for (int k = 0; k < 3; k++) { p->grid_lo[k] = lo[k]; p->grid_n[k] = n[k]; }
            }
        }
        // Under stream capture (HR_FRAME_GRAPH) the kernel node is recorded EVERY frame — over zero cells when the table is up to date — so that a
        // frame that rebuilds it (an animated ray_length) has the topology of the frames around it and hipGraphExecUpdate keeps the instantiated
        // graph (the extra node used to force the synchronise + re-instantiate path).
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (!enqueued && st && hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusActive)
        {
            hipLaunchKernelGGL(k_ao_entry_grid, dim3(1), dim3(256), 0, st, (const Node8*)scene->nodes.p, (uint32_t*)p->entry_grid.p, 0.0f, 0.0f, 0.0f, 1.0f, 0, 0, 0, 0.0f);
            HR_HIP(hipGetLastError());
        }
        if (p->grid_scene == scene->uid)
        {
            a.grid = (const uint32_t*)p->entry_grid.p; a.grid_inv_c = 1.0f / p->grid_c;
            for (int k = 0; k < 3; k++) { a.grid_lo[k] = p->grid_lo[k]; a.grid_n[k] = p->grid_n[k]; }
        }
    }
    if (p->want_stats)
    {
        hipLaunchKernelGGL(k_ao_trace<true>, dim3(cdiv(n_tiles, AO_TRACE_WAVES)), dim3(64 * AO_TRACE_WAVES), 0, st, a);
        HR_HIP(hipGetLastError());
        return HR_OK;
    }
    int ev = p->prof.begin("ray_trace", st, px * 12 + px * prm->spp / 8);
    hipLaunchKernelGGL(k_ao_trace<false>, dim3(cdiv(n_tiles, AO_TRACE_WAVES)), dim3(64 * AO_TRACE_WAVES), 0, st, a);
    p->prof.end(ev, st);
    HR_HIP(hipGetLastError());
    if (a.cost)
    {
        const hr_status os = p->tile_order.traced(n_tiles, st);
        if (os != HR_OK) return os;
    }
    return HR_OK;
}

hr_status hr_ao_launch_order(hr_ao* p, uint32_t* out, int32_t* n_tiles)
{
    HR_CHECK_ARG(p);
    return p->tile_order.read(out, n_tiles, p->last_stream);
}

hr_status hr_ao_trace_stats(hr_ao* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_ao_params* prm, uint64_t* out3, void* stream)
{
    HR_CHECK_ARG(p && out3);
    p->want_stats = true;
    hr_status s = hr_ao_ray_trace(p, scene, in, prm, stream);
    p->want_stats = false;
    if (s != HR_OK) return s;
    HR_HIP(hipStreamSynchronize((hipStream_t)stream));
    uint64_t host[4];
    HR_HIP(hipMemcpy(host, p->counters.p, 32, hipMemcpyDeviceToHost));
    hr_status rs = hr_ao_ray_count(p, &out3[0]);
    if (rs != HR_OK) return rs;
    out3[1] = host[2]; out3[2] = host[3];
    return HR_OK;
}

hr_status hr_ao_temporal(hr_ao* p, const hr_frame_inputs* in, const hr_ao_params* prm, void* stream_)
{
    HR_CHECK_ARG(p && in && prm && prm->spp >= 1 && prm->spp <= p->max_spp);
    HR_CHECK_ARG(in->cur.depth && in->cur.gb2 && in->cur.gb3 && in->prev.depth && in->prev.gb2 && in->prev.gb3 && in->cur.width == p->w && in->prev.width == p->w);
    hipStream_t st = (hipStream_t)stream_;
    p->last_stream = st;
    const int pp = in->ping_pong ? 1 : 0, w = p->w, y0 = p->y0, y1 = p->y1;
    AOTemporalArgs a;
    for (int i = 0; i < 16; i++) a.vpi[i] = in->ubo.view_proj_inverse[i];
    a.mask = (const uint32_t*)p->mask.p; a.mw = p->mw; a.mh = p->mh; a.spp = prm->spp;
    a.gb2  = ImgRGBA16F { (const uint2*)in->cur.gb2, w, y0, y1 };
    a.gb3  = ImgRGBA16F { (const uint2*)in->cur.gb3, w, y0, y1 };
    a.pgb2 = ImgRGBA16F { (const uint2*)in->prev.gb2, w, y0, y1 };
    a.pgb3 = ImgRGBA16F { (const uint2*)in->prev.gb3, w, y0, y1 };
    a.depth = ImgR32F { in->cur.depth, w, y0, y1 }; a.pdepth = ImgR32F { in->prev.depth, w, y0, y1 };
    a.hist = ImgR16F { (const uint16_t*)p->color[!pp].p, w, y0, y1 };
    a.hist_len = ImgR16F { (const uint16_t*)p->length[!pp].p, w, y0, y1 };
    a.out = (uint16_t*)p->color[pp].p; a.out_len = (uint16_t*)p->length[pp].p; a.tile_class = (uint8_t*)p->tile_class.p;
    a.apron_flag = (y0 > 0 || y1 < p->h) ? (uint32_t*)((char*)p->counters.p + 48) : nullptr;   // row bands only
    a.band_y0 = p->band_y0; a.band_y1 = p->band_y1;
    a.w = w; a.h = p->h; a.y0 = y0; a.y1 = y1;
    a.tile_y0 = y0 / 8; a.tiles_x = p->tiles_x; a.tiles_y = cdiv(y1, 8) - a.tile_y0;
    a.alpha = prm->alpha;
    a.geo_hist = nullptr; a.geo_out = nullptr; a.geo_band = 0;
    if (!prm->exact && p->geo[0].p)
    {
        // A band (round 5) reprojects from the records as well, but takes only the GEOMETRY from them (a.geo_band: GEO = 1, the AO history stays
        // with the R16F image): the rows next to a band boundary hold the neighbour's AO history after the per-frame exchange
        // (hr_ao_exchange_history), the record's AO half there is this GPU's own redundant result.  The geometry half is a copy of the
        // G-buffer either way, and a band computes — and so records — every row it reads history from (history_halo == halo for AO).
        a.geo_band = (y0 > 0 || y1 < p->h) ? 1 : 0;
        const bool had_records = p->geo_valid;
        if (p->geo_valid && !p->first_frame && pp != p->geo_pp && in->prev.gb2 == p->geo_gb2 && in->prev.gb3 == p->geo_gb3 && in->prev.gb2 != in->cur.gb2 && in->prev.gb3 != in->cur.gb3) a.geo_hist = p->geo[p->geo_parity].p;
        p->geo_parity ^= 1;
        a.geo_out = p->geo[p->geo_parity].p;
        p->geo_valid = true; p->geo_pp = pp; p->geo_gb2 = in->cur.gb2; p->geo_gb3 = in->cur.gb3;
        if (p->dbg_require_geo && had_records && !a.geo_hist) { hr::set_last_error("hr_ao_temporal: HR_DEBUG_REQUIRE_GEO is set and the record path was not taken"); return HR_ERR_INVALID_ARG; }
    }
    else p->geo_valid = false;
    p->last_pp = pp;
    const uint64_t px = (uint64_t)w * (y1 - y0);
    a.sort = TileSortArgs { nullptr, nullptr, 0, 0, 0, 0 };
    int ev = p->prof.begin("temporal_accumulation", st, px * 48 + px * prm->spp / 8);
    if (prm->exact && prm->spp > 1) hipLaunchKernelGGL(k_ao_temporal<true>, dim3(cdiv(a.tiles_x * a.tiles_y, 4)), dim3(256), 0, st, a);
    else if (prm->exact) hipLaunchKernelGGL(k_ao_temporal<false>, dim3(cdiv(a.tiles_x * a.tiles_y, 4)), dim3(256), 0, st, a);
    else
    {
        a.sort = p->tile_order.ride();   // the trace kernel's next launch order rides along (tile_order.h)
        launch_ao_temporal_fast(a, a.tiles_x * a.tiles_y, st);
    }
    p->prof.end(ev, st);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

hr_status hr_ao_blur(hr_ao* p, const hr_frame_inputs* in, const hr_ao_params* prm, int32_t pass, void* stream_)
{
    HR_CHECK_ARG(p && in && prm && (pass == 0 || pass == 1) && prm->blur_radius >= 1 && prm->blur_radius <= 16);
    hipStream_t st = (hipStream_t)stream_;
    p->last_stream = st;
    const int w = p->w, y0 = p->y0, y1 = p->y1;
    AOBlurArgs a;
    a.in = ImgR16F { (const uint16_t*)(pass == 0 ? p->color[p->last_pp].p : p->blur[0].p), w, y0, y1 };
    a.depth = ImgR32F { in->cur.depth, w, y0, y1 };
    a.gb2 = ImgRGBA16F { (const uint2*)in->cur.gb2, w, y0, y1 };
    a.tile_class = (const uint8_t*)p->tile_class.p;
    a.out = (uint16_t*)p->blur[pass].p;
    for (int i = 0; i < 4; i++) a.zbp[i] = in->z_buffer_params[i];
    a.w = w; a.h = p->h; a.y0 = y0; a.y1 = y1; a.tiles_x = p->tiles_x;
    a.dx = pass == 0 ? 1 : 0; a.dy = pass == 0 ? 0 : 1; a.radius = prm->blur_radius; // X first, then Y (quirk 8)
    const uint64_t px = (uint64_t)w * (y1 - y0);
    int ev = p->prof.begin(pass == 0 ? "blur_x" : "blur_y", st, px * 16);
    if (!prm->exact) launch_ao_blur_fast(a, st);
    else if (a.radius == 4) hipLaunchKernelGGL(k_ao_blur<4>, dim3(cdiv(w, 32), cdiv(y1 - y0, 8)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_ao_blur<-1>, dim3(cdiv(w, 32), cdiv(y1 - y0, 8)), dim3(256), 0, st, a);
    p->prof.end(ev, st);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

// tolerance mode, radius 4: X and Y pass in one launch, the X image stays in LDS (kf_ao_blur_xy); IMG_BLUR0 is not written
static hr_status ao_blur_xy(hr_ao* p, const hr_frame_inputs* in, const hr_ao_params* prm, void* stream_, bool* done)
{
    hipStream_t st = (hipStream_t)stream_;
    p->last_stream = st;
    const int w = p->w, y0 = p->y0, y1 = p->y1;
    AOBlurArgs a;
    a.in = ImgR16F { (const uint16_t*)p->color[p->last_pp].p, w, y0, y1 };
    a.depth = ImgR32F { in->cur.depth, w, y0, y1 };
    a.gb2 = ImgRGBA16F { (const uint2*)in->cur.gb2, w, y0, y1 };
    a.tile_class = (const uint8_t*)p->tile_class.p;
    a.out = (uint16_t*)p->blur[1].p;
    for (int i = 0; i < 4; i++) a.zbp[i] = in->z_buffer_params[i];
    a.w = w; a.h = p->h; a.y0 = y0; a.y1 = y1; a.tiles_x = p->tiles_x;
    a.dx = 1; a.dy = 1; a.radius = prm->blur_radius;
    const uint64_t px = (uint64_t)w * (y1 - y0);
    int ev = p->prof.begin("blur_xy", st, px * 16);
    *done = launch_ao_blur_xy_fast(a, st);
    p->prof.end(ev, st);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

hr_status hr_ao_upsample(hr_ao* p, const hr_frame_inputs* in, const hr_ao_params* prm, void* stream_)
{
    HR_CHECK_ARG(p && in && prm);
    if (p->scale == 0) return HR_OK;
    HR_CHECK_ARG(in->cur_full.gb2 && in->cur_full.gb3 && in->cur_full.width == p->full_w && in->cur_full.height == p->full_h);
    hipStream_t st = (hipStream_t)stream_;
    p->last_stream = st;
    UpsampleArgs a;
    a.W = p->full_w; a.H = p->full_h; a.w = p->w; a.h = p->h;
    a.G2 = (const uint2*)in->cur_full.gb2; a.G3 = (const uint2*)in->cur_full.gb3;
    a.g2 = (const uint2*)in->cur.gb2; a.g3 = (const uint2*)in->cur.gb3;
    a.in = p->blur[1].p; a.in_channels = 1; a.channels = 1; a.out = p->upsample.p; a.sky_value = 1.0f; a.power = prm->power;
    const uint64_t PX = (uint64_t)p->full_w * p->full_h, px = (uint64_t)p->w * p->h;
    int ev = p->prof.begin("upsample", st, PX * 18 + px * 18);
    if (prm->exact) launch_upsample(a, st);
    else launch_upsample_fast(a, st);
    p->prof.end(ev, st);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

// Everything of RayTracedAO::render after the ray trace (ray_traced_ao.cpp:102-111); see hr_shadows_denoise
hr_status hr_ao_denoise(hr_ao* p, const hr_frame_inputs* in, const hr_ao_params* prm, void* stream)
{
    HR_CHECK_ARG(p && in && prm);
    p->last_denoise = prm->denoise != 0;
    if (!prm->denoise) return HR_OK;
    hr_status s;
    HR_SCOPED_SAMPLE("Denoise");   // ray_traced_ao.cpp:909
    if ((s = hr_ao_temporal(p, in, prm, stream)) != HR_OK) return s;
    {
        HR_SCOPED_SAMPLE("Bilateral Blur");   // ray_traced_ao.cpp:1034
        bool fused = false;
        if (!prm->exact && p->fuse && prm->blur_radius == 4 && (s = ao_blur_xy(p, in, prm, stream, &fused)) != HR_OK) return s;
        if (!fused)
        {
            if ((s = hr_ao_blur(p, in, prm, 0, stream)) != HR_OK) return s;
            if ((s = hr_ao_blur(p, in, prm, 1, stream)) != HR_OK) return s;
        }
    }
    if (p->scale != 0 && (s = hr_ao_upsample(p, in, prm, stream)) != HR_OK) return s;
    return HR_OK;
}

// RayTracedAO::render (ray_traced_ao.cpp:98-112)
hr_status hr_ao_render(hr_ao* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_ao_params* prm, void* stream)
{
    HR_SCOPED_SAMPLE("Ambient Occlusion");
    HR_CHECK_ARG(p && scene && in && prm);
    HR_HIP(hipSetDevice(p->ctx->device));
    p->prof.begin_frame();
    p->last_denoise = prm->denoise != 0;
    hr_status s = hr_ao_ray_trace(p, scene, in, prm, stream);
    if (s != HR_OK) return s;
    return hr_ao_denoise(p, in, prm, stream);
}

static void fill_view(hr_image_view* v, void* data, int w, int h, int bpp, hr_format f)
{
    v->data = data; v->width = w; v->height = h; v->row_pitch_bytes = w * bpp; v->format = f;
}

// 0 mask (plane 0..spp-1 stacked vertically), 1/2 AO colour[0/1], 3/4 history length[0/1], 5/6 blur[0/1], 7 upsample, 8 tile classes
hr_status hr_ao_image(hr_ao* p, int32_t which, hr_image_view* v)
{
    HR_CHECK_ARG(p && v);
    switch (which)
    {
        case 0: fill_view(v, p->mask.p, p->mw, p->mh * p->max_spp, 4, HR_FORMAT_R32_UINT); break;
        case 1: fill_view(v, p->color[0].p, p->w, p->h, 2, HR_FORMAT_R16F); break;
        case 2: fill_view(v, p->color[1].p, p->w, p->h, 2, HR_FORMAT_R16F); break;
        case 3: fill_view(v, p->length[0].p, p->w, p->h, 2, HR_FORMAT_R16F); break;
        case 4: fill_view(v, p->length[1].p, p->w, p->h, 2, HR_FORMAT_R16F); break;
        case 5: fill_view(v, p->blur[0].p, p->w, p->h, 2, HR_FORMAT_R16F); break;
        case 6: fill_view(v, p->blur[1].p, p->w, p->h, 2, HR_FORMAT_R16F); break;
        case 7: fill_view(v, p->upsample.p, p->full_w, p->full_h, 2, HR_FORMAT_R16F); break;
        case 8: fill_view(v, p->tile_class.p, p->tiles_x, p->tiles_y, 1, (hr_format)0); break;
        default: set_last_error("hr_ao_image: unknown image index"); return HR_ERR_INVALID_ARG;
    }
    return HR_OK;
}

// RayTracedAO::output_ds (ray_traced_ao.cpp:128-148)
hr_status hr_ao_output(hr_ao* p, hr_output_kind kind, hr_image_view* v)
{
    HR_CHECK_ARG(p && v);
    if (!p->last_denoise || kind == HR_OUTPUT_RAY_TRACE) return hr_ao_image(p, 0, v);
    if (kind == HR_OUTPUT_TEMPORAL_ACCUMULATION) return hr_ao_image(p, 1 + p->last_pp, v);
    if (kind == HR_OUTPUT_ATROUS || p->scale == 0) return hr_ao_image(p, 6, v);
    return hr_ao_image(p, 7, v);
}

} // extern "C"
