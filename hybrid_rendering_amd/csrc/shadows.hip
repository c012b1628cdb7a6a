// RayTracedShadows on MI355X — HIP replacement for src/ray_traced_shadows.{h,cpp} and
// src/shaders/shadows/*.  Stage map (reference file:line -> kernel):
//   ray_trace()              ray_traced_shadows.cpp:972-1011, shadows_ray_trace.comp:89-132   -> k_shadows_trace
//   reset_args()+temporal()  :1015-1090, shadows_denoise_reprojection.comp:196-293            -> k_shadows_temporal
//   a_trous_filter()         :1094-1215, shadows_denoise_atrous.comp:94-174 +
//                            shadows_denoise_copy_shadow_tiles.comp:32-36 (fused via tile class) -> k_shadows_atrous
//   upsample()               :1219-1255, shadows_upsample.comp:62-109                          -> k_upsample (upsample.h)
#include "hr_internal.h"
#include "reproject.h"
#include "traverse.h"
#include "shading.h"
#include "upsample.h"
#include "pass_args.h"
#include "tile_order.h"

using namespace hr;

// ------------------------------------------------------------------------------------------------
struct TraceArgs
{
    float           vpi[16];
    hr_light        light;
    const float*    depth;
    const uint2*    gb2;
    const uint8_t*  sobol;
    const uint8_t*  sr;
    uint32_t*       mask;
    uint16_t*       ray_slots;  // rays fired per 8x8 tile (no shared atomic counter: one contended address serialised ~13 ns per wave)
    const Node8*    nodes;
    const TriGPU*   tris;
    unsigned long long* stats; // nullable: [0] nodes visited, [1] triangles tested
    unsigned long long* timeline; // nullable (HR_DEBUG_TIMELINE): per tile {start, end} in 100 MHz ticks + hw id
    int             w, h;      // pass image (full frame)
    int             y0, y1;    // resident rows
    int             mw;        // mask words per row
    int             tiles_x, tiles_y, tile_y0, debug_skip_traversal, debug_only_tx, debug_only_ty;
    float           bias;
    uint32_t        num_frames;
    uint32_t*       occluder;  // nullable: per pixel, the index (into tris) of the triangle that occluded its ray last frame (see k_shadows_trace)
    uint32_t        n_tri_refs;
    const uint32_t* order;     // nullable: launch slot -> tile, heaviest tiles of the last frame first (tile_order.h)
    uint16_t*       cost;      // nullable: per tile, how long its wave lived (100 MHz ticks)
};

// One wave = one 8x8 pixel tile = two 8x4 mask words; lane l -> pixel (l & 7, l >> 3), so the
// wave ballot IS the packed mask (bit y*8+x of shadows_ray_trace.comp:126).
#define TRACE_WAVES 1 // waves (8x8 tiles) per workgroup: 1 lets the dispatcher back-fill a CU wave by wave — tile
                      // costs differ by >10x, and with 4-wave groups the finished waves' slots idle until the slowest ends
#ifndef SHADOWS_TRACE_EU
#define SHADOWS_TRACE_EU 1   // minimum waves per SIMD the register allocator must leave room for (A/B: see docs/EXPERIMENTS.md R4.3)
#endif
template <bool STATS>
__global__ __launch_bounds__(64 * TRACE_WAVES, SHADOWS_TRACE_EU) void k_shadows_trace(TraceArgs a)
{
    __shared__ uint32_t s_stack[TRACE_WAVES][HR_STACK_ENTRIES * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int launch_slot = blockIdx.x * TRACE_WAVES + wave;
    if (launch_slot >= a.tiles_x * a.tiles_y) return;
    // Tile order: row-major, or last frame's heaviest tiles first (tile_order.h).  Stride permutations lose BVH locality in L2.
    const int slot = a.order ? (int)a.order[launch_slot] : launch_slot;
    const int tx = slot % a.tiles_x, ty_local = slot / a.tiles_x, ty = ty_local + a.tile_y0;
    if (a.debug_only_tx >= 0 && (tx != a.debug_only_tx || ty != a.debug_only_ty)) return;
    unsigned long long t_begin = 0;
    if (a.timeline || a.cost) t_begin = wall_clock64();
    const int x = tx * 8 + (lane & 7), y = ty * 8 + (lane >> 3);
    bool      lit = false, fired = false;
    uint32_t  nn = 0, nt = 0, wave_max = 0;
    // depth, normal and blue-noise texel of the pixel in ONE round trip (a lane outside the image reads a valid address and
    // ignores the values) instead of depth -> normal -> noise one after the other
    const int    kind = trace_lane_kind(x, y, a.w, a.h, a.y0, a.y1);
    const size_t pix  = kind == 1 ? (size_t)y * a.w + x : (size_t)a.y0 * a.w;
    const float    d_pre  = a.depth[pix];
    const uint2    g2_pre = a.gb2[pix];
    const uint32_t bn_pre = blue_noise_texel(x, y, a.sr);
    const uint32_t occ_pre = a.occluder ? a.occluder[pix] : 0xffffffffu;
    if (kind)
    {
        const float d = kind == 1 ? d_pre : 0.0f;      // edge thread: out-of-image fetches read 0
        if (d != 1.0f)
        {
            const float tu = __fdiv_rn((float)x + 0.5f, (float)a.w), tv = __fdiv_rn((float)y + 0.5f, (float)a.h);
            const f3    P  = world_pos_from_depth(tu, tv, d, a.vpi);
            const uint2 g2 = kind == 1 ? g2_pre : make_uint2(0u, 0u);
            const f3    N  = oct_decode(h2f_lo(g2.x), h2f_hi(g2.x));
            const f3    ro = add3(P, scale3(N, a.bias));
            const float r0 = sample_blue_noise_t(bn_pre, (int)a.num_frames, 0, a.sobol);
            const float r1 = sample_blue_noise_t(bn_pre, (int)a.num_frames, 1, a.sobol);
            f3    Wi;
            float t_max, att;
            fetch_light_shadow(a.light, P, N, r0, r1, Wi, t_max, att);
            if (att > 0.0f)
            {
                fired = true;
                // Occluder cache: "is ANY triangle hit in (t_min, t_max)" is a pure function of the geometry, so testing one particular
                // triangle FIRST and answering "occluded" when it is hit cannot change the mask — and the triangle that shadowed this
                // pixel last frame (static light, camera moving a fraction of a pixel) shadows it again almost always.  66 % of the bench
                // frame's shadow rays are occluded; a tile whose lanes are all answered by their cached triangle skips the walk, and in
                // mixed tiles the occluded lanes (the ones that would otherwise walk until their first hit) drop out of the wave's
                // longest-lane race.  A stale or foreign index is harmless: the CURRENT scene's triangle at that index is tested.
                bool occluded = false;
                uint32_t hit_tri = 0xffffffffu;
                if (a.occluder && kind == 1 && !a.debug_skip_traversal)
                {
                    const uint32_t c = occ_pre;
                    if (c < a.n_tri_refs)
                    {
                        const RayPre rp = ray_prepare(ro, Wi);
                        float t, u, v;
                        occluded = ray_tri_raw<false>(rp, load_tri_raw(a.tris, c), 0.01f, t_max, t, u, v);
                        if (STATS) nt++;
                        hit_tri = c;
                    }
                }
                if (a.debug_skip_traversal) lit = (ro.x + Wi.y > -1e30f);
                else if (!occluded)
                {
                    hit_tri = 0xffffffffu;
                    occluded = trace_any<STATS>(a.nodes, a.tris, ro, Wi, 0.01f, t_max, s_stack[wave], lane, nn, nt, 0u, nullptr, &hit_tri);
                }
                if (!a.debug_skip_traversal) lit = !occluded;
                if (a.occluder && kind == 1 && hit_tri != occ_pre) a.occluder[pix] = hit_tri;
            }
        }
    }
    const unsigned long long bits = __ballot(lit);
    const unsigned long long fb   = __ballot(fired);
    if (STATS)
    {
        // wave reduction of the counters
        uint32_t mx = nn + nt; // per-lane traversal steps; the wave runs until its slowest lane is done
        for (int o = 32; o > 0; o >>= 1) { uint32_t t = __shfl_down(mx, o); mx = t > mx ? t : mx; }
        wave_max = mx;
        for (int o = 32; o > 0; o >>= 1) { nn += __shfl_down(nn, o); nt += __shfl_down(nt, o); }
    }
    if (lane == 0)
    {
        const int my = ty * 2;
        if (my * 4 >= a.y0 && my * 4 < a.y1) a.mask[(size_t)my * a.mw + tx] = (uint32_t)(bits & 0xffffffffull);
        if ((my + 1) * 4 >= a.y0 && (my + 1) * 4 < a.y1 && (my + 1) * 4 < a.h) a.mask[(size_t)(my + 1) * a.mw + tx] = (uint32_t)(bits >> 32);
        a.ray_slots[(size_t)ty * a.tiles_x + tx] = (uint16_t)__popcll(fb);
        if (a.cost)
        {
            const unsigned long long ticks = wall_clock64() - t_begin;
            a.cost[slot] = (uint16_t)(ticks > 65535ull ? 65535ull : ticks);
        }
        if (a.timeline)
        {
            const unsigned long long t_end = wall_clock64();
            uint32_t hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            uint32_t xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            a.timeline[(size_t)slot * 4 + 0] = t_begin;
            a.timeline[(size_t)slot * 4 + 1] = t_end;
            a.timeline[(size_t)slot * 4 + 2] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
            a.timeline[(size_t)slot * 4 + 3] = (unsigned long long)wave_max | ((unsigned long long)(nn + nt) << 32); // STATS build only
        }
        if (STATS && a.stats)
        {
            atomicAdd(a.stats + 0, (unsigned long long)nn);
            atomicAdd(a.stats + 1, (unsigned long long)nt);
            atomicAdd(a.stats + 2, (unsigned long long)wave_max);
        }
    }
}

#ifdef HR_DEV_PATHS   // A/B path that lost (13 % slower, docs/EXPERIMENTS.md 4.2); build with HR_CFLAGS=-DHR_DEV_PATHS, select with HR_TRACE_KERNEL=queue
// ------------------------------------------------------------------------------------------------
// Persistent-wave variant (HR_TRACE_KERNEL=queue; NOT the default — measured 13% slower on the bench scene although it
// raises the SIMD lane utilisation of the traversal loop from 0.50 to 0.83-0.89: mixing rays of several tiles and
// traversal stages in one wave multiplies the distinct BVH nodes a wave fetches per step, and the loop is bound by
// those vector-memory requests, not by idle lanes).  A workgroup owns PW_TILES consecutive 8x8 tiles.
//   phase 1  every wave generates the shadow rays of its tiles and appends the FIRED ones to a ray queue in LDS
//            (wave ballot + popcount prefix, one LDS atomic per wave);
//   phase 2  the waves drain the queue: a lane whose ray has terminated fetches the next ray as soon as
//            PW_REFILL lanes of its wave are idle (ballot / popcount prefix compaction), so the SIMD lanes stay
//            busy although only ~1/3 of the pixels fire a ray and traversal lengths differ by >10x;
//   phase 3  visibility bits collected in LDS (atomicOr) are written out as packed 8x4 mask words.
// The any-hit answer of a ray does not depend on which lane traces it, so the masks are bit-identical.
#define PW_TILES 4
#define PW_REFILL 12
template <bool STATS>
__global__ __launch_bounds__(256) void k_shadows_trace_pw(TraceArgs a)
{
    __shared__ float4   s_qa[PW_TILES * 64];   // origin.xyz, t_max
    __shared__ float4   s_qb[PW_TILES * 64];   // direction.xyz, pixel code (tile_local << 6 | bit)
    __shared__ uint32_t s_stack[4][HR_STACK_ENTRIES * 64];
    __shared__ uint32_t s_bits[PW_TILES][2];
    __shared__ int      s_head, s_tail;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_tiles = a.tiles_x * a.tiles_y;
    if (threadIdx.x < PW_TILES * 2) s_bits[threadIdx.x >> 1][threadIdx.x & 1] = 0u;
    if (threadIdx.x == 0) { s_head = 0; s_tail = 0; }
    __syncthreads();
    // ---- phase 1: ray generation -------------------------------------------------------------------------
    for (int r = 0; r < PW_TILES / 4; r++)
    {
        const int tl   = r * 4 + wave;
        const int tile = blockIdx.x * PW_TILES + tl;
        bool      fired = false;
        f3        ro = mk3(0, 0, 0), Wi = mk3(0, 0, 1);
        float     t_max = 0.0f;
        int       tx = 0, ty = 0;
        if (tile < n_tiles)
        {
            tx = tile % a.tiles_x; ty = tile / a.tiles_x + a.tile_y0;
            const int x = tx * 8 + (lane & 7), y = ty * 8 + (lane >> 3);
            const int kind = trace_lane_kind(x, y, a.w, a.h, a.y0, a.y1);
            if (kind)
            {
                const float d = kind == 1 ? a.depth[(size_t)y * a.w + x] : 0.0f;
                if (d != 1.0f)
                {
                    const float tu = __fdiv_rn((float)x + 0.5f, (float)a.w), tv = __fdiv_rn((float)y + 0.5f, (float)a.h);
                    const f3    P  = world_pos_from_depth(tu, tv, d, a.vpi);
                    const uint2 g2 = kind == 1 ? a.gb2[(size_t)y * a.w + x] : make_uint2(0u, 0u);
                    const f3    N  = oct_decode(h2f_lo(g2.x), h2f_hi(g2.x));
                    ro = add3(P, scale3(N, a.bias));
                    const float r0 = sample_blue_noise(x, y, (int)a.num_frames, 0, a.sobol, a.sr);
                    const float r1 = sample_blue_noise(x, y, (int)a.num_frames, 1, a.sobol, a.sr);
                    float att;
                    fetch_light_shadow(a.light, P, N, r0, r1, Wi, t_max, att);
                    fired = att > 0.0f;
                }
            }
        }
        const unsigned long long fb = __ballot(fired);
        if (fb)
        {
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_tail, __popcll(fb));
            base = __builtin_amdgcn_readfirstlane(base);
            if (fired)
            {
                const int q = base + __popcll(fb & ((1ull << lane) - 1ull));
                s_qa[q] = make_float4(ro.x, ro.y, ro.z, t_max);
                s_qb[q] = make_float4(Wi.x, Wi.y, Wi.z, __uint_as_float((uint32_t)(tl << 6 | lane)));
            }
        }
        if (lane == 0 && tile < n_tiles) a.ray_slots[(size_t)ty * a.tiles_x + tx] = (uint16_t)__popcll(fb);
    }
    __syncthreads();
    // ---- phase 2: drain the queue with refilling lanes ----------------------------------------------------------
    const int  tail = s_tail;
    uint32_t   spill_array[HR_SPILL_ENTRIES];
    AnyHitLane st;
    bool       has = false;
    uint32_t   code = 0, nn = 0, nt = 0, steps = 0;
    while (true)
    {
        const unsigned long long idle = __ballot(!has);
        const int n_idle = __popcll(idle);
        if (n_idle >= PW_REFILL || n_idle == 64)
        {
            int head = 0;
            if (lane == 0) head = (s_head < tail) ? atomicAdd(&s_head, n_idle) : tail;
            head = __builtin_amdgcn_readfirstlane(head);
            if (head >= tail && n_idle == 64) break;     // queue empty and nothing in flight
            if (!has)
            {
                const int q = head + __popcll(idle & ((1ull << lane) - 1ull));
                if (q < tail)
                {
                    const float4 qa = s_qa[q], qb = s_qb[q];
                    code = __float_as_uint(qb.w);
                    if (a.debug_skip_traversal) atomicOr(&s_bits[code >> 6][(code >> 5) & 1u], 1u << (code & 31u));
                    else
                    {
                        anyhit_begin(st, mk3(qa.x, qa.y, qa.z), mk3(qb.x, qb.y, qb.z), 0.01f, qa.w, s_stack[wave], lane, spill_array);
                        has = true;
                    }
                }
            }
            if (!__ballot(has)) { if (head >= tail) break; else continue; }
        }
        if (has)
        {
            const int res = anyhit_step<STATS>(st, a.nodes, a.tris, nn, nt);
            if (res == 2) atomicOr(&s_bits[code >> 6][(code >> 5) & 1u], 1u << (code & 31u));
            if (res != 0) has = false;
        }
        if (STATS) steps++;
    }
    __syncthreads();
    // ---- phase 3: packed mask words -----------------------------------------------------------------------------
    if (threadIdx.x < PW_TILES * 2)
    {
        const int tl = threadIdx.x >> 1, half = threadIdx.x & 1, tile = blockIdx.x * PW_TILES + tl;
        if (tile < n_tiles)
        {
            const int tx = tile % a.tiles_x, ty = tile / a.tiles_x + a.tile_y0, my = ty * 2 + half;
            if (my * 4 >= a.y0 && my * 4 < a.y1 && my * 4 < a.h) a.mask[(size_t)my * a.mw + tx] = s_bits[tl][half];
        }
    }
    if (STATS && a.stats)
    {
        uint32_t mx = steps;
        for (int o = 32; o > 0; o >>= 1) { uint32_t t = __shfl_down(mx, o); mx = t > mx ? t : mx; }
        for (int o = 32; o > 0; o >>= 1) { nn += __shfl_down(nn, o); nt += __shfl_down(nt, o); }
        if (lane == 0)
        {
            atomicAdd(a.stats + 0, (unsigned long long)nn);
            atomicAdd(a.stats + 1, (unsigned long long)nt);
            atomicAdd(a.stats + 2, (unsigned long long)mx);
        }
    }
}
#endif // HR_DEV_PATHS

// ------------------------------------------------------------------------------------------------

#ifndef TEMPORAL_WAVES
#define TEMPORAL_WAVES 4
#endif
__global__ __launch_bounds__(64 * TEMPORAL_WAVES) void k_shadows_temporal(TemporalArgs a)
{
    __shared__ uint32_t s_mask[TEMPORAL_WAVES][18];
    __shared__ uint32_t s_rows[TEMPORAL_WAVES][24];
    __shared__ float    s_vpi[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * TEMPORAL_WAVES + wave;
    const bool tile_ok = tile < a.tiles_x * a.tiles_y;
    const int tx = tile_ok ? tile % a.tiles_x : 0, ty = (tile_ok ? tile / a.tiles_x : 0) + a.tile_y0;
    if (threadIdx.x < 16) s_vpi[threadIdx.x] = a.vpi[threadIdx.x];
    if (tile_ok && lane < 18)
    {
        // populate_cache (:114-123): 3x6 masks around the tile; out-of-image masks read 0 (pinned)
        const int cx = tx - 1 + lane % 3, cy = ty * 2 - 2 + lane / 3;
        uint32_t  v  = 0u;
        if (cx >= 0 && cy >= 0 && cx < a.mw && cy < a.mh && cy * 4 >= a.y0 - 8 && cy * 4 < a.y1 + 8) v = a.mask[(size_t)cy * a.mw + cx];
        s_mask[wave][lane] = v;
    }
    __syncthreads();
    const int lx = lane & 7, ly = lane >> 3;
    const int x = tx * 8 + lx, y = ty * 8 + ly;

    // 17x17 box sum of visibility bits (neighborhood_mean :157-190): integer popcounts are exact.
    // The 24-bit row patterns of the 24 cached pixel rows are the same for every lane of the wave: lanes 0..23 build
    // one each (3 byte extracts) into LDS, then every lane popcounts its 17-row x 17-bit window.
    if (lane < 24)
    {
        const int m = lane >> 2, br = lane & 3;
        const uint32_t w0 = s_mask[wave][m * 3 + 0], w1 = s_mask[wave][m * 3 + 1], w2 = s_mask[wave][m * 3 + 2];
        s_rows[wave][lane] = ((w0 >> (br * 8)) & 0xffu) | (((w1 >> (br * 8)) & 0xffu) << 8) | (((w2 >> (br * 8)) & 0xffu) << 16);
    }
    __syncthreads();
    int sum = 0;
#pragma unroll
    for (int yy = 0; yy <= 16; yy++) sum += __popc((s_rows[wave][ly + yy] >> lx) & 0x1ffffu);
    if (!tile_ok) return;
    const float mean = __fdiv_rn((float)sum, 289.0f);

    const bool in_image = x < a.w && y < a.h && y >= a.y0 && y < a.y1;
    // A thread right of / below a ragged image has no pixel, but the shader has no bounds check (:196-293): it reads depth 0
    // and G-buffer 0 (pinned out-of-image fetch), runs the body (its stores are dropped) and still votes in
    // g_should_denoise — e.g. with the mask bit its twin in the ray-trace dispatch produced (device_math.h trace_lane_kind).
    const bool edge = x >= a.w || y >= a.h;
    float      out_v = 0.0f, out_var = 0.0f, m0 = 0.0f, m1 = 0.0f, hlen = 0.0f;
    bool       flag = false, apron_miss = false;
    if (in_image || edge)
    {
        const size_t pix = in_image ? (size_t)y * a.w + x : (size_t)a.y0 * a.w;
        const float d = edge ? 0.0f : a.depth.p[pix];
        // centre G-buffer texel: fetched and decoded ONCE, for the reprojection and for the a-trous iterations (nd)
        const uint2 cg2 = edge ? make_uint2(0u, 0u) : a.gb2.p[pix], cg3 = edge ? make_uint2(0u, 0u) : a.gb3.p[pix];
        const f3    cn  = oct_decode(h2f_lo(cg2.x), h2f_hi(cg2.x));
        if (d != 1.0f)
        {
            const int   cry = ly + 8, crx = lx + 8;
            const float visibility = (float)((s_mask[wave][(cry >> 2) * 3 + (crx >> 3)] >> ((cry & 3) * 8 + (crx & 7))) & 1u);
            ReprojIn in;
            in.x = x; in.y = y; in.depth = d; in.vpi = a.vpi; // kernel argument => scalar registers
            in.gb2 = a.gb2; in.gb3 = a.gb3; in.pgb2 = a.pgb2; in.pgb3 = a.pgb3; in.pdepth = a.pdepth;
            in.w = a.w; in.h = a.h;
            in.has_center = true; in.c2 = cg2; in.c3 = cg3; in.cur_n = cn;
            float hv, hm[2];
            ImgR16F none { nullptr, 0, 0, 0 };
            bool success = false;
            if (!a.debug_skip_reproject) success = reproject<true, true, false, ImgRG16F>(in, a.hist, a.hist_moments, none, &hv, hm, hlen);
            else { hv = 0.0f; hm[0] = hm[1] = 0.0f; hlen = 0.0f; }
            apron_miss = in.apron_miss && in_image && y >= a.band_y0 && y < a.band_y1;
            hlen = min2(32.0f, success ? hlen + 1.0f : 1.0f);
            if (success)
            {
                float sv = max2(mean - mean * mean, 0.0f);
                float sd = hr_sqrt(sv);
                hv       = clamp1(hv, mean - 0.5f * sd, mean + 0.5f * sd);
            }
            const float al = success ? max2(a.alpha, __fdiv_rn(1.0f, hlen)) : 1.0f;
            const float am = success ? max2(a.moments_alpha, __fdiv_rn(1.0f, hlen)) : 1.0f;
            m0      = visibility;
            m1      = m0 * m0;
            m0      = mix1(hm[0], m0, am);
            m1      = mix1(hm[1], m1, am);
            out_var = max2(0.0f, m1 - m0 * m0);
            out_v   = mix1(hv, visibility, al);
            flag    = out_v > 0.0f;
        }
        if (in_image)
        {
            a.out_moments[pix] = make_uint2(pack_h2(m0, m1), pack_h2(hlen, 0.0f));
            // the a-trous iterations read every pixel's normal 9x4 times: store the decoded one
            a.nd[pix]  = make_float4(cn.x, cn.y, cn.z, h2f_hi(cg3.y));
            a.out[pix] = pack_h2(out_v, out_var);
        }
    }
    if (a.apron_flag && __ballot(apron_miss) && lane == 0) atomicOr(a.apron_flag, 1u);
    // tile classification (:275-291): any lit pixel => the tile needs the à-trous filter
    const unsigned long long any = __ballot(flag);
    if (lane == 0) a.tile_class[(size_t)ty * a.tiles_x + tx] = any ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------

// edge_stopping.glsl:31-62 with NORMAL + LUMA weights.  FAST = the reference's default parameters
// (phi_normal == 32, sigma_depth == 1): x / 1.0f == x and pow(x, 32) is five squarings — same bits, fewer ops.
template <bool FAST>
HR_DEV float edge_weight(float cd, float sd, float phi_z, f3 cn, f3 sn, float phi_n, float cl, float sl, const DivBy& phi_l)
{
    const float dz = -fabsf(cd - sd);
    const float wZ = det_exp(FAST ? dz : __fdiv_rn(dz, phi_z));
    const float dn = clamp1(dot3(cn, sn), 0.0f, 1.0f);
    float wN;
    if (FAST) { float b = dn * dn; b = b * b; b = b * b; b = b * b; wN = b * b; } // det_powi(dn, 32): r = 1 * b^32
    else wN = det_pow_auto(dn, phi_n);
    const float wL = div_by(fabsf(cl - sl), phi_l);   // correctly rounded, denominator work shared by the taps of a pixel
    return det_exp((0.0f - max2(wL, 0.0f)) - max2(wZ, 0.0f)) * wN;
}

// RADIUS >= 0: compile-time filter radius (the reference default is 1: a fully unrolled 3x3 stencil with
// constant kernel weights); RADIUS < 0: run-time radius.
template <int RADIUS, bool FAST>
__global__ __launch_bounds__(256) void k_shadows_atrous(AtrousArgs a)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = a.y0 + blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= a.w || y >= a.y1) return;
    const size_t o = (size_t)y * a.w + x;
    if (!a.tile_class[(size_t)(y >> 3) * a.tiles_x + (x >> 3)])
    {
        a.out[o] = 0u; // shadows_denoise_copy_shadow_tiles.comp:35
        if (a.out2) a.out2[o] = 0u;
        return;
    }
    const uint32_t c  = a.in.p[o];
    // RADIUS == 1: issue every load of this pixel (centre, 3x3 variance taps, decoded normals, the 8 stencil taps)
    // before the first use, so the wave pays ONE memory round trip instead of three dependent ones.
    uint32_t t_in[8];
    float4   t_nd[8];
    bool     t_ok[8];
    if (RADIUS == 1)
    {
#pragma unroll
        for (int t = 0; t < 8; t++)
        {
            const int k = t < 4 ? t : t + 1, xx = k % 3 - 1, yy = k / 3 - 1;
            const int px = x + xx * a.step, py = y + yy * a.step;
            t_ok[t] = px >= 0 && py >= 0 && px < a.w && py < a.h;
            const bool res = t_ok[t] && py >= a.y0 && py < a.y1;
            t_in[t] = res ? a.in.p[(size_t)py * a.w + px] : 0u;
            t_nd[t] = res ? a.nd[(size_t)py * a.w + px] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    }
    const float    cv = h2f_lo(c);
    // compute_variance_center (:65-88)
    float var = 0.0f;
#pragma unroll
    for (int yy = -1; yy <= 1; yy++)
#pragma unroll
        for (int xx = -1; xx <= 1; xx++)
        {
            const float k = (xx == 0 ? (yy == 0 ? 0.25f : 0.125f) : (yy == 0 ? 0.125f : 0.0625f));
            var += h2f_hi(a.in.raw(x + xx, y + yy)) * k;
        }
    const float4 cnd = a.nd[o];
    const float center_depth = cnd.w;
    uint32_t    result;
    if (center_depth < 0.0f) result = c;
    else
    {
        const f3    cn    = mk3(cnd.x, cnd.y, cnd.z);
        const DivBy phi_v = div_prepare(a.phi_visibility * hr_sqrt(max2(0.0f, 1e-10f + var)));
        float sum_w = 1.0f, sum_v = cv, sum_var = h2f_hi(c);
        if (RADIUS == 1)
        {
            // Weights of all eight taps first, UNCONDITIONALLY (a tap outside the image was loaded as zeros: finite math,
            // weight unused) — the eight dependent exp -> divide -> exp chains then interleave instead of running one
            // after the other behind per-tap branches; measured 25.3 -> 22.8 us per iteration.  The accumulation below
            // keeps the reference's order (yy outer, xx inner) and skips the outside taps.
            float w8[8];
#pragma unroll
            for (int t = 0; t < 8; t++)
                w8[t] = edge_weight<FAST>(center_depth, t_nd[t].w, a.sigma_depth, cn, mk3(t_nd[t].x, t_nd[t].y, t_nd[t].z), a.phi_normal, cv, h2f_lo(t_in[t]), phi_v);
#pragma unroll
            for (int t = 0; t < 8; t++)
            {
                if (!t_ok[t]) continue;
                const int   k = t < 4 ? t : t + 1, xx = k % 3 - 1, yy = k / 3 - 1;
                const float kx = xx == 0 ? 1.0f : __fdiv_rn(2.0f, 3.0f), ky = yy == 0 ? 1.0f : __fdiv_rn(2.0f, 3.0f);
                const float sv = h2f_lo(t_in[t]);
                const float wv = w8[t] * (kx * ky);
                sum_w += wv;
                sum_v += wv * sv;
                sum_var += (wv * wv) * h2f_hi(t_in[t]);
            }
        }
        else
        {
        const int R = RADIUS >= 0 ? RADIUS : a.radius;
        for (int yy = -R; yy <= R; yy++)
            for (int xx = -R; xx <= R; xx++)
            {
                const int px = x + xx * a.step, py = y + yy * a.step;
                if (px < 0 || py < 0 || px >= a.w || py >= a.h || (xx == 0 && yy == 0)) continue;
                const int   axx = xx < 0 ? -xx : xx, ayy = yy < 0 ? -yy : yy;
                const float kx = axx == 0 ? 1.0f : (axx == 1 ? __fdiv_rn(2.0f, 3.0f) : __fdiv_rn(1.0f, 6.0f));
                const float ky = ayy == 0 ? 1.0f : (ayy == 1 ? __fdiv_rn(2.0f, 3.0f) : __fdiv_rn(1.0f, 6.0f));
                const uint32_t s  = a.in.raw(px, py);
                const float4   snd = (py >= a.y0 && py < a.y1) ? a.nd[(size_t)py * a.w + px] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                const float    sv = h2f_lo(s);
                const f3       sn = mk3(snd.x, snd.y, snd.z);
                const float    w  = edge_weight<FAST>(center_depth, snd.w, a.sigma_depth, cn, sn, a.phi_normal, cv, sv, phi_v);
                const float    wv = w * (kx * ky);
                sum_w += wv;
                sum_v += wv * sv;
                sum_var += (wv * wv) * h2f_hi(s);
            }
        }
        float ov = __fdiv_rn(sum_v, sum_w), ovar = __fdiv_rn(sum_var, sum_w * sum_w);
        if (a.power != 0.0f) ov = det_pow_auto(ov, a.power);
        result = pack_h2(ov, ovar);
    }
    a.out[o] = result;
    if (a.out2) a.out2[o] = result;
}

// ------------------------------------------------------------------------------------------------
struct hr_shadows
{
    hr_ctx* ctx = nullptr;
    int     full_w = 0, full_h = 0, w = 0, h = 0, scale = 0;
    int     y0 = 0, y1 = 0;           // resident rows (band + halo)
    int     band_y0 = 0, band_y1 = 0;
    int     ry0 = 0, ry1 = 0;         // rows whose history / G-buffer may be read (band + history halo)
    int     mw = 0, mh = 0, tiles_x = 0, tiles_y = 0;
    DevBuf  mask, temporal_out, moments[2], prev_image, atrous[2], upsample, tile_class, counters, nd, ray_slots, occluder;
    bool    occluder_cache = true;    // developer A/B switch HR_SHADOW_CACHE=0 (read once at create)
    bool    first_frame = true;
    int     read_idx = 0;             // ATrous::read_idx
    bool    last_denoise = true;
    int     last_ping_pong = 0;
    StageProfiler prof;
    hipStream_t   last_stream = nullptr;
    bool          want_stats = false;
    int           latched_exact = -1;       // arithmetic mode of the last temporal stage: the `nd` side image it wrote has a mode-specific layout
    bool          fuse = true;              // tolerance mode: a-trous iterations 0 + 1 in one launch (developer A/B switch HR_FUSE=0)
    bool          persistent_waves = false; // HR_TRACE_KERNEL=queue selects the persistent-wave ray-queue kernel (A/B measurements)
    // developer switches (tools/timeline.py, tools/passbench.py), read from the environment ONCE in hr_shadows_create — the
    // render path never calls getenv
    int           dbg_only_tx = -1, dbg_only_ty = -1;
    bool          dbg_skip_traversal = false, dbg_skip_reproject = false, dbg_timeline_stats = false;
    std::string   dbg_timeline;             // HR_DEBUG_TIMELINE=<file>
    uint64_t      last_wave_max_steps = 0; // sum over waves of the slowest lane's (node + triangle) steps
    // Tolerance mode: the temporal kernel writes an 8-byte geometry record per pixel {oct normal, mesh id | linear z} — copies of the
    // current G-buffer's words — for the a-trous taps; the two halves of `nd` alternate, so last frame's records are still there when
    // the next frame reprojects.  They stand for the caller's PREVIOUS G-buffer exactly when the caller hands back, as in->prev, the
    // images it passed as in->cur in the previous call (the reference's ping-pong, g_buffer.cpp:208-211): then the reprojection reads
    // them (4 images, 16 gathers per pixel, full cache lines) instead of prev GB2 / GB3 (5 images, 21 gathers, half of every line).
    // A previous G-buffer that ALIASES the current one (one buffer rewritten in place) is read as the caller passed it.
    bool          geo_history = true;       // developer A/B switch HR_GEO_HISTORY=0 (read once at create)
    bool          geo_valid = false;        // `nd` half geo_parity holds the records of the last frame this pass rendered
    bool          dbg_require_geo = false;  // HR_DEBUG_REQUIRE_GEO (tests): from the second frame on, a tolerance-mode temporal stage must take the record path
    int           geo_parity = 0;
    const void*   geo_gb2 = nullptr;        // in->cur.gb2 / gb3 of that frame
    const void*   geo_gb3 = nullptr;
    void*         nd_cur = nullptr;         // the side image of the frame in flight (a-trous stages)
    TileOrder     tile_order;               // heaviest-first launch order of the trace kernel (tile_order.h)
};

bool hr::profiling_enabled(const hr_shadows* p) { return p && p->prof.enabled; }

extern "C" {

void hr_shadows_default_params(hr_shadows_params* p)
{
    p->denoise = 1; p->bias = 0.5f; p->alpha = 0.01f; p->moments_alpha = 0.2f; p->phi_visibility = 10.0f;
    p->phi_normal = 32.0f; p->sigma_depth = 1.0f; p->power = 1.2f; p->radius = 1; p->filter_iterations = 4; p->feedback_iteration = 1;
    p->exact = 1;
}

hr_status hr_shadows_create(hr_ctx* ctx, int32_t full_width, int32_t full_height, hr_scale scale, const hr_band* band, hr_shadows** out)
{
    HR_CHECK_ARG(ctx && out && full_width > 0 && full_height > 0 && (int)scale >= 0 && (int)scale <= 2);
    HR_HIP(hipSetDevice(ctx->device));
    hr_shadows* p = new hr_shadows();
    p->ctx = ctx; p->full_w = full_width; p->full_h = full_height; p->scale = (int)scale;
#ifdef HR_DEV_PATHS
    { const char* e = getenv("HR_TRACE_KERNEL"); p->persistent_waves = (e && std::string(e) == "queue"); }
#endif
    if (const char* e = getenv("HR_DEBUG_ONLY_TILE")) sscanf(e, "%d,%d", &p->dbg_only_tx, &p->dbg_only_ty);
    if (const char* e = getenv("HR_FUSE")) p->fuse = atoi(e) != 0;
    if (const char* e = getenv("HR_SHADOW_CACHE")) p->occluder_cache = atoi(e) != 0;
    if (const char* e = getenv("HR_GEO_HISTORY")) p->geo_history = atoi(e) != 0;
    if (const char* e = getenv("HR_DEBUG_REQUIRE_GEO")) p->dbg_require_geo = atoi(e) != 0;   // test switch: a temporal stage that cannot reproject from the records fails
    if (const char* e = getenv("HR_TILE_ORDER")) p->tile_order.enabled = atoi(e) != 0;
    p->tile_order.tag = "shadows";
    p->tile_order.min_spread = HR_ORDER_COHERENT_SPREAD;   // rays towards one light: coherent
    p->dbg_skip_traversal = getenv("HR_DEBUG_SKIP_TRAVERSAL") != nullptr;
    p->dbg_skip_reproject = getenv("HR_DEBUG_SKIP_REPROJECT") != nullptr;
    p->dbg_timeline_stats = getenv("HR_DEBUG_TIMELINE_STATS") != nullptr;
    if (const char* e = getenv("HR_DEBUG_TIMELINE")) p->dbg_timeline = e;
    // m_width = extent / 2^scale (ray_traced_shadows.cpp:80-83: float divide then truncation)
    p->w = full_width >> (int)scale; p->h = full_height >> (int)scale;
    p->y0 = 0; p->y1 = p->h; p->band_y0 = 0; p->band_y1 = p->h; p->ry0 = 0; p->ry1 = p->h;
    if (band && band->band_y1 > band->band_y0)
    {
        if (band->band_y0 < 0 || band->band_y1 > p->h || (band->band_y0 & 7) || ((band->band_y1 & 7) && band->band_y1 != p->h) || band->halo < 0 || (band->halo & 7))
        {
            set_last_error("band rows must be multiples of 8 inside the pass image; halo a multiple of 8");
            delete p;
            return HR_ERR_INVALID_ARG;
        }
        p->band_y0 = band->band_y0; p->band_y1 = band->band_y1;
        p->y0 = band->band_y0 - band->halo < 0 ? 0 : band->band_y0 - band->halo;
        p->y1 = band->band_y1 + band->halo > p->h ? p->h : band->band_y1 + band->halo;
        const int hh = band->history_halo > band->halo ? band->history_halo : band->halo;
        p->ry0 = band->band_y0 - hh < 0 ? 0 : band->band_y0 - hh;
        p->ry1 = band->band_y1 + hh > p->h ? p->h : band->band_y1 + hh;
    }
    p->mw = cdiv(p->w, 8); p->mh = cdiv(p->h, 4);
    p->tiles_x = cdiv(p->w, 8); p->tiles_y = cdiv(p->h, 8);
    const size_t px = (size_t)p->w * p->h; // full-frame sized (bands index with absolute rows)
    hr_status s;
#define A(buf, n) if ((s = p->buf.alloc(n)) != HR_OK) { delete p; return s; }
    A(mask, (size_t)p->mw * p->mh * 4)
    A(temporal_out, px * 4)
    A(moments[0], px * 8)
    A(moments[1], px * 8)
    A(prev_image, px * 4)
    A(atrous[0], px * 4)
    A(atrous[1], px * 4)
    A(upsample, (size_t)full_width * full_height * 2)
    A(tile_class, (size_t)p->tiles_x * p->tiles_y)
    A(nd, px * 16)
    A(counters, 64)
    A(ray_slots, (size_t)p->tiles_x * p->tiles_y * 2)
    A(occluder, px * 4)
#undef A
    if ((s = p->tile_order.init(p->tiles_x * (cdiv(p->y1, 8) - p->y0 / 8))) != HR_OK) { delete p; return s; }
    HR_HIP(hipMemset(p->occluder.p, 0xff, p->occluder.bytes));   // no cached occluder
    HR_HIP(hipMemset(p->counters.p, 0, 64));
    HR_HIP(hipMemset(p->ray_slots.p, 0, p->ray_slots.bytes));
    HR_HIP(hipMemset(p->mask.p, 0, p->mask.bytes));
    HR_HIP(hipMemset(p->tile_class.p, 0, p->tile_class.bytes));
    *out = p;
    return HR_OK;
}

hr_status hr_shadows_destroy(hr_shadows* p)
{
    if (!p) return HR_OK;
    (void)hipSetDevice(p->ctx->device);
    (void)hipDeviceSynchronize();
    delete p;
    return HR_OK;
}

hr_status hr_shadows_reset_history(hr_shadows* p)
{
    HR_CHECK_ARG(p);
    p->first_frame = true;
    p->geo_valid = false;
    p->tile_order.invalidate();   // costs of a frame that may never have run are not sorted (the launch list itself is always a permutation)
    return HR_OK;
}

// Row bands: did any history tap of the frames rendered since the last call fall on an image row this GPU does not hold (per-frame
// motion beyond hr_band.history_halo)?  Such taps were treated as disoccluded — the image stays valid but is no longer identical to
// the single-GPU one; the integrator widens history_halo (or re-creates the band) when this fires.  Synchronises the stream.
hr_status hr_shadows_history_apron_exceeded(hr_shadows* p, int32_t* exceeded)
{
    HR_CHECK_ARG(p && exceeded);
    uint32_t v = 0;
    HR_HIP(hipStreamSynchronize(p->last_stream));
    HR_HIP(hipMemcpy(&v, (char*)p->counters.p + 48, 4, hipMemcpyDeviceToHost));
    if (v) HR_HIP(hipMemset((char*)p->counters.p + 48, 0, 4));
    *exceeded = v ? 1 : 0;
    return HR_OK;
}

hr_status hr_shadows_set_profiling(hr_shadows* p, int32_t enable)
{
    HR_CHECK_ARG(p);
    p->prof.enabled = enable != 0;
    return HR_OK;
}

hr_status hr_shadows_get_stage_times(hr_shadows* p, hr_stage_times* out)
{
    HR_CHECK_ARG(p && out);
    p->prof.collect(out);
    return HR_OK;
}

hr_status hr_shadows_ray_count(hr_shadows* p, uint64_t* rays)
{
    HR_CHECK_ARG(p && rays);
    HR_HIP(hipStreamSynchronize(p->last_stream));
    std::vector<uint16_t> slots((size_t)p->tiles_x * p->tiles_y);
    HR_HIP(hipMemcpy(slots.data(), p->ray_slots.p, slots.size() * 2, hipMemcpyDeviceToHost));
    uint64_t total = 0;
    for (uint16_t v : slots) total += v;
    *rays = total;
    return HR_OK;
}

hr_status hr_shadows_tile_ray_counts(hr_shadows* p, uint16_t* out, int32_t* tiles_x, int32_t* tiles_y)
{
    HR_CHECK_ARG(p);
    if (tiles_x) *tiles_x = p->tiles_x;
    if (tiles_y) *tiles_y = p->tiles_y;
    if (!out) return HR_OK;
    HR_HIP(hipStreamSynchronize(p->last_stream));
    HR_HIP(hipMemcpy(out, p->ray_slots.p, (size_t)p->tiles_x * p->tiles_y * 2, hipMemcpyDeviceToHost));
    return HR_OK;
}

hr_status hr_shadows_launch_order(hr_shadows* p, uint32_t* out, int32_t* n_tiles)
{
    HR_CHECK_ARG(p);
    return p->tile_order.read(out, n_tiles, p->last_stream);
}

static hr_status check_inputs(const hr_shadows* p, const hr_frame_inputs* in, bool need_prev)
{
    HR_CHECK_ARG(in->cur.depth && in->cur.gb2 && in->cur.gb3);
    HR_CHECK_ARG(in->cur.width == p->w && in->cur.height == p->h);
    if (need_prev) HR_CHECK_ARG(in->prev.depth && in->prev.gb2 && in->prev.gb3 && in->prev.width == p->w && in->prev.height == p->h);
    return HR_OK;
}

hr_status hr_shadows_ray_trace(hr_shadows* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_shadows_params* prm, void* stream_)
{
    HR_CHECK_ARG(p && scene && in && prm);
    hr_status s = check_inputs(p, in, false);
    if (s != HR_OK) return s;
    HR_CHECK_ARG(in->sobol && in->scrambling_ranking);
    hipStream_t st = (hipStream_t)stream_;
    p->last_stream = st;
    // clear_images() (ray_traced_shadows.cpp:938-968): first frame zeroes the feedback image and the
    // history moments slot that will be read.
    if (p->first_frame)
    {
        HR_HIP(hipMemsetAsync(p->prev_image.p, 0, p->prev_image.bytes, st));
        HR_HIP(hipMemsetAsync(p->moments[!in->ping_pong].p, 0, p->moments[0].bytes, st));
        p->first_frame = false;
    }
    TraceArgs a;
    for (int i = 0; i < 16; i++) a.vpi[i] = in->ubo.view_proj_inverse[i];
    a.light = in->ubo.light;
    a.depth = in->cur.depth; a.gb2 = (const uint2*)in->cur.gb2;
    a.sobol = in->sobol; a.sr = in->scrambling_ranking;
    a.mask = (uint32_t*)p->mask.p; a.ray_slots = (uint16_t*)p->ray_slots.p;
    a.nodes = (const Node8*)scene->nodes.p; a.tris = (const TriGPU*)scene->tris.p;
    a.stats = nullptr;
    a.timeline = nullptr;
    a.w = p->w; a.h = p->h; a.y0 = p->y0; a.y1 = p->y1; a.mw = p->mw;
    a.tile_y0 = p->y0 / 8; a.tiles_x = p->tiles_x; a.tiles_y = cdiv(p->y1, 8) - a.tile_y0;
    a.bias = prm->bias; a.num_frames = in->num_frames;
    a.occluder = p->occluder_cache ? (uint32_t*)p->occluder.p : nullptr;
    a.n_tri_refs = (uint32_t)(scene->tris.bytes / sizeof(TriGPU));
    const int n_tiles = a.tiles_x * a.tiles_y;
    const int n_slots = n_tiles;
    a.debug_only_tx = p->dbg_only_tx; a.debug_only_ty = p->dbg_only_ty;
    a.debug_skip_traversal = p->dbg_skip_traversal ? 1 : 0;
    if ((s = p->tile_order.flush(st)) != HR_OK) return s;   // last launch's costs, if no temporal stage took them along
    a.order = p->tile_order.order_arg(n_tiles); a.cost = p->tile_order.cost_arg(n_tiles);
    const uint64_t px = (uint64_t)p->w * (p->y1 - p->y0);
    if (p->want_stats)
    {
        a.cost = nullptr;
        // instrumented build of the same kernel: counts node visits / triangle tests
        HR_HIP(hipMemsetAsync((char*)p->counters.p + 16, 0, 24, st));
        a.stats = (unsigned long long*)((char*)p->counters.p + 16);
#ifdef HR_DEV_PATHS
        if (p->persistent_waves) hipLaunchKernelGGL(k_shadows_trace_pw<true>, dim3(cdiv(n_tiles, PW_TILES)), dim3(256), 0, st, a);
        else
#endif
        hipLaunchKernelGGL(k_shadows_trace<true>, dim3(cdiv(n_slots, TRACE_WAVES)), dim3(64 * TRACE_WAVES), 0, st, a);
        HR_HIP(hipGetLastError());
        return HR_OK;
    }
    if (!p->dbg_timeline.empty())
    {
        const char* tl = p->dbg_timeline.c_str();
        // developer switch (tools/timeline.py): per-wave start/end ticks of ONE launch of the tile kernel, written to the file
        DevBuf buf;
        hr_status bs = buf.alloc((size_t)n_tiles * 32);
        if (bs != HR_OK) return bs;
        a.timeline = (unsigned long long*)buf.p;
        if (p->dbg_timeline_stats) hipLaunchKernelGGL(k_shadows_trace<true>, dim3(cdiv(n_slots, TRACE_WAVES)), dim3(64 * TRACE_WAVES), 0, st, a);
        else hipLaunchKernelGGL(k_shadows_trace<false>, dim3(cdiv(n_slots, TRACE_WAVES)), dim3(64 * TRACE_WAVES), 0, st, a);
        HR_HIP(hipStreamSynchronize(st));
        std::vector<unsigned long long> host((size_t)n_tiles * 4);
        HR_HIP(hipMemcpy(host.data(), buf.p, host.size() * 8, hipMemcpyDeviceToHost));
        if (FILE* f = fopen(tl, "wb")) { fwrite(host.data(), 8, host.size(), f); fclose(f); }
        a.timeline = nullptr;
    }
    int ev = p->prof.begin("ray_trace", st, px * 12 + px / 8);
#ifdef HR_DEV_PATHS
    if (p->persistent_waves) hipLaunchKernelGGL(k_shadows_trace_pw<false>, dim3(cdiv(n_tiles, PW_TILES)), dim3(256), 0, st, a);
    else
#endif
    hipLaunchKernelGGL(k_shadows_trace<false>, dim3(cdiv(n_slots, TRACE_WAVES)), dim3(64 * TRACE_WAVES), 0, st, a);
    p->prof.end(ev, st);
    HR_HIP(hipGetLastError());
    if (a.cost && !p->persistent_waves && (s = p->tile_order.traced(n_tiles, st)) != HR_OK) return s;
    return HR_OK;
}

hr_status hr_shadows_trace_divergence(hr_shadows* p, uint64_t* wave_max_steps)
{
    HR_CHECK_ARG(p && wave_max_steps);
    *wave_max_steps = p->last_wave_max_steps;
    return HR_OK;
}

static hr_status shadows_trace_stats_impl(hr_shadows* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_shadows_params* prm, uint64_t* out3, void* stream, bool walk_only)
{
    HR_CHECK_ARG(p && out3);
    // walk_only: the counts are those of the WALK (SURVEY 8d's BVH term, comparable between frames and builds): the occluder cache, whose
    // contents depend on the previous frames, is bypassed for the statistics pass.  Otherwise: the counts of the kernel render() launches
    const bool cache = p->occluder_cache;
    p->want_stats = true;
    if (walk_only) p->occluder_cache = false;
    hr_status s = hr_shadows_ray_trace(p, scene, in, prm, stream);
    p->want_stats = false;
    p->occluder_cache = cache;
    if (s != HR_OK) return s;
    HR_HIP(hipStreamSynchronize((hipStream_t)stream));
    uint64_t host[5];
    HR_HIP(hipMemcpy(host, p->counters.p, 40, hipMemcpyDeviceToHost));
    hr_status rs = hr_shadows_ray_count(p, &out3[0]);
    if (rs != HR_OK) return rs;
    out3[1] = host[2]; out3[2] = host[3];
    p->last_wave_max_steps = host[4];
    return HR_OK;
}

hr_status hr_shadows_trace_stats(hr_shadows* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_shadows_params* prm, uint64_t* out3, void* stream)
{
    return shadows_trace_stats_impl(p, scene, in, prm, out3, stream, true);
}

hr_status hr_shadows_trace_stats_timed(hr_shadows* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_shadows_params* prm, uint64_t* out3, void* stream)
{
    return shadows_trace_stats_impl(p, scene, in, prm, out3, stream, false);
}

hr_status hr_shadows_temporal(hr_shadows* p, const hr_frame_inputs* in, const hr_shadows_params* prm, void* stream_)
{
    HR_CHECK_ARG(p && in && prm);
    hr_status s = check_inputs(p, in, true);
    if (s != HR_OK) return s;
    hipStream_t st = (hipStream_t)stream_;
    p->last_stream = st;
    p->latched_exact = prm->exact ? 1 : 0;
    TemporalArgs a;
    for (int i = 0; i < 16; i++) a.vpi[i] = in->ubo.view_proj_inverse[i];
    a.mask = (const uint32_t*)p->mask.p; a.mw = p->mw; a.mh = p->mh;
    const int w = p->w, y0 = p->y0, y1 = p->y1, ry0 = p->ry0, ry1 = p->ry1;
    a.gb2  = ImgRGBA16F { (const uint2*)in->cur.gb2, w, ry0, ry1 };
    a.gb3  = ImgRGBA16F { (const uint2*)in->cur.gb3, w, ry0, ry1 };
    a.pgb2 = ImgRGBA16F { (const uint2*)in->prev.gb2, w, ry0, ry1 };
    a.pgb3 = ImgRGBA16F { (const uint2*)in->prev.gb3, w, ry0, ry1 };
    a.depth  = ImgR32F { in->cur.depth, w, ry0, ry1 };
    a.pdepth = ImgR32F { in->prev.depth, w, ry0, ry1 };
    a.hist         = ImgRG16F { (const uint32_t*)p->prev_image.p, w, ry0, ry1 };
    a.hist_moments = ImgRGBA16F { (const uint2*)p->moments[!in->ping_pong].p, w, ry0, ry1 };
    a.out = (uint32_t*)p->temporal_out.p; a.out_moments = (uint2*)p->moments[in->ping_pong ? 1 : 0].p;
    a.tile_class = (uint8_t*)p->tile_class.p;
    a.nd = (float4*)p->nd.p;
    a.geo_hist = nullptr;
    a.apron = GeoApronArgs { nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, -1 };
    if (!prm->exact)
    {
        // the records of the last frame stand for in->prev when the caller hands back what it passed as in->cur then; a band keeps to
        // the caller's images (its records cover the rows it computed, not the history apron its neighbours own)
        // (round 5: bands too — the records of the rows a band reads history from but does not compute, hr_band.history_halo beyond
        // hr_band.halo, are copies of the current G-buffer as well, written by a few extra workgroups of the temporal launch: GeoApronArgs)
        const size_t half = (size_t)p->w * p->h * 8;
        const bool   had_records = p->geo_history && p->geo_valid;
        if (p->geo_history && p->geo_valid && !p->first_frame && in->prev.gb2 == p->geo_gb2 && in->prev.gb3 == p->geo_gb3 && in->prev.gb2 != in->cur.gb2 && in->prev.gb3 != in->cur.gb3)
            a.geo_hist = (const char*)p->nd.p + (size_t)p->geo_parity * half;
        p->geo_parity ^= 1;
        a.nd = (float4*)((char*)p->nd.p + (size_t)p->geo_parity * half);
        p->geo_valid = true; p->geo_gb2 = in->cur.gb2; p->geo_gb3 = in->cur.gb3;
        if (p->geo_history && (ry0 < y0 || ry1 > y1))
            a.apron = GeoApronArgs { in->cur.gb2, in->cur.gb3, a.nd, w, ry0, y0, y1, ry1, -1 };
        if (p->dbg_require_geo && had_records && !a.geo_hist) { set_last_error("hr_shadows_temporal: HR_DEBUG_REQUIRE_GEO is set and the record path was not taken"); return HR_ERR_INVALID_ARG; }
    }
    else p->geo_valid = false;   // the parity mode's float4 layout covers both halves
    p->nd_cur = a.nd;
    a.w = w; a.h = p->h; a.y0 = y0; a.y1 = y1;
    a.tile_y0 = y0 / 8; a.tiles_x = p->tiles_x; a.tiles_y = cdiv(y1, 8) - a.tile_y0;
    a.alpha = prm->alpha; a.moments_alpha = prm->moments_alpha;
    a.debug_skip_reproject = p->dbg_skip_reproject ? 1 : 0;
    a.apron_flag = (p->y0 > 0 || p->y1 < p->h) ? (uint32_t*)((char*)p->counters.p + 48) : nullptr;   // row bands only
    a.band_y0 = p->band_y0; a.band_y1 = p->band_y1;
    p->last_ping_pong = in->ping_pong ? 1 : 0;
    const uint64_t px = (uint64_t)w * (y1 - y0);
    int ev = p->prof.begin("temporal_accumulation", st, px * 64 + px / 8);
    a.sort = TileSortArgs { nullptr, nullptr, 0, 0, 0, 0 };
    if (prm->exact) hipLaunchKernelGGL(k_shadows_temporal, dim3(cdiv(a.tiles_x * a.tiles_y, TEMPORAL_WAVES)), dim3(64 * TEMPORAL_WAVES), 0, st, a);
    else
    {
        a.sort = p->tile_order.ride();   // the trace kernel's next launch order rides along (tile_order.h)
        launch_shadows_temporal_fast(a, a.tiles_x * a.tiles_y, st);
    }
    p->prof.end(ev, st);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

static hr_status check_mode(hr_shadows* p, const hr_shadows_params* prm);

hr_status hr_shadows_atrous_iteration(hr_shadows* p, const hr_frame_inputs* in, const hr_shadows_params* prm, int32_t i, void* stream_)
{
    HR_CHECK_ARG(p && in && prm && i >= 0 && i < prm->filter_iterations && prm->filter_iterations <= 8 && prm->radius >= 0 && prm->radius <= 2);
    hr_status s = check_inputs(p, in, false);
    if (s != HR_OK) return s;
    if ((s = check_mode(p, prm)) != HR_OK) return s;
    hipStream_t st = (hipStream_t)stream_;
    p->last_stream = st;
    // ping-pong as a_trous_filter() (:1101-1107,1175): read = i odd, write = i even ? 1 : 0
    const int read_idx = i & 1, write_idx = (i & 1) ^ 1;
    AtrousArgs a;
    const int w = p->w, y0 = p->y0, y1 = p->y1;
    a.in  = ImgRG16F { (const uint32_t*)(i == 0 ? p->temporal_out.p : p->atrous[read_idx].p), w, y0, y1 };
    a.nd = (const float4*)(p->nd_cur ? p->nd_cur : p->nd.p);
    a.tile_class = (const uint8_t*)p->tile_class.p;
    a.out  = (uint32_t*)p->atrous[write_idx].p;
    a.out2 = (prm->feedback_iteration == i) ? (uint32_t*)p->prev_image.p : nullptr; // vkCmdCopyImage :1177-1207
    a.w = w; a.h = p->h; a.y0 = y0; a.y1 = y1; a.tiles_x = p->tiles_x;
    a.radius = prm->radius; a.step = 1 << i;
    a.phi_visibility = prm->phi_visibility; a.phi_normal = prm->phi_normal; a.sigma_depth = prm->sigma_depth;
    a.power = (i == prm->filter_iterations - 1) ? prm->power : 0.0f;
    p->read_idx = write_idx;
    static const char* names[8] = { "atrous_0", "atrous_1", "atrous_2", "atrous_3", "atrous_4", "atrous_5", "atrous_6", "atrous_7" };
    const uint64_t px = (uint64_t)w * (y1 - y0);
    int ev = p->prof.begin(names[i], st, px * 24 + (a.out2 ? px * 4 : 0));
    const dim3 grid(cdiv(w, 32), cdiv(y1 - y0, 8));
    const bool fast = (prm->phi_normal == 32.0f && prm->sigma_depth == 1.0f);
    if (!prm->exact) launch_shadows_atrous_fast(a, st);
    else if (prm->radius == 1 && fast) hipLaunchKernelGGL((k_shadows_atrous<1, true>), grid, dim3(256), 0, st, a);
    else if (prm->radius == 1) hipLaunchKernelGGL((k_shadows_atrous<1, false>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_shadows_atrous<-1, false>), grid, dim3(256), 0, st, a);
    p->prof.end(ev, st);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

// the temporal stage writes the normal / depth side image `nd` in a mode-specific layout (packed uint2 in tolerance mode, float4 in the
// parity mode): an a-trous stage in the other mode would read garbage — refuse it instead (round-2 advisor)
static hr_status check_mode(hr_shadows* p, const hr_shadows_params* prm)
{
    if (p->latched_exact >= 0 && p->latched_exact != (prm->exact ? 1 : 0))
    {
        set_last_error("hr_shadows: params.exact changed between the temporal and the a-trous stage of one frame");
        return HR_ERR_INVALID_ARG;
    }
    return HR_OK;
}

// tolerance mode, radius 1: iterations 0 and 1 in one launch (kf_shadows_atrous01; iteration 0's image stays in LDS)
static hr_status shadows_atrous01(hr_shadows* p, const hr_frame_inputs* in, const hr_shadows_params* prm, void* stream_, bool* done)
{
    hr_status s = check_inputs(p, in, false);
    if (s != HR_OK) return s;
    if ((s = check_mode(p, prm)) != HR_OK) return s;
    hipStream_t st = (hipStream_t)stream_;
    p->last_stream = st;
    AtrousArgs a;
    const int w = p->w, y0 = p->y0, y1 = p->y1;
    a.in  = ImgRG16F { (const uint32_t*)p->temporal_out.p, w, y0, y1 };
    a.nd = (const float4*)(p->nd_cur ? p->nd_cur : p->nd.p);
    a.tile_class = (const uint8_t*)p->tile_class.p;
    a.out  = (uint32_t*)p->atrous[0].p;   // iteration 1 writes atrous[0] (ping-pong of a_trous_filter(), :1101-1107)
    a.out2 = (prm->feedback_iteration == 1) ? (uint32_t*)p->prev_image.p : nullptr;
    uint32_t* first2 = (prm->feedback_iteration == 0) ? (uint32_t*)p->prev_image.p : nullptr;
    a.w = w; a.h = p->h; a.y0 = y0; a.y1 = y1; a.tiles_x = p->tiles_x;
    a.radius = prm->radius; a.step = 1;
    a.phi_visibility = prm->phi_visibility; a.phi_normal = prm->phi_normal; a.sigma_depth = prm->sigma_depth;
    a.power = 0.0f;
    const uint64_t px = (uint64_t)w * (y1 - y0);
    // algorithmic bytes: in 4 + normal/depth 8 + out 4 (+ 4 feedback); the G-buffer side image counts once
    int ev = p->prof.begin("atrous_01", st, px * 16 + ((a.out2 || first2) ? px * 4 : 0));
    *done = launch_shadows_atrous01_fast(a, first2, prm->filter_iterations == 2 ? prm->power : 0.0f, st);
    p->prof.end(ev, st);
    if (*done) p->read_idx = 0;
    HR_HIP(hipGetLastError());
    return HR_OK;
}

hr_status hr_shadows_upsample(hr_shadows* p, const hr_frame_inputs* in, const hr_shadows_params* prm, void* stream_)
{
    HR_CHECK_ARG(p && in && prm);
    if (p->scale == 0) return HR_OK; // ray_traced_shadows.cpp:113
    HR_CHECK_ARG(in->cur_full.gb2 && in->cur_full.gb3 && in->cur_full.width == p->full_w && in->cur_full.height == p->full_h);
    hipStream_t st = (hipStream_t)stream_;
    p->last_stream = st;
    UpsampleArgs a;
    a.W = p->full_w; a.H = p->full_h; a.w = p->w; a.h = p->h;
    a.G2 = (const uint2*)in->cur_full.gb2; a.G3 = (const uint2*)in->cur_full.gb3;
    a.g2 = (const uint2*)in->cur.gb2; a.g3 = (const uint2*)in->cur.gb3;
    a.in = p->atrous[p->read_idx].p; a.in_channels = 2; a.channels = 1;
    a.out = p->upsample.p; a.sky_value = 0.0f; a.power = 0.0f;
    const uint64_t PX = (uint64_t)p->full_w * p->full_h, px = (uint64_t)p->w * p->h;
    int ev = p->prof.begin("upsample", st, PX * 18 + px * 20);
    if (prm->exact) launch_upsample(a, st);
    else launch_upsample_fast(a, st);
    p->prof.end(ev, st);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

// Everything of RayTracedShadows::render after the ray trace (ray_traced_shadows.cpp:104-113): temporal accumulation, the a-trous
// chain, the upsample of a scaled pass.  A caller that overlaps work with the trace (row bands: the history exchange) calls
// hr_shadows_ray_trace + hr_shadows_denoise instead of hr_shadows_render and gets the same launches — in tolerance mode the fused ones.
hr_status hr_shadows_denoise(hr_shadows* p, const hr_frame_inputs* in, const hr_shadows_params* prm, void* stream)
{
    HR_CHECK_ARG(p && in && prm);
    p->last_denoise = prm->denoise != 0;
    if (!prm->denoise) return HR_OK;
    hr_status s;
    if ((s = hr_shadows_temporal(p, in, prm, stream)) != HR_OK) return s;
    {
        HR_SCOPED_SAMPLE("A-Trous Filter");   // ray_traced_shadows.cpp:1096
        bool fused = false;
        if (!prm->exact && p->fuse && prm->filter_iterations >= 2 && prm->filter_iterations <= 8 && prm->radius == 1 &&
            (s = shadows_atrous01(p, in, prm, stream, &fused)) != HR_OK) return s;
        for (int i = fused ? 2 : 0; i < prm->filter_iterations; i++)
            if ((s = hr_shadows_atrous_iteration(p, in, prm, i, stream)) != HR_OK) return s;
    }
    if (p->scale != 0 && (s = hr_shadows_upsample(p, in, prm, stream)) != HR_OK) return s;
    return HR_OK;
}

hr_status hr_shadows_render(hr_shadows* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_shadows_params* prm, void* stream)
{
    HR_SCOPED_SAMPLE("Ray Traced Shadows");
    HR_CHECK_ARG(p && scene && in && prm);
    HR_HIP(hipSetDevice(p->ctx->device));
    p->prof.begin_frame();
    p->last_denoise = prm->denoise != 0;
    hr_status s = hr_shadows_ray_trace(p, scene, in, prm, stream);
    if (s != HR_OK) return s;
    return hr_shadows_denoise(p, in, prm, stream);
}

static void fill_view(hr_image_view* v, void* data, int w, int h, int bpp, hr_format f)
{
    v->data = data; v->width = w; v->height = h; v->row_pitch_bytes = w * bpp; v->format = f;
}

hr_status hr_shadows_image(hr_shadows* p, int32_t which, hr_image_view* v)
{
    HR_CHECK_ARG(p && v);
    switch (which)
    {
        case 0: fill_view(v, p->mask.p, p->mw, p->mh, 4, HR_FORMAT_R32_UINT); break;
        case 1: fill_view(v, p->temporal_out.p, p->w, p->h, 4, HR_FORMAT_RG16F); break;
        case 2: fill_view(v, p->moments[0].p, p->w, p->h, 8, HR_FORMAT_RGBA16F); break;
        case 3: fill_view(v, p->moments[1].p, p->w, p->h, 8, HR_FORMAT_RGBA16F); break;
        case 4: fill_view(v, p->prev_image.p, p->w, p->h, 4, HR_FORMAT_RG16F); break;
        case 5: fill_view(v, p->atrous[0].p, p->w, p->h, 4, HR_FORMAT_RG16F); break;
        case 6: fill_view(v, p->atrous[1].p, p->w, p->h, 4, HR_FORMAT_RG16F); break;
        case 7: fill_view(v, p->upsample.p, p->full_w, p->full_h, 2, HR_FORMAT_R16F); break;
        case 8: fill_view(v, p->tile_class.p, p->tiles_x, p->tiles_y, 1, (hr_format)0); break;
        case 9:   // tolerance mode: the geometry records {GB2.x, GB3.y} the last temporal stage wrote — tests / tools
            if (!p->nd_cur || p->latched_exact != 0) { set_last_error("hr_shadows_image(9): no geometry records (parity mode, or no temporal stage yet)"); return HR_ERR_INVALID_ARG; }
            fill_view(v, p->nd_cur, p->w, p->h, 8, HR_FORMAT_RGBA16F); break;
        default: set_last_error("hr_shadows_image: unknown image index"); return HR_ERR_INVALID_ARG;
    }
    return HR_OK;
}

// RayTracedShadows::output_ds (ray_traced_shadows.cpp:135-155)
hr_status hr_shadows_output(hr_shadows* p, hr_output_kind kind, hr_image_view* v)
{
    HR_CHECK_ARG(p && v);
    if (!p->last_denoise || kind == HR_OUTPUT_RAY_TRACE) return hr_shadows_image(p, 0, v);
    if (kind == HR_OUTPUT_TEMPORAL_ACCUMULATION) return hr_shadows_image(p, 1, v);
    if (kind == HR_OUTPUT_ATROUS || p->scale == 0) return hr_shadows_image(p, 5 + p->read_idx, v);
    return hr_shadows_image(p, 7, v);
}

} // extern "C"
