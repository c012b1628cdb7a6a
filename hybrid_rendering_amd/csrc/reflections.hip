// RayTracedReflections on MI355X — HIP replacement for src/ray_traced_reflections.{h,cpp} and
// src/shaders/reflections/*.
//   ray_trace()              ray_traced_reflections.cpp:997-1057, reflections_ray_trace.{rgen:119-171,rchit:117-150,rmiss:26-30} -> k_refl_trace
//   temporal_accumulation()  :1087-1139, reflections_denoise_reprojection.comp:174-289                                         -> k_refl_temporal
//   a_trous_filter()         :1143-1256, reflections_denoise_atrous.comp:94-181 + reflections_denoise_copy_tiles.comp:34-38   -> k_refl_atrous
//   upsample()               :1260-1296, reflections_upsample.comp:62-109                                                     -> k_upsample<4>
// NB clear_images() resets m_first_frame before ray_trace() reads it (:962-991 vs :1017-1018), so sample_gi /
// approximate_with_ddgi are never forced off; they are taken from the params as given.
#include "hr_internal.h"
#include "reproject.h"
#include "shading.h"
#include "upsample.h"
#include "pass_args.h"
#include "tile_order.h"

using namespace hr;

HR_DEV f3 reflect3(f3 I, f3 N) { return sub3(I, scale3(N, 2.0f * dot3(N, I))); }

// reflections_ray_trace.rgen:78-105
HR_DEV f3 importance_sample_ggx(float Ex, float Ey, f3 N, float roughness)
{
    const float a = roughness * roughness, m2 = a * a;
    const float phi = 2.0f * HR_M_PI * Ex;
    const float ct  = hr_sqrt(__fdiv_rn(1.0f - Ey, 1.0f + (m2 - 1.0f) * Ey));
    const float st  = hr_sqrt(1.0f - ct * ct);
    float s, c;
    det_sincos(phi, s, c);
    const f3 H  = mk3(c * st, s * st, ct);
    const f3 up = fabsf(N.z) < 0.999f ? mk3(0.0f, 0.0f, 1.0f) : mk3(1.0f, 0.0f, 0.0f);
    const f3 tangent   = normalize3(cross3(up, N));
    const f3 bitangent = cross3(N, tangent);
    return normalize3(add3(add3(scale3(tangent, H.x), scale3(bitangent, H.y)), scale3(N, H.z)));
}

struct EnvDev
{
    CubeMap         sky;
    const uint2*    prefiltered;
    int             pre_size, pre_levels;
    const uint32_t* lut;
    int             lut_size;
    HR_DEV f3 prefiltered_fetch(f3 dir, float lod) const
    {
        int level = (int)floorf(lod + 0.5f);
        level     = level < 0 ? 0 : (level > pre_levels - 1 ? pre_levels - 1 : level);
        size_t off = 0;
        for (int l = 0; l < level; l++) off += (size_t)6 * (pre_size >> l) * (pre_size >> l);
        CubeMap c { prefiltered + off, pre_size >> level };
        return c.fetch(dir);
    }
    HR_DEV void lut_fetch(float u, float v, float& a, float& b) const
    {
        int ix = (int)floorf(u * (float)lut_size), iy = (int)floorf(v * (float)lut_size);
        ix = ix < 0 ? 0 : (ix > lut_size - 1 ? lut_size - 1 : ix);
        iy = iy < 0 ? 0 : (iy > lut_size - 1 ? lut_size - 1 : iy);
        const uint32_t q = lut[(size_t)iy * lut_size + ix];
        a = h2f_lo(q); b = h2f_hi(q);
    }
};

struct ReflTraceArgs
{
    DDGIU          d;
    hr_light       light;
    float          vpi[16];
    float          cam[3];
    const float*   depth;
    const uint2*   gb2;
    const uint2*   gb3;
    const uint8_t* sobol;
    const uint8_t* sr;
    const Node8*   nodes;
    const TriGPU*  tris;
    SceneShading   sh;
    EnvDev         env;
    AtlasRGBA      irr;
    AtlasRG        dep;
    uint2*         out;
    uint32_t*      ray_slots;  // rays per 8x8 tile
    int            w, h, y0, y1, tiles_x, tiles_y, tile_y0;
    float          bias, trim;
    uint32_t       num_frames;
    int            sample_gi, approximate_with_ddgi;
    float          gi_intensity, rough_ddgi_intensity, ibl_intensity;
    unsigned long long* stats;   // instrumented build only (k_refl_trace<true>): [0] node steps, [1] triangle tests, [2] rays
    const uint32_t* order;       // nullable: launch slot -> tile, last frame's heaviest tiles first (tile_order.h)
    uint16_t*       cost;        // nullable: per tile, how long its wave lived (100 MHz ticks)
};

#ifdef HR_TRACE_DIVERGENCE
static __device__ unsigned long long g_div_refl[16];   // 0-7 reflection rays, 8-15 secondary rays
extern "C" int hr_debug_divergence_refl(uint64_t* out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_div_refl), sizeof(g_div_refl)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_div_refl), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif
#ifndef REFL_TRACE_WAVES
#define REFL_TRACE_WAVES 1
#endif
#ifndef REFL_COOP
#define REFL_COOP 1   // wave-cooperative triangle tests (traverse.h trace_coop); 0 = the per-lane loops
#endif
#ifndef REFL_TRACE_EU
#define REFL_TRACE_EU 1   // minimum waves per SIMD the register allocator must leave room for: 1 / 6 / 7 -> 216 / 226 / 235 us
#endif
// STATS: the instrumented build behind hr_reflections_trace_stats (see k_ddgi_trace); the product launches <false>.
// The DDGI irradiance gathers (rough pixels: reflections_ray_trace.rgen:152 approximate_with_ddgi; hit points: .rchit:87-111 indirect_lighting)
// run the PARITY arithmetic in both modes (round 6).  Round 4-5's tolerance mode sent them through ddgi_sample_fast.h (149 vs 185 us at
// 1080p): ~0.02 % of the trace image's colours then sat one fp16 ulp off, and one frame later `m2 - m1^2` of the stored luminance
// moments turned that into 0.4-100 % of a small variance, which normalises the a-trous luminance weights — six fuzzed sequences missed
// the 99.9 % population bound because of it (docs/EXPERIMENTS.md R5.8, R6.1), none does with the parity gather.  What takes the edge off
// its cost: a lane is a rough pixel or has a hit point, never both, so ONE gather site per wave serves both kinds (the rolled eight-probe
// loop runs once per wave instead of once at each of two sites: 185 -> 172 us at 1080p, 488 -> 466 us at 4K; prepared denominators for the
// atlas coordinates, shading.h IrrDiv: -> 165 / 454; the fast gather was 149 / 382).
// The trace image is bit-exact in tolerance mode too.
template <bool STATS>
__global__ __launch_bounds__(64 * REFL_TRACE_WAVES, REFL_TRACE_EU) void k_refl_trace(ReflTraceArgs a)
{
    __shared__ uint32_t s_stack[REFL_TRACE_WAVES][HR_STACK_ENTRIES * 64];
#if REFL_COOP
    __shared__ CoopWave s_coop[REFL_TRACE_WAVES];
#endif
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int launch_slot = blockIdx.x * REFL_TRACE_WAVES + wave;
    if (launch_slot >= a.tiles_x * a.tiles_y) return;   // wave-uniform
    const int tile = a.order ? (int)a.order[launch_slot] : launch_slot;
    const unsigned long long t_begin = a.cost ? wall_clock64() : 0ull;
    const int x = (tile % a.tiles_x) * 8 + (lane & 7), y = (tile / a.tiles_x + a.tile_y0) * 8 + (lane >> 3);
    uint32_t  rays = 0;
    HR_DIV(DivCounters dvp = {}, dvs = {};)
    // per pixel: the reflection ray (or none); the wave stays converged around the traversal
    const size_t o = (size_t)y * a.w + x;
    bool  geom = false, trace = false;
    f3    color = mk3(0.0f, 0.0f, 0.0f), dir = mk3(0.0f, 0.0f, 1.0f), ray_origin = mk3(0.0f, 0.0f, 0.0f);
    float ray_length = -1.0f;
    // the lane's DDGI gather, if any (kind 1: rough pixel, 2: hit point) and what its result is combined with
    int gather = 0;
    f3  gP = mk3(0.0f, 0.0f, 0.0f), gN = mk3(0.0f, 0.0f, 1.0f), gWo = mk3(0.0f, 0.0f, 1.0f);
    f3  hLo = mk3(0.0f, 0.0f, 0.0f), hkD = hLo, hcd = hLo, hspec = hLo;
    if (x < a.w && y >= a.y0 && y < a.y1)
    {
        const float  dp = a.depth[o];
        if (dp == 1.0f) a.out[o] = make_uint2(0u, pack_h2(0.0f, -1.0f));
        else
        {
            geom = true;
            const uint2 g2 = a.gb2[o], g3 = a.gb3[o];
            const float roughness = h2f_lo(g3.x);
            const float tu = __fdiv_rn((float)x + 0.5f, (float)a.w), tv = __fdiv_rn((float)y + 0.5f, (float)a.h);
            const f3 P  = world_pos_from_depth(tu, tv, dp, a.vpi);
            const f3 N  = oct_decode(h2f_lo(g2.x), h2f_hi(g2.x));
            const f3 Wo = normalize3(sub3(mk3(a.cam[0], a.cam[1], a.cam[2]), P));
            ray_origin  = add3(P, scale3(N, a.bias));
            if (roughness < 0.05f) { dir = reflect3(neg3(Wo), N); trace = true; }
            else if (roughness > 0.75f && a.approximate_with_ddgi == 1)
            {
                gather = 1; gP = P; gN = reflect3(neg3(Wo), N); gWo = Wo;
            }
            else
            {
                const float r0 = sample_blue_noise(x, y, (int)a.num_frames, 0, a.sobol, a.sr) * a.trim;
                const float r1 = sample_blue_noise(x, y, (int)a.num_frames, 1, a.sobol, a.sr) * a.trim;
                const f3    Wh = importance_sample_ggx(r0, r1, N, roughness);
                dir   = reflect3(neg3(Wo), Wh);
                trace = true;
            }
        }
    }
    if (trace) rays++;
    uint32_t st_n = 0, st_t = 0;
#if REFL_COOP
    HitRec hit;
    if (!STATS) hit = trace_coop<false>(trace, a.nodes, a.tris, ray_origin, dir, 0.001f, 10000.0f, s_stack[wave], s_coop[wave], lane, 0u HR_DIV(, &dvp));
    else
    {
        hit.prim = -1;
        if (trace) hit = trace_closest<STATS>(a.nodes, a.tris, ray_origin, dir, 0.001f, 10000.0f, s_stack[wave], lane, nullptr, &st_n, &st_t);
    }
#else
    HitRec hit;
    hit.prim = -1;
    DivCounters* dvp_ptr = nullptr;   // (without HR_TRACE_DIVERGENCE the HR_DIV() argument vanished and &st_n slid into the `dv` slot)
    HR_DIV(dvp_ptr = &dvp;)
    if (trace) hit = trace_closest<STATS>(a.nodes, a.tris, ray_origin, dir, 0.001f, 10000.0f, s_stack[wave], lane, dvp_ptr, &st_n, &st_t);
#endif
    if (trace)
    {
        if (hit.prim < 0) { color = a.env.sky.fetch(dir); ray_length = -1.0f; }
        else
        {
            const SurfaceHit s = surface_at(a.sh, hit);
            const f3 hWo = neg3(dir);
            const f3 F0  = mix3(mk3(0.04f, 0.04f, 0.04f), s.albedo, s.metallic);
            const f3 c_diffuse = mix3(mul3(s.albedo, sub3(one3(), F0)), mk3(0.0f, 0.0f, 0.0f), s.metallic);
            TraceCtx tc { a.nodes, a.tris, s_stack[wave], lane };
            HR_DIV(tc.dv = &dvs;)
            CubeMap  none { nullptr, 0 };
            hLo = direct_lighting<STATS>(tc, a.light, hWo, s.N, s.P, F0, c_diffuse, s.roughness, one3(), false, 0.0f, 0.0f, none, rays);
            if (STATS) { st_n += tc.nn; st_t += tc.nt; }
            if (a.sample_gi == 1)
            {
                const f3    R   = reflect3(neg3(hWo), s.N);
                const float ndv = max2(dot3(s.N, hWo), 0.0f);
                const f3    F   = fresnel_schlick_roughness(ndv, F0, s.roughness);
                hkD             = scale3(sub3(one3(), F), 1.0f - s.metallic);
                const f3    pre = a.env.prefiltered_fetch(R, s.roughness * 4.0f);
                float bx, by;
                a.env.lut_fetch(ndv, s.roughness, bx, by);
                hspec = scale3(mul3(pre, add3(scale3(F, bx), mk3(by, by, by))), a.ibl_intensity);
                hcd   = c_diffuse;
                gather = 2; gP = s.P; gN = s.N; gWo = hWo;
            }
            color      = hLo;
            ray_length = 0.001f + hit.t;
        }
    }
    // ONE parity gather for the rough pixels and the hit points of the tile (shading.h; the traversal stack is idle here — the cooperative A/B form borrows it)
    {
        const f3 net = sample_irradiance_net_coop(gather != 0, a.d, gP, gN, gWo, a.irr, a.dep, reinterpret_cast<float*>(s_stack[wave]), lane);
        if (gather == 1) color = scale3(irradiance_from_net(a.d, net), a.rough_ddgi_intensity);
        else if (gather == 2)
        {
            const f3 diffuse = mul3(scale3(hcd, a.gi_intensity), irradiance_from_net(a.d, net));
            color = add3(hLo, add3(mul3(hkD, diffuse), hspec));
        }
    }
    if (geom) a.out[o] = make_uint2(pack_h2(min2(color.x, 0.7f), min2(color.y, 0.7f)), pack_h2(min2(color.z, 0.7f), ray_length));
    HR_DIV(div_flush(dvp, g_div_refl); div_flush(dvs, g_div_refl + 8);)
    for (int o2 = 32; o2 > 0; o2 >>= 1) rays += __shfl_down(rays, o2);
    if (lane == 0) a.ray_slots[tile] = rays;
    if (lane == 0 && a.cost)
    {
        const unsigned long long ticks = wall_clock64() - t_begin;
        a.cost[tile] = (uint16_t)(ticks > 65535ull ? 65535ull : ticks);
    }
    if (STATS)
    {
        for (int o2 = 32; o2 > 0; o2 >>= 1) { st_n += __shfl_down(st_n, o2); st_t += __shfl_down(st_t, o2); }
        if (lane == 0) { atomicAdd(a.stats + 0, (unsigned long long)st_n); atomicAdd(a.stats + 1, (unsigned long long)st_t); atomicAdd(a.stats + 2, (unsigned long long)rays); }
    }
}

// ------------------------------------------------------------------------------------------------

#define RT_TW 32
#define RT_TH 8
#define RT_R 8
// 32x8 pixel tile per workgroup; the 48x24 input colours around it are staged in LDS as fp32 so the
// 17x17 neighbourhood statistics (neighborhood_standard_deviation :133-157, 289 un-tiled fetches per
// pixel in the reference) read LDS.  The fp32 running sums keep the reference's order (dx outer, dy inner).
__global__ __launch_bounds__(256) void k_refl_temporal(ReflTemporalArgs a)
{
    __shared__ float s_col[(RT_TH + 2 * RT_R)][(RT_TW + 2 * RT_R)][3];
    __shared__ float s_vpi[16], s_pvp[16];
    __shared__ int   s_flag[4];
    const int bx0 = blockIdx.x * RT_TW, by0 = a.y0 + blockIdx.y * RT_TH;
    if (threadIdx.x < 16) { s_vpi[threadIdx.x] = a.vpi[threadIdx.x]; s_pvp[threadIdx.x] = a.pvp[threadIdx.x]; }
    if (threadIdx.x < 4) s_flag[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < (RT_TH + 2 * RT_R) * (RT_TW + 2 * RT_R); i += 256)
    {
        const int   cy = i / (RT_TW + 2 * RT_R), cx = i % (RT_TW + 2 * RT_R);
        const uint2 q  = a.in.raw(bx0 - RT_R + cx, by0 - RT_R + cy);
        s_col[cy][cx][0] = h2f_lo(q.x); s_col[cy][cx][1] = h2f_hi(q.x); s_col[cy][cx][2] = h2f_lo(q.y);
    }
    __syncthreads();
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int x = bx0 + lx, y = by0 + ly;
    bool flag = false;
    if (x < a.w && y < a.y1)
    {
        const size_t o = (size_t)y * a.w + x;
        const float  d = a.depth.p[o];
        const float  roughness = h2f_lo(a.gb3.p[o].x);
        float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f, r3 = 0.0f, m0 = 0.0f, m1 = 0.0f, hl = 0.0f;
        if (d != 1.0f)
        {
            const uint2 cq = a.in.p[o];
            const f3    color = mk3(h2f_lo(cq.x), h2f_hi(cq.x), h2f_lo(cq.y));
            const float ray_length = h2f_hi(cq.y);
            ReprojIn in;
            in.x = x; in.y = y; in.depth = d; in.vpi = s_vpi;
            in.cam_pos = mk3(a.cam[0], a.cam[1], a.cam[2]); in.prev_vp = s_pvp; in.ray_length = ray_length;
            in.gb2 = a.gb2; in.gb3 = a.gb3; in.pgb2 = a.pgb2; in.pgb3 = a.pgb3; in.pdepth = a.pdepth; in.w = a.w; in.h = a.h;
            float hc[3], hm[2];
            ImgR16F none { nullptr, 0, 0, 0 };
            const bool success = reproject<false, true, true, ImgRGBA16F>(in, a.hist, a.hist_moments, none, hc, hm, hl);
            if (a.apron_flag && in.apron_miss && y >= a.band_y0 && y < a.band_y1) atomicOr(a.apron_flag, 1u);   // rare (motion beyond the history apron of a row band)
            hl = min2(32.0f, success ? hl + 1.0f : 1.0f);
            f3 history = mk3(hc[0], hc[1], hc[2]);
            if (success)
            {
                f3 s1 = mk3(0.0f, 0.0f, 0.0f), s2 = mk3(0.0f, 0.0f, 0.0f);
                for (int dx = 0; dx <= 2 * RT_R; dx++)
                    for (int dy = 0; dy <= 2 * RT_R; dy++)
                    {
                        const float* c = s_col[ly + dy][lx + dx];
                        const f3     s = mk3(c[0], c[1], c[2]);
                        s1 = add3(s1, s);
                        s2 = add3(s2, mul3(s, s));
                    }
                const f3 mean = div3s(s1, 289.0f);
                const f3 var  = sub3(div3s(s2, 289.0f), mul3(mean, mean));
                const f3 sd   = mk3(hr_sqrt(max2(var.x, 0.0f)), hr_sqrt(max2(var.y, 0.0f)), hr_sqrt(max2(var.z, 0.0f)));
                const f3 amin = sub3(mean, sd), amax = add3(mean, sd);
                // clip_aabb (:111-129)
                const f3 center = scale3(add3(amax, amin), 0.5f);
                const f3 extent = add3(scale3(sub3(amax, amin), 0.5f), mk3(0.001f, 0.001f, 0.001f));
                const f3 cv     = sub3(history, center);
                const float mx  = max2(max2(fabsf(__fdiv_rn(cv.x, extent.x)), fabsf(__fdiv_rn(cv.y, extent.y))), fabsf(__fdiv_rn(cv.z, extent.z)));
                if (mx > 1.0f) history = add3(center, div3s(cv, mx));
            }
            const float max_acc = a.moving ? 8.0f : hl;
            const float al = success ? max2(a.alpha, __fdiv_rn(1.0f, max_acc)) : 1.0f;
            const float am = success ? max2(a.moments_alpha, __fdiv_rn(1.0f, max_acc)) : 1.0f;
            m0 = luminance(color);
            m1 = m0 * m0;
            m0 = mix1(hm[0], m0, am);
            m1 = mix1(hm[1], m1, am);
            r3 = max2(0.0f, m1 - m0 * m0);
            const f3 acc = mix3(history, color, al);
            r0 = acc.x; r1 = acc.y; r2 = acc.z;
        }
        a.out_moments[o] = make_uint2(pack_h2(m0, m1), pack_h2(hl, 0.0f));
        a.out[o]         = make_uint2(pack_h2(r0, r1), pack_h2(r2, r3));
        if (d != 1.0f && roughness >= 0.05f) flag = (a.approximate_with_ddgi == 1) ? (roughness <= 0.75f) : true;
    }
    // tile classification per 8x8 tile (4 tiles per workgroup): :262-272
    if (flag) atomicOr(&s_flag[lx >> 3], 1);
    __syncthreads();
    if (threadIdx.x < 4)
    {
        const int tx = (bx0 >> 3) + threadIdx.x, ty = by0 >> 3;
        if (tx < a.tiles_x) a.tile_class[(size_t)ty * a.tiles_x + tx] = s_flag[threadIdx.x] ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------
// pow(x, phi_normal): phi_normal = 32 (the reference default) is five squarings — the multiplications det_powi does for n = 32
HR_DEV float pow_phi_normal(float x, float p)
{
    if (p == 32.0f) { float b = x * x; b = b * b; b = b * b; b = b * b; return 1.0f * (b * b); }
    return det_pow_auto(x, p);
}


// RADIUS == 1 (the reference default): fully unrolled 3x3 stencil, every load issued up front and the eight tap weights
// computed unconditionally before the ordered accumulation (see k_shadows_atrous); RADIUS < 0: run-time radius.
template <int RADIUS>
__global__ __launch_bounds__(256) void k_refl_atrous(ReflAtrousArgs a)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = a.y0 + blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= a.w || y >= a.y1) return;
    const size_t o = (size_t)y * a.w + x;
    const uint2  c = a.in.p[o];
    uint2        result = c;
    if (a.tile_class[(size_t)(y >> 3) * a.tiles_x + (x >> 3)])
    {
        const f3    cc = mk3(h2f_lo(c.x), h2f_hi(c.x), h2f_lo(c.y));
        const float center_luma = luminance(cc);
        float var = 0.0f;
#pragma unroll
        for (int yy = -1; yy <= 1; yy++)
#pragma unroll
            for (int xx = -1; xx <= 1; xx++)
            {
                const float k = (xx == 0 ? (yy == 0 ? 0.25f : 0.125f) : (yy == 0 ? 0.125f : 0.0625f));
                var += h2f_hi(a.in.raw(x + xx, y + yy).y) * k;
            }
        const uint2 g2 = a.gb2.p[o], g3 = a.gb3.p[o];
        const float d = a.depth.p[o], roughness = h2f_lo(g3.x);
        if (d == 1.0f) result = make_uint2(0u, 0u);
        else if (!(roughness < 0.05f || (a.approximate_with_ddgi == 1 && roughness > 0.75f)))
        {
            const f3    cn = oct_decode(h2f_lo(g2.x), h2f_hi(g2.x));
            const float center_depth = h2f_hi(g3.y);
            // one denominator, many numerators: share the denominator half of the correctly rounded divisions (device_math.h)
            const DivBy phi_c = div_prepare(a.phi_color * hr_sqrt(max2(0.0f, 1e-10f + var))), by_sigma = div_prepare(a.sigma_depth);
            float sum_w = 1.0f, s0 = cc.x, s1 = cc.y, s2 = cc.z, s3 = h2f_hi(c.y);
            if (RADIUS == 1)
            {
                uint2 t_in[8], t_g2[8], t_g3[8];
                bool  t_ok[8];
#pragma unroll
                for (int t = 0; t < 8; t++)
                {
                    const int k = t < 4 ? t : t + 1, xx = k % 3 - 1, yy = k / 3 - 1;
                    const int px = x + xx * a.step, py = y + yy * a.step;
                    t_ok[t] = px >= 0 && py >= 0 && px < a.w && py < a.h;
                    t_in[t] = a.in.raw(px, py); t_g2[t] = a.gb2.raw(px, py); t_g3[t] = a.gb3.raw(px, py);
                }
                float w8[8];
#pragma unroll
                for (int t = 0; t < 8; t++)
                {
                    const f3    sc = mk3(h2f_lo(t_in[t].x), h2f_hi(t_in[t].x), h2f_lo(t_in[t].y));
                    const float sl = luminance(sc);
                    const f3    sn = oct_decode(h2f_lo(t_g2[t].x), h2f_hi(t_g2[t].x));
                    const float wZ = det_exp(div_by(-fabsf(center_depth - h2f_hi(t_g3[t].y)), by_sigma));
                    const float wN = pow_phi_normal(clamp1(dot3(cn, sn), 0.0f, 1.0f), a.phi_normal);
                    const float wL = div_by(fabsf(center_luma - sl), phi_c);
                    w8[t] = det_exp((0.0f - max2(wL, 0.0f)) - max2(wZ, 0.0f)) * wN;
                }
#pragma unroll
                for (int t = 0; t < 8; t++)
                {
                    if (!t_ok[t]) continue;
                    const int   k = t < 4 ? t : t + 1, xx = k % 3 - 1, yy = k / 3 - 1;
                    const float kx = xx == 0 ? 1.0f : __fdiv_rn(2.0f, 3.0f), ky = yy == 0 ? 1.0f : __fdiv_rn(2.0f, 3.0f);
                    const f3    sc = mk3(h2f_lo(t_in[t].x), h2f_hi(t_in[t].x), h2f_lo(t_in[t].y));
                    const float wc = w8[t] * (kx * ky);
                    sum_w += wc;
                    s0 += wc * sc.x; s1 += wc * sc.y; s2 += wc * sc.z;
                    s3 += (wc * wc) * h2f_hi(t_in[t].y);
                }
            }
            else
            {
            for (int yy = -a.radius; yy <= a.radius; yy++)
                for (int xx = -a.radius; xx <= a.radius; xx++)
                {
                    const int px = x + xx * a.step, py = y + yy * a.step;
                    if (px < 0 || py < 0 || px >= a.w || py >= a.h || (xx == 0 && yy == 0)) continue;
                    const int   axx = xx < 0 ? -xx : xx, ayy = yy < 0 ? -yy : yy;
                    const float kx = axx == 0 ? 1.0f : (axx == 1 ? __fdiv_rn(2.0f, 3.0f) : __fdiv_rn(1.0f, 6.0f));
                    const float ky = ayy == 0 ? 1.0f : (ayy == 1 ? __fdiv_rn(2.0f, 3.0f) : __fdiv_rn(1.0f, 6.0f));
                    const uint2 s = a.in.raw(px, py), q2 = a.gb2.raw(px, py), q3 = a.gb3.raw(px, py);
                    const f3    sc = mk3(h2f_lo(s.x), h2f_hi(s.x), h2f_lo(s.y));
                    const float sl = luminance(sc);
                    const f3    sn = oct_decode(h2f_lo(q2.x), h2f_hi(q2.x));
                    const float wZ = det_exp(div_by(-fabsf(center_depth - h2f_hi(q3.y)), by_sigma));
                    const float wN = pow_phi_normal(clamp1(dot3(cn, sn), 0.0f, 1.0f), a.phi_normal);
                    const float wL = div_by(fabsf(center_luma - sl), phi_c);
                    const float w  = det_exp((0.0f - max2(wL, 0.0f)) - max2(wZ, 0.0f)) * wN;
                    const float wc = w * (kx * ky);
                    sum_w += wc;
                    s0 += wc * sc.x; s1 += wc * sc.y; s2 += wc * sc.z;
                    s3 += (wc * wc) * h2f_hi(s.y);
                }
            }
            result = make_uint2(pack_h2(__fdiv_rn(s0, sum_w), __fdiv_rn(s1, sum_w)), pack_h2(__fdiv_rn(s2, sum_w), __fdiv_rn(s3, sum_w * sum_w)));
        }
    }
    a.out[o] = result;
    if (a.out2) a.out2[o] = result;
}

// ------------------------------------------------------------------------------------------------
struct hr_reflections
{
    hr_ctx* ctx = nullptr;
    int     full_w = 0, full_h = 0, w = 0, h = 0, scale = 0, y0 = 0, y1 = 0, band_y0 = 0, band_y1 = 0, tiles_x = 0, tiles_y = 0;
    DevBuf  trace, color[2], moments[2], prev_image, atrous[2], upsample, tile_class, counters, ray_slots, geo[2];
    bool    first_frame = true, last_denoise = true, want_stats = false;
    int     read_idx = 0, last_pp = 0;
    bool    last_blur_as_input = false;   // which image the NEXT frame's temporal stage reads as colour history (hr_reflections_image 10)
    bool    fuse = true;   // tolerance mode: a-trous iterations 0 + 1 in one launch (developer A/B switch HR_FUSE=0, read once at create)
    StageProfiler prof;
    hipStream_t   last_stream = nullptr;
    // Tolerance mode: the temporal kernel writes an 8-byte geometry record per pixel {oct normal, mesh id | linear z} (copies of the
    // current G-buffer's words) into geo[geo_parity]: this frame's a-trous taps read it instead of two half-used G-buffer lines, and —
    // when the caller hands back as in->prev the images it passed as in->cur (the reference's ping-pong, g_buffer.cpp:208-211) — so does
    // the next frame's reprojection (see hr_shadows).
    bool          geo_history = true;   // developer A/B switch HR_GEO_HISTORY=0 (read once at create)
    bool          geo_valid = false;
    bool          dbg_require_geo = false;   // HR_DEBUG_REQUIRE_GEO (tests)
    int           geo_parity = 0;
    const void*   geo_gb2 = nullptr;
    const void*   geo_gb3 = nullptr;
    const void*   geo_cur = nullptr;    // this frame's records (a-trous stages), nullptr in the parity mode / for bands
    TileOrder     tile_order;           // heaviest-first launch order of the trace kernel (tile_order.h)
};

bool hr::profiling_enabled(const hr_reflections* p) { return p && p->prof.enabled; }

extern "C" {

void hr_reflections_default_params(hr_reflections_params* p)
{
    p->denoise = 1; p->sample_gi = 1; p->approximate_with_ddgi = 1; p->gi_intensity = 0.5f; p->rough_ddgi_intensity = 0.5f;
    p->ibl_indirect_specular_intensity = 0.05f; p->bias = 0.5f; p->trim = 0.8f; p->alpha = 0.01f; p->moments_alpha = 0.2f; p->blur_as_input = 0;
    p->phi_color = 10.0f; p->phi_normal = 32.0f; p->sigma_depth = 1.0f; p->radius = 1; p->filter_iterations = 4; p->feedback_iteration = 1;
    p->camera_delta[0] = p->camera_delta[1] = p->camera_delta[2] = 0.0f; p->frame_time = 0.0f; p->exact = 1;
}

hr_status hr_reflections_create(hr_ctx* ctx, int32_t full_width, int32_t full_height, hr_scale scale, const hr_band* band, hr_reflections** out)
{
    HR_CHECK_ARG(ctx && out && full_width > 0 && full_height > 0 && (int)scale >= 0 && (int)scale <= 2);
    HR_HIP(hipSetDevice(ctx->device));
    hr_reflections* p = new hr_reflections();
    p->ctx = ctx; p->full_w = full_width; p->full_h = full_height; p->scale = (int)scale;
    if (const char* e = getenv("HR_FUSE")) p->fuse = atoi(e) != 0;
    if (const char* e = getenv("HR_GEO_HISTORY")) p->geo_history = atoi(e) != 0;
    if (const char* e = getenv("HR_DEBUG_REQUIRE_GEO")) p->dbg_require_geo = atoi(e) != 0;   // test switch, see hr_shadows
    if (const char* e = getenv("HR_TILE_ORDER")) p->tile_order.enabled = atoi(e) != 0;
    p->tile_order.tag = "reflections";
    p->w = full_width >> (int)scale; p->h = full_height >> (int)scale; p->y0 = 0; p->y1 = p->h;
    if (band && band->band_y1 > band->band_y0)
    {
        p->band_y0 = band->band_y0; p->band_y1 = band->band_y1;
        p->y0 = band->band_y0 - band->halo < 0 ? 0 : band->band_y0 - band->halo;
        p->y1 = band->band_y1 + band->halo > p->h ? p->h : band->band_y1 + band->halo;
        if ((p->y0 & 7) || ((p->y1 & 7) && p->y1 != p->h)) { set_last_error("band rows must be multiples of 8"); delete p; return HR_ERR_INVALID_ARG; }
    }
    p->tiles_x = cdiv(p->w, 8); p->tiles_y = cdiv(p->h, 8);
    const size_t px = (size_t)p->w * p->h;
    hr_status s;
#define A(buf, n) if ((s = p->buf.alloc(n)) != HR_OK) { delete p; return s; }
    A(trace, px * 8) A(color[0], px * 8) A(color[1], px * 8) A(moments[0], px * 8) A(moments[1], px * 8) A(prev_image, px * 8)
    A(atrous[0], px * 8) A(atrous[1], px * 8) A(upsample, (size_t)full_width * full_height * 8) A(tile_class, (size_t)p->tiles_x * p->tiles_y) A(counters, 64) A(ray_slots, (size_t)p->tiles_x * p->tiles_y * 4)
    if (p->geo_history) { A(geo[0], px * 8) A(geo[1], px * 8) }   // (round 5: bands too — a band computes, and so records, every row it reads history from: history_halo == halo)
#undef A
    if ((s = p->tile_order.init(p->tiles_x * (cdiv(p->y1, 8) - p->y0 / 8))) != HR_OK) { delete p; return s; }
    HR_HIP(hipMemset(p->counters.p, 0, 64));
    HR_HIP(hipMemset(p->ray_slots.p, 0, p->ray_slots.bytes));
    *out = p;
    return HR_OK;
}

hr_status hr_reflections_destroy(hr_reflections* p)
{
    if (!p) return HR_OK;
    (void)hipSetDevice(p->ctx->device);
    (void)hipDeviceSynchronize();
    delete p;
    return HR_OK;
}
hr_status hr_reflections_reset_history(hr_reflections* p) { HR_CHECK_ARG(p); p->first_frame = true; p->geo_valid = false; p->tile_order.invalidate(); return HR_OK; }
hr_status hr_reflections_history_apron_exceeded(hr_reflections* p, int32_t* exceeded)   // see hr_shadows_history_apron_exceeded
{
    HR_CHECK_ARG(p && exceeded);
    uint32_t v = 0;
    HR_HIP(hipStreamSynchronize(p->last_stream));
    HR_HIP(hipMemcpy(&v, (char*)p->counters.p + 48, 4, hipMemcpyDeviceToHost));
    if (v) HR_HIP(hipMemset((char*)p->counters.p + 48, 0, 4));
    *exceeded = v ? 1 : 0;
    return HR_OK;
}

hr_status hr_reflections_set_profiling(hr_reflections* p, int32_t e) { HR_CHECK_ARG(p); p->prof.enabled = e != 0; return HR_OK; }
hr_status hr_reflections_get_stage_times(hr_reflections* p, hr_stage_times* out) { HR_CHECK_ARG(p && out); p->prof.collect(out); return HR_OK; }
hr_status hr_reflections_ray_count(hr_reflections* p, uint64_t* rays)
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

hr_status hr_reflections_ray_trace(hr_reflections* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_environment* env, hr_ddgi* ddgi,
                                   const hr_reflections_params* prm, void* stream_)
{
    HR_CHECK_ARG(p && scene && in && env && ddgi && prm && env->sky && env->sky_size > 0);
    HR_CHECK_ARG(in->cur.depth && in->cur.gb2 && in->cur.gb3 && in->cur.width == p->w && in->cur.height == p->h && in->sobol && in->scrambling_ranking);
    if (prm->sample_gi) HR_CHECK_ARG(env->prefiltered && env->prefiltered_levels > 0 && env->brdf_lut && env->brdf_lut_size > 0);
    hipStream_t st = (hipStream_t)stream_;
    p->last_stream = st;
    const int pp = in->ping_pong ? 1 : 0;
    if (p->first_frame)
    {
        // clear_images() (:962-991)
        HR_HIP(hipMemsetAsync(p->prev_image.p, 0, p->prev_image.bytes, st));
        HR_HIP(hipMemsetAsync(p->color[!pp].p, 0, p->color[0].bytes, st));
        HR_HIP(hipMemsetAsync(p->moments[!pp].p, 0, p->moments[0].bytes, st));
        p->first_frame = false;
    }
    ReflTraceArgs a;
    hr_status s = hr_ddgi_get_uniforms(ddgi, &a.d);
    if (s != HR_OK) return s;
    hr_image_view iv, dv;
    if ((s = hr_ddgi_current_read(ddgi, &iv, &dv)) != HR_OK) return s;
    a.irr = AtlasRGBA { (const uint2*)iv.data, iv.width, iv.height };
    a.dep = AtlasRG { (const uint32_t*)dv.data, dv.width, dv.height };
    a.light = in->ubo.light;
    for (int i = 0; i < 16; i++) a.vpi[i] = in->ubo.view_proj_inverse[i];
    for (int i = 0; i < 3; i++) a.cam[i] = in->ubo.cam_pos[i];
    a.depth = in->cur.depth; a.gb2 = (const uint2*)in->cur.gb2; a.gb3 = (const uint2*)in->cur.gb3;
    a.sobol = in->sobol; a.sr = in->scrambling_ranking;
    a.nodes = (const Node8*)scene->nodes.p; a.tris = (const TriGPU*)scene->tris.p;
    scene_shading_from(scene, a.sh);
    a.env.sky = CubeMap { (const uint2*)env->sky, env->sky_size };
    a.env.prefiltered = (const uint2*)env->prefiltered; a.env.pre_size = env->prefiltered_size; a.env.pre_levels = env->prefiltered_levels;
    a.env.lut = (const uint32_t*)env->brdf_lut; a.env.lut_size = env->brdf_lut_size;
    a.out = (uint2*)p->trace.p; a.ray_slots = (uint32_t*)p->ray_slots.p + (size_t)(p->y0 / 8) * p->tiles_x;
    a.w = p->w; a.h = p->h; a.y0 = p->y0; a.y1 = p->y1;
    a.tile_y0 = p->y0 / 8; a.tiles_x = p->tiles_x; a.tiles_y = cdiv(p->y1, 8) - a.tile_y0;
    a.bias = prm->bias; a.trim = prm->trim; a.num_frames = in->num_frames;
    a.sample_gi = prm->sample_gi ? 1 : 0; a.approximate_with_ddgi = prm->approximate_with_ddgi ? 1 : 0;
    a.gi_intensity = prm->gi_intensity; a.rough_ddgi_intensity = prm->rough_ddgi_intensity; a.ibl_intensity = prm->ibl_indirect_specular_intensity;
    const uint64_t px = (uint64_t)p->w * (p->y1 - p->y0);
    a.stats = nullptr;
    const int n_tiles = a.tiles_x * a.tiles_y;
    if ((s = p->tile_order.flush(st)) != HR_OK) return s;   // last launch's costs, if no temporal stage took them along
    a.order = p->tile_order.order_arg(n_tiles); a.cost = p->tile_order.cost_arg(n_tiles);
    if (p->want_stats)
    {
        a.cost = nullptr;
        // instrumented build of the same kernel (hr_reflections_trace_stats): counters + 8 .. 32
        HR_HIP(hipMemsetAsync((char*)p->counters.p + 8, 0, 24, st));
        a.stats = (unsigned long long*)((char*)p->counters.p + 8);
        hipLaunchKernelGGL((k_refl_trace<true>), dim3(cdiv(a.tiles_x * a.tiles_y, REFL_TRACE_WAVES)), dim3(64 * REFL_TRACE_WAVES), 0, st, a);
        HR_HIP(hipGetLastError());
        return HR_OK;
    }
    int ev = p->prof.begin("ray_trace", st, px * 28);
    hipLaunchKernelGGL((k_refl_trace<false>), dim3(cdiv(a.tiles_x * a.tiles_y, REFL_TRACE_WAVES)), dim3(64 * REFL_TRACE_WAVES), 0, st, a);
    p->prof.end(ev, st);
    HR_HIP(hipGetLastError());
    if (a.cost)
    {
        if ((s = p->tile_order.traced(n_tiles, st)) != HR_OK) return s;
    }
    return HR_OK;
}

// Instrumented ray trace (same rays, same trace image): out3 = rays traced (reflection rays + the hit shader's light rays), BVH node
// steps, triangle tests — the BVH term of SURVEY §8d's algorithmic bytes.
hr_status hr_reflections_trace_stats(hr_reflections* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_environment* env, hr_ddgi* ddgi,
                                     const hr_reflections_params* prm, uint64_t* out3, void* stream)
{
    HR_CHECK_ARG(p && out3);
    const bool first = p->first_frame;
    p->want_stats = true;
    const hr_status s = hr_reflections_ray_trace(p, scene, in, env, ddgi, prm, stream);
    p->want_stats = false;
    p->first_frame = first;   // a statistics pass is not a frame: the next render() still clears the history as it would have
    if (s != HR_OK) return s;
    HR_HIP(hipStreamSynchronize((hipStream_t)stream));
    uint64_t host[3];
    HR_HIP(hipMemcpy(host, (char*)p->counters.p + 8, 24, hipMemcpyDeviceToHost));
    out3[0] = host[2]; out3[1] = host[0]; out3[2] = host[1];
    return HR_OK;
}

hr_status hr_reflections_temporal(hr_reflections* p, const hr_frame_inputs* in, const hr_reflections_params* prm, void* stream_)
{
    HR_CHECK_ARG(p && in && prm && in->cur.depth && in->cur.gb2 && in->cur.gb3 && in->prev.depth && in->prev.gb2 && in->prev.gb3);
    HR_CHECK_ARG(in->cur.width == p->w && in->prev.width == p->w && in->cur.height == p->h && in->prev.height == p->h);
    hipStream_t st = (hipStream_t)stream_;
    p->last_stream = st;
    const int pp = in->ping_pong ? 1 : 0, w = p->w, y0 = p->y0, y1 = p->y1;
    ReflTemporalArgs a;
    for (int i = 0; i < 16; i++) { a.vpi[i] = in->ubo.view_proj_inverse[i]; a.pvp[i] = in->ubo.prev_view_proj[i]; }
    for (int i = 0; i < 3; i++) a.cam[i] = in->ubo.cam_pos[i];
    auto img = [&](const void* ptr) { return ImgRGBA16F { (const uint2*)ptr, w, y0, y1 }; };
    a.in = img(p->trace.p); a.gb2 = img(in->cur.gb2); a.gb3 = img(in->cur.gb3); a.pgb2 = img(in->prev.gb2); a.pgb3 = img(in->prev.gb3);
    a.hist = img(prm->blur_as_input ? p->prev_image.p : p->color[!pp].p); // :1124
    a.hist_moments = img(p->moments[!pp].p);
    a.depth = ImgR32F { in->cur.depth, w, y0, y1 }; a.pdepth = ImgR32F { in->prev.depth, w, y0, y1 };
    a.out = (uint2*)p->color[pp].p; a.out_moments = (uint2*)p->moments[pp].p; a.tile_class = (uint8_t*)p->tile_class.p;
    a.w = w; a.h = p->h; a.y0 = y0; a.y1 = y1; a.tiles_x = p->tiles_x;
    a.apron_flag = (y0 > 0 || y1 < p->h) ? (uint32_t*)((char*)p->counters.p + 48) : nullptr;   // row bands only
    a.band_y0 = p->band_y0; a.band_y1 = p->band_y1;
    a.alpha = prm->alpha; a.moments_alpha = prm->moments_alpha; a.approximate_with_ddgi = prm->approximate_with_ddgi ? 1 : 0;
    const float* cd = prm->camera_delta;
    a.moving = (sqrtf((cd[0] * cd[0] + cd[1] * cd[1]) + cd[2] * cd[2]) > 0.0f) ? 1 : 0; // compute_max_accumulated_frame :162-168
    a.geo_hist = nullptr; a.geo_out = nullptr; p->geo_cur = nullptr;
    if (!prm->exact && p->geo[0].p)
    {
        const bool had_records = p->geo_valid;
        if (p->geo_valid && !p->first_frame && in->prev.gb2 == p->geo_gb2 && in->prev.gb3 == p->geo_gb3 && in->prev.gb2 != in->cur.gb2 && in->prev.gb3 != in->cur.gb3) a.geo_hist = p->geo[p->geo_parity].p;
        p->geo_parity ^= 1;
        a.geo_out = p->geo[p->geo_parity].p;
        p->geo_cur = a.geo_out;
        p->geo_valid = true; p->geo_gb2 = in->cur.gb2; p->geo_gb3 = in->cur.gb3;
        if (p->dbg_require_geo && had_records && !a.geo_hist) { hr::set_last_error("hr_reflections_temporal: HR_DEBUG_REQUIRE_GEO is set and the record path was not taken"); return HR_ERR_INVALID_ARG; }
    }
    else p->geo_valid = false;
    p->last_pp = pp;
    p->last_blur_as_input = prm->blur_as_input != 0;
    const uint64_t px = (uint64_t)w * (y1 - y0);
    a.sort = TileSortArgs { nullptr, nullptr, 0, 0, 0, 0 };
    int ev = p->prof.begin("temporal_accumulation", st, px * 80);
    if (prm->exact) hipLaunchKernelGGL(k_refl_temporal, dim3(cdiv(w, RT_TW), cdiv(y1 - y0, RT_TH)), dim3(256), 0, st, a);
    else
    {
        a.sort = p->tile_order.ride();   // the trace kernel's next launch order rides along (tile_order.h)
        launch_refl_temporal_fast(a, st);
    }
    p->prof.end(ev, st);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

hr_status hr_reflections_atrous_iteration(hr_reflections* p, const hr_frame_inputs* in, const hr_reflections_params* prm, int32_t i, void* stream_)
{
    HR_CHECK_ARG(p && in && prm && i >= 0 && i < prm->filter_iterations && prm->filter_iterations <= 8 && prm->radius >= 0 && prm->radius <= 2);
    hipStream_t st = (hipStream_t)stream_;
    p->last_stream = st;
    const int read_idx = i & 1, write_idx = (i & 1) ^ 1, w = p->w, y0 = p->y0, y1 = p->y1;
    ReflAtrousArgs a;
    auto img = [&](const void* ptr) { return ImgRGBA16F { (const uint2*)ptr, w, y0, y1 }; };
    a.in = img(i == 0 ? p->color[p->last_pp].p : p->atrous[read_idx].p);
    a.gb2 = img(in->cur.gb2); a.gb3 = img(in->cur.gb3); a.depth = ImgR32F { in->cur.depth, w, y0, y1 };
    a.geo = prm->exact ? nullptr : p->geo_cur;
    a.tile_class = (const uint8_t*)p->tile_class.p;
    a.out = (uint2*)p->atrous[write_idx].p;
    a.out2 = (prm->feedback_iteration == i && prm->blur_as_input) ? (uint2*)p->prev_image.p : nullptr; // :1218
    a.w = w; a.h = p->h; a.y0 = y0; a.y1 = y1; a.tiles_x = p->tiles_x; a.radius = prm->radius; a.step = 1 << i;
    a.phi_color = prm->phi_color; a.phi_normal = prm->phi_normal; a.sigma_depth = prm->sigma_depth;
    a.approximate_with_ddgi = prm->approximate_with_ddgi ? 1 : 0;
    p->read_idx = write_idx;
    static const char* names[8] = { "atrous_0", "atrous_1", "atrous_2", "atrous_3", "atrous_4", "atrous_5", "atrous_6", "atrous_7" };
    const uint64_t px = (uint64_t)w * (y1 - y0);
    int ev = p->prof.begin(names[i], st, px * 36);
    if (!prm->exact) launch_refl_atrous_fast(a, st);
    else if (a.radius == 1) hipLaunchKernelGGL(k_refl_atrous<1>, dim3(cdiv(w, 32), cdiv(y1 - y0, 8)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_refl_atrous<-1>, dim3(cdiv(w, 32), cdiv(y1 - y0, 8)), dim3(256), 0, st, a);
    p->prof.end(ev, st);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

// tolerance mode, radius 1: iterations 0 and 1 in one launch (kf_refl_atrous01; iteration 0's image stays in LDS)
static hr_status reflections_atrous01(hr_reflections* p, const hr_frame_inputs* in, const hr_reflections_params* prm, void* stream_, bool* done)
{
    hipStream_t st = (hipStream_t)stream_;
    p->last_stream = st;
    const int w = p->w, y0 = p->y0, y1 = p->y1;
    ReflAtrousArgs a;
    auto img = [&](const void* ptr) { return ImgRGBA16F { (const uint2*)ptr, w, y0, y1 }; };
    a.in = img(p->color[p->last_pp].p);
    a.gb2 = img(in->cur.gb2); a.gb3 = img(in->cur.gb3); a.depth = ImgR32F { in->cur.depth, w, y0, y1 };
    a.geo = nullptr;   // the fused kernel stages every texel's full GB3 word pair (roughness) anyway
    a.tile_class = (const uint8_t*)p->tile_class.p;
    a.out = (uint2*)p->atrous[0].p;
    a.out2 = (prm->feedback_iteration == 1 && prm->blur_as_input) ? (uint2*)p->prev_image.p : nullptr;
    uint2* first2 = (prm->feedback_iteration == 0 && prm->blur_as_input) ? (uint2*)p->prev_image.p : nullptr;
    a.w = w; a.h = p->h; a.y0 = y0; a.y1 = y1; a.tiles_x = p->tiles_x; a.radius = prm->radius; a.step = 1;
    a.phi_color = prm->phi_color; a.phi_normal = prm->phi_normal; a.sigma_depth = prm->sigma_depth;
    a.approximate_with_ddgi = prm->approximate_with_ddgi ? 1 : 0;
    const uint64_t px = (uint64_t)w * (y1 - y0);
    // algorithmic bytes: in 8 + GB2 8 + GB3 8 + depth 4 + out 8 — the G-buffer counts once for the two iterations
    int ev = p->prof.begin("atrous_01", st, px * 36);
    *done = launch_refl_atrous01_fast(a, first2, st);
    p->prof.end(ev, st);
    if (*done) p->read_idx = 0;
    HR_HIP(hipGetLastError());
    return HR_OK;
}

hr_status hr_reflections_upsample(hr_reflections* p, const hr_frame_inputs* in, const hr_reflections_params* prm, void* stream_)
{
    HR_CHECK_ARG(p && in && prm);
    if (p->scale == 0) return HR_OK;
    HR_CHECK_ARG(in->cur_full.gb2 && in->cur_full.gb3 && in->cur_full.width == p->full_w && in->cur_full.height == p->full_h);
    hipStream_t st = (hipStream_t)stream_;
    p->last_stream = st;
    UpsampleArgs a;
    a.W = p->full_w; a.H = p->full_h; a.w = p->w; a.h = p->h;
    a.G2 = (const uint2*)in->cur_full.gb2; a.G3 = (const uint2*)in->cur_full.gb3; a.g2 = (const uint2*)in->cur.gb2; a.g3 = (const uint2*)in->cur.gb3;
    a.in = p->atrous[p->read_idx].p; a.in_channels = 4; a.channels = 4; a.out = p->upsample.p; a.sky_value = 0.0f; a.power = 0.0f;
    const uint64_t PX = (uint64_t)p->full_w * p->full_h, px = (uint64_t)p->w * p->h;
    int ev = p->prof.begin("upsample", st, PX * 24 + px * 24);
    if (prm->exact) launch_upsample(a, st);
    else launch_upsample_fast(a, st);
    p->prof.end(ev, st);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

// Everything of RayTracedReflections::render after the ray trace (ray_traced_reflections.cpp:113-122); see hr_shadows_denoise
hr_status hr_reflections_denoise(hr_reflections* p, const hr_frame_inputs* in, const hr_reflections_params* prm, void* stream)
{
    HR_CHECK_ARG(p && in && prm);
    p->last_denoise = prm->denoise != 0;
    if (!prm->denoise) return HR_OK;
    hr_status s;
    if ((s = hr_reflections_temporal(p, in, prm, stream)) != HR_OK) return s;
    {
        HR_SCOPED_SAMPLE("A-Trous Filter");   // ray_traced_reflections.cpp:1145
        bool fused = false;
        if (!prm->exact && p->fuse && prm->filter_iterations >= 2 && prm->filter_iterations <= 8 && prm->radius == 1 &&
            (s = reflections_atrous01(p, in, prm, stream, &fused)) != HR_OK) return s;
        for (int i = fused ? 2 : 0; i < prm->filter_iterations; i++)
            if ((s = hr_reflections_atrous_iteration(p, in, prm, i, stream)) != HR_OK) return s;
    }
    if (p->scale != 0 && (s = hr_reflections_upsample(p, in, prm, stream)) != HR_OK) return s;
    return HR_OK;
}

hr_status hr_reflections_render(hr_reflections* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_environment* env, hr_ddgi* ddgi,
                                const hr_reflections_params* prm, void* stream)
{
    HR_SCOPED_SAMPLE("Ray Traced Reflections");
    HR_CHECK_ARG(p && scene && in && env && ddgi && prm);
    HR_HIP(hipSetDevice(p->ctx->device));
    p->prof.begin_frame();
    p->last_denoise = prm->denoise != 0;
    hr_status s = hr_reflections_ray_trace(p, scene, in, env, ddgi, prm, stream);
    if (s != HR_OK) return s;
    return hr_reflections_denoise(p, in, prm, stream);
}

static void fill_view(hr_image_view* v, void* data, int w, int h, int bpp, hr_format f)
{
    v->data = data; v->width = w; v->height = h; v->row_pitch_bytes = w * bpp; v->format = f;
}

hr_status hr_reflections_image(hr_reflections* p, int32_t which, hr_image_view* v)
{
    HR_CHECK_ARG(p && v);
    void* ptr = nullptr;
    switch (which)
    {
        case 0: ptr = p->trace.p; break;
        case 1: ptr = p->color[0].p; break;
        case 2: ptr = p->color[1].p; break;
        case 3: ptr = p->moments[0].p; break;
        case 4: ptr = p->moments[1].p; break;
        case 5: ptr = p->prev_image.p; break;
        // 10: the colour history the next frame's temporal stage will read — the feedback image with blur_as_input, else the temporal
        // output written this frame (ray_traced_reflections.cpp:1124,1218): what a row-tiled host exchanges with its neighbours
        case 10: ptr = p->last_blur_as_input ? p->prev_image.p : p->color[p->last_pp].p; break;
        case 6: ptr = p->atrous[0].p; break;
        case 7: ptr = p->atrous[1].p; break;
        case 8: fill_view(v, p->upsample.p, p->full_w, p->full_h, 8, HR_FORMAT_RGBA16F); return HR_OK;
        case 9: fill_view(v, p->tile_class.p, p->tiles_x, p->tiles_y, 1, (hr_format)0); return HR_OK;
        default: set_last_error("hr_reflections_image: unknown image index"); return HR_ERR_INVALID_ARG;
    }
    fill_view(v, ptr, p->w, p->h, 8, HR_FORMAT_RGBA16F);
    return HR_OK;
}

hr_status hr_reflections_output(hr_reflections* p, hr_output_kind kind, hr_image_view* v)
{
    HR_CHECK_ARG(p && v);
    if (!p->last_denoise || kind == HR_OUTPUT_RAY_TRACE) return hr_reflections_image(p, 0, v);
    if (kind == HR_OUTPUT_TEMPORAL_ACCUMULATION) return hr_reflections_image(p, 1 + p->last_pp, v);
    if (kind == HR_OUTPUT_ATROUS || p->scale == 0) return hr_reflections_image(p, 6 + p->read_idx, v);
    return hr_reflections_image(p, 8, v);
}

} // extern "C"
