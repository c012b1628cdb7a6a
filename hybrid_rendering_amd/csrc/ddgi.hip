// DDGI on MI355X — HIP replacement for src/ddgi.{h,cpp} and src/shaders/gi/*.
//   ray_trace()         ddgi.cpp:767-825, gi_ray_trace.{rgen:78-100, rchit:95-128, rmiss:24-27} -> k_ddgi_trace
//   probe_update()      :829-900, gi_probe_update.glsl:136-184 (irradiance + depth variants)     -> k_ddgi_probe_update (one launch for
//   border_update()     :904-939, gi_border_update.glsl:151-175                                     both atlases and their borders)
//   sample_probe_grid() :943-986, gi_sample_probe_grid.comp:75-99                                -> k_ddgi_sample
#include "hr_internal.h"
#include "shading.h"
#include "pass_args.h"
#ifdef HR_DEV_PATHS
#include "trace_queue.h"
#endif

using namespace hr;

// gi_ray_trace.rgen:61-72.  PHI = sqrt(5)*0.5+0.5 evaluated in fp32; PHI - 1 is exact.
HR_DEV f3 spherical_fibonacci(float i, float n)
{
    const float PHI_M1 = 0.61803400516510009765625f;
    const float a   = i * PHI_M1;
    const float phi = 2.0f * HR_M_PI * (a - floorf(a));
    const float ct  = 1.0f - (2.0f * i + 1.0f) * __fdiv_rn(1.0f, n);
    const float st  = hr_sqrt(clamp1(1.0f - ct * ct, 0.0f, 1.0f));
    float s, c;
    det_sincos(phi, s, c);
    return mk3(c * st, s * st, ct);
}

struct DDGITraceArgs
{
    DDGIU         d;
    hr_light      light;
    float         orientation[9];
    const Node8*  nodes;
    const TriGPU* tris;
    SceneShading  sh;
    CubeMap       sky;
    AtlasRGBA     prev_irr;
    AtlasRG       prev_depth;
    uint2*        radiance;
    uint2*        dirdist;
    uint32_t*     ray_slots;  // rays per wave
    uint32_t      num_frames;
    int           infinite_bounces;
    float         gi_intensity;
    int           n_probes;      // one past the last probe traced
    int           probe_begin;   // first probe traced (probe shard: z-slabs of the grid)
    unsigned long long* stats;   // instrumented build only (k_ddgi_trace<true>): [0] node steps, [1] triangle tests, [2] rays
};

// one thread per (probe, ray); a wave covers 64 consecutive rays of one probe (same origin: the rays share the nodes around
// it — the transposed mapping, one direction from 64 probes per wave, measured 0.40 -> 0.45 ms)
#ifdef HR_TRACE_DIVERGENCE
static __device__ unsigned long long g_div_ddgi[16];   // 0-7 primary rays, 8-15 secondary rays
extern "C" int hr_debug_divergence_ddgi(uint64_t* out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_div_ddgi), sizeof(g_div_ddgi)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_div_ddgi), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif
#ifndef DDGI_TRACE_WAVES
#define DDGI_TRACE_WAVES 1
#endif
#ifndef DDGI_SEQ
#define DDGI_SEQ 0   // 1: the two visibility rays of a hit point as ONE lane-sequential wave loop (traverse.h trace_any_seq) instead of two wave-level
                     // traversals.  Measured (round 3, 16x8x16x256, bit-identical atlases): 322.4 us vs 324.3 — the ray LENGTHS are heavy-tailed, one
                     // straggler ray per wave sets the wave's time whether the other lanes' rays are summed or not; kept as a measured A/B path
#endif
#ifndef DDGI_COOP
#define DDGI_COOP 1   // wave-cooperative triangle tests (traverse.h trace_coop); 0 = the per-lane loops
#endif
#ifndef DDGI_TRACE_EU
#define DDGI_TRACE_EU 6   // minimum waves per SIMD the register allocator must leave room for: 1 / 6 / 7 -> 325 / 318 / 338 us
#endif
// STATS: the instrumented build behind hr_ddgi_trace_stats — per-lane closest-hit walk (same hits as the cooperative one), every node
// step and triangle test of the probe rays and of the hit shader's light / sky rays counted (the BVH term of the
// algorithmic bytes).  The product launches <false>.
template <bool STATS>
__global__ __launch_bounds__(64 * DDGI_TRACE_WAVES, DDGI_TRACE_EU) void k_ddgi_trace(DDGITraceArgs a)
{
    __shared__ uint32_t s_stack[DDGI_TRACE_WAVES][HR_STACK_ENTRIES * 64];
#if DDGI_COOP
    __shared__ CoopWave s_coop[DDGI_TRACE_WAVES];
#endif
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int R = a.d.rays_per_probe;
    const long long gid = (long long)blockIdx.x * (64 * DDGI_TRACE_WAVES) + threadIdx.x;
    const int probe = a.probe_begin + (int)(gid / R), ray = (int)(gid % R);
    uint32_t  rays = 0;
    HR_DIV(DivCounters dvp = {}, dvs = {};)
    const bool valid = probe < a.n_probes;
    f3 origin = mk3(0.0f, 0.0f, 0.0f), dir = mk3(0.0f, 0.0f, 1.0f);
    if (valid)
    {
        origin = probe_location(a.d, probe);
        const f3 f      = spherical_fibonacci((float)ray, (float)R);
        const float* M  = a.orientation;
        dir = normalize3(mk3((M[0] * f.x + M[3] * f.y) + M[6] * f.z, (M[1] * f.x + M[4] * f.y) + M[7] * f.z, (M[2] * f.x + M[5] * f.y) + M[8] * f.z));
        rays++;
    }
    uint32_t st_n = 0, st_t = 0;
#if DDGI_COOP
    HitRec h;
    if (!STATS) h = trace_coop<false>(valid, a.nodes, a.tris, origin, dir, 0.001f, 10000.0f, s_stack[wave], s_coop[wave], lane, 0u HR_DIV(, &dvp));
    else
    {
        h.prim = -1;
        if (valid) h = trace_closest<STATS>(a.nodes, a.tris, origin, dir, 0.001f, 10000.0f, s_stack[wave], lane, nullptr, &st_n, &st_t);
    }
#else
    HitRec h;
    h.prim = -1;
    DivCounters* dvp_ptr = nullptr;   // (without HR_TRACE_DIVERGENCE the HR_DIV() argument vanished and &st_n slid into the `dv` slot)
    HR_DIV(dvp_ptr = &dvp;)
    if (valid) h = trace_closest<STATS>(a.nodes, a.tris, origin, dir, 0.001f, 10000.0f, s_stack[wave], lane, dvp_ptr, &st_n, &st_t);
#endif
    if (valid)
    {
        Rng   rng = rng_init((uint32_t)ray, (uint32_t)probe, a.num_frames);
        f3    L;
        float hit_distance = 10000.0f;
#ifdef HR_ABL_DDGI_PRIMARY_ONLY   // developer ablation (tools/ablate.sh): what does each part of the kernel cost?
        if (true) { L = mk3(h.t, h.u, h.v); hit_distance = h.t; }
        else
#endif
        if (h.prim < 0) L = a.sky.fetch(dir);
        else
        {
            const SurfaceHit s = surface_at(a.sh, h);
            const f3 Wo = neg3(dir);
            const f3 F0 = mix3(mk3(0.04f, 0.04f, 0.04f), s.albedo, s.metallic);
            const f3 c_diffuse = mix3(mul3(s.albedo, sub3(one3(), F0)), mk3(0.0f, 0.0f, 0.0f), s.metallic);
            const float r2x = next_float(rng), r2y = next_float(rng);
            TraceCtx tc { a.nodes, a.tris, s_stack[wave], lane };
            HR_DIV(tc.dv = &dvs;)
#if DDGI_SEQ && !defined(HR_DEV_PATHS)
#error "-DDDGI_SEQ=1 is an A/B path: build with -DHR_DEV_PATHS"
#endif
#if DDGI_SEQ
            // the light ray and the sky ray of the hit point back to back per lane inside ONE wave loop (traverse.h trace_any_seq) instead of
            // two wave-level traversals; the visibilities enter direct_lighting's result exactly as in shading.h DirectSplit
            const DirectSplit ds = direct_lighting_split(a.light, Wo, s.N, s.P, F0, c_diffuse, s.roughness, one3(), true, r2x, r2y, a.sky);
            const f3    sdir[2] = { ds.ray1 ? ds.Wi1 : ds.Wi2, ds.Wi2 };
            const float stm[2]  = { ds.ray1 ? ds.t_max1 : 10000.0f, 10000.0f };
            const int   nsec    = ds.ray1 ? 2 : 1;
            rays += (uint32_t)nsec;
            const uint32_t occ = trace_any_seq<2>(true, nsec, a.nodes, a.tris, ds.origin, sdir, 0.01f, stm, s_stack[wave], lane, 0u HR_DIV(, &dvs));
            const bool o1 = ds.ray1 && (occ & 1u), o2 = ds.ray1 ? ((occ >> 1) & 1u) != 0u : (occ & 1u) != 0u;
            f3 Lo = o1 ? mk3(0.0f, 0.0f, 0.0f) : ds.P1;
            if (!o2) Lo = add3(Lo, ds.P2);
#else
            f3 Lo = direct_lighting<STATS>(tc, a.light, Wo, s.N, s.P, F0, c_diffuse, s.roughness, one3(), true, r2x, r2y, a.sky, rays);
            if (STATS) { st_n += tc.nn; st_t += tc.nt; }
#endif
#ifndef HR_ABL_DDGI_NO_IRRADIANCE
            if (a.infinite_bounces == 1)
#else
            if (false)
#endif
            {
                const f3 F   = fresnel_schlick_roughness(max2(dot3(s.N, Wo), 0.0f), F0, s.roughness);
                const f3 kD  = scale3(sub3(one3(), F), 1.0f - s.metallic);
                const f3 irr = sample_irradiance(a.d, s.P, s.N, Wo, a.prev_irr, a.prev_depth);
                Lo = add3(Lo, mul3(mul3(scale3(kD, a.gi_intensity), c_diffuse), irr));
            }
            L = Lo;
            hit_distance = 0.001f + h.t;
        }
        const size_t o = (size_t)probe * R + ray;
#if defined(HR_TRACE_DIVERGENCE) && defined(HR_DDGI_DUMP_STEPS)   // developer study (tools/ddgi_sort_study.py): node steps of the primary ray in the unused .w
        a.radiance[o] = make_uint2(pack_h2(L.x, L.y), pack_h2(L.z, (float)dvp.lane_nodes));
#else
        a.radiance[o] = make_uint2(pack_h2(L.x, L.y), pack_h2(L.z, 0.0f));
#endif
        a.dirdist[o]  = make_uint2(pack_h2(dir.x, dir.y), pack_h2(dir.z, hit_distance));
    }
    HR_DIV(div_flush(dvp, g_div_ddgi); div_flush(dvs, g_div_ddgi + 8);)
    for (int o = 32; o > 0; o >>= 1) rays += __shfl_down(rays, o);
    if (lane == 0) a.ray_slots[blockIdx.x * DDGI_TRACE_WAVES + wave] = rays;
    if (STATS)
    {
        for (int o = 32; o > 0; o >>= 1) { st_n += __shfl_down(st_n, o); st_t += __shfl_down(st_t, o); }
        if (lane == 0) { atomicAdd(a.stats + 0, (unsigned long long)st_n); atomicAdd(a.stats + 1, (unsigned long long)st_t); atomicAdd(a.stats + 2, (unsigned long long)rays); }
    }
}

#ifdef HR_DEV_PATHS   // A/B path that lost (>= 500 us against 333, docs/EXPERIMENTS.md 4.3); build with HR_CFLAGS=-DHR_DEV_PATHS, select with HR_DDGI_WAVEFRONT=1
// ---- wavefront form of ray_trace() (trace_queue.h): gen -> closest-hit queue -> shade + secondary queue -> any-hit queue -> combine
struct DDGIWaveArgs
{
    DDGITraceArgs  t;
    RayRec*        rays;        // primary rays, index = (probe - probe_begin) * R + ray
    const float4*  hits;
    RayRec*        sec_rays;    // secondary queue (light rays and sky rays of the hit points)
    uint32_t*      sec_count;
    const uint8_t* sec_occluded;
    uint4*         part;        // 3 x uint4 per primary ray: P1.xyz, P2.x | P2.yz, I.xy | I.z, slot1, slot2, flags
};
#define DDGI_NO_SLOT 0xffffffffu

__global__ __launch_bounds__(256) void k_ddgi_gen(DDGIWaveArgs w)
{
    const DDGITraceArgs& a = w.t;
    const int R = a.d.rays_per_probe;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int probe = a.probe_begin + (int)(gid / R), ray = (int)(gid % R);
    if (probe >= a.n_probes) return;
    const f3 origin = probe_location(a.d, probe);
    const f3 f      = spherical_fibonacci((float)ray, (float)R);
    const float* M  = a.orientation;
    const f3 dir = normalize3(mk3((M[0] * f.x + M[3] * f.y) + M[6] * f.z, (M[1] * f.x + M[4] * f.y) + M[7] * f.z, (M[2] * f.x + M[5] * f.y) + M[8] * f.z));
    float4* q = reinterpret_cast<float4*>(w.rays + gid);
    q[0] = make_float4(origin.x, origin.y, origin.z, 0.001f);
    q[1] = make_float4(dir.x, dir.y, dir.z, 10000.0f);
}

// append one ray per flagged lane to the secondary queue: one atomic per wave
HR_DEV uint32_t queue_append(bool want, uint32_t* count, RayRec* q, f3 o, f3 d, float t_min, float t_max)
{
    const unsigned long long b = __ballot(want);
    if (b == 0ull) return DDGI_NO_SLOT;
    const int first = __ffsll((long long)b) - 1;
    uint32_t  base  = 0u;
    if ((int)(threadIdx.x & 63) == first) base = atomicAdd(count, (uint32_t)__popcll(b));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
    if (!want) return DDGI_NO_SLOT;
    const uint32_t slot = base + lanes_below(b);
    float4* p = reinterpret_cast<float4*>(q + slot);
    p[0] = make_float4(o.x, o.y, o.z, t_min);
    p[1] = make_float4(d.x, d.y, d.z, t_max);
    return slot;
}

// gi_ray_trace.rchit:95-128 / .rmiss:24-27 without the two visibility queries: one thread per primary ray
__global__ __launch_bounds__(64) void k_ddgi_shade(DDGIWaveArgs w)
{
    const DDGITraceArgs& a = w.t;
    const int lane = threadIdx.x & 63;
    const int R = a.d.rays_per_probe;
    const long long gid = (long long)blockIdx.x * 64 + threadIdx.x;
    const int probe = a.probe_begin + (int)(gid / R), ray = (int)(gid % R);
    const bool valid = probe < a.n_probes;
    uint32_t rays = valid ? 1u : 0u;
    bool     want1 = false, want2 = false;
    f3       so = mk3(0.0f, 0.0f, 0.0f), d1 = mk3(0.0f, 0.0f, 1.0f), d2 = mk3(0.0f, 0.0f, 1.0f);
    float    tmax1 = 0.0f;
    f3       P1 = mk3(0.0f, 0.0f, 0.0f), P2 = P1, I = P1;
    uint32_t flags = 0u;   // 1: add P2 when the sky ray is unoccluded, 2: add I
    if (valid)
    {
        const float4 q1  = reinterpret_cast<const float4*>(w.rays + gid)[1];
        const f3     dir = mk3(q1.x, q1.y, q1.z);
        const float4 hq  = w.hits[gid];
        HitRec h;
        h.t = hq.x; h.u = hq.y; h.v = hq.z; h.prim = (int32_t)__float_as_uint(hq.w);
        float hit_distance = 10000.0f;
        if (h.prim < 0) P1 = a.sky.fetch(dir);
        else
        {
            Rng rng = rng_init((uint32_t)ray, (uint32_t)probe, a.num_frames);
            const SurfaceHit s = surface_at(a.sh, h);
            const f3 Wo = neg3(dir);
            const f3 F0 = mix3(mk3(0.04f, 0.04f, 0.04f), s.albedo, s.metallic);
            const f3 c_diffuse = mix3(mul3(s.albedo, sub3(one3(), F0)), mk3(0.0f, 0.0f, 0.0f), s.metallic);
            const float r2x = next_float(rng), r2y = next_float(rng);
            const DirectSplit ds = direct_lighting_split(a.light, Wo, s.N, s.P, F0, c_diffuse, s.roughness, one3(), true, r2x, r2y, a.sky);
            so = ds.origin; d1 = ds.Wi1; d2 = ds.Wi2; tmax1 = ds.t_max1; want1 = ds.ray1; want2 = true;
            P1 = ds.P1; P2 = ds.P2; flags = 1u;
            if (a.infinite_bounces == 1)
            {
                const f3 F   = fresnel_schlick_roughness(max2(dot3(s.N, Wo), 0.0f), F0, s.roughness);
                const f3 kD  = scale3(sub3(one3(), F), 1.0f - s.metallic);
                const f3 irr = sample_irradiance(a.d, s.P, s.N, Wo, a.prev_irr, a.prev_depth);
                I = mul3(mul3(scale3(kD, a.gi_intensity), c_diffuse), irr);
                flags |= 2u;
            }
            hit_distance = 0.001f + h.t;
        }
        a.dirdist[(size_t)probe * R + ray] = make_uint2(pack_h2(dir.x, dir.y), pack_h2(dir.z, hit_distance));
    }
    const uint32_t slot1 = queue_append(want1, w.sec_count, w.sec_rays, so, d1, 0.01f, tmax1);
    const uint32_t slot2 = queue_append(want2, w.sec_count, w.sec_rays, so, d2, 0.01f, 10000.0f);
    if (valid)
    {
        uint4* o = w.part + (size_t)gid * 3;
        o[0] = make_uint4(__float_as_uint(P1.x), __float_as_uint(P1.y), __float_as_uint(P1.z), __float_as_uint(P2.x));
        o[1] = make_uint4(__float_as_uint(P2.y), __float_as_uint(P2.z), __float_as_uint(I.x), __float_as_uint(I.y));
        o[2] = make_uint4(__float_as_uint(I.z), slot1, slot2, flags);
    }
    rays += (want1 ? 1u : 0u) + (want2 ? 1u : 0u);
    for (int o = 32; o > 0; o >>= 1) rays += __shfl_down(rays, o);
    if (lane == 0) a.ray_slots[blockIdx.x] = rays;
}

// lighting.glsl:117-196 with the two visibilities known (shading.h DirectSplit) + the irradiance term of gi_ray_trace.rchit:119-126
__global__ __launch_bounds__(256) void k_ddgi_combine(DDGIWaveArgs w)
{
    const DDGITraceArgs& a = w.t;
    const int R = a.d.rays_per_probe;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int probe = a.probe_begin + (int)(gid / R), ray = (int)(gid % R);
    if (probe >= a.n_probes) return;
    const uint4* p = w.part + (size_t)gid * 3;
    const uint4  p0 = p[0], p1 = p[1], p2 = p[2];
    const f3 P1 = mk3(__uint_as_float(p0.x), __uint_as_float(p0.y), __uint_as_float(p0.z));
    const f3 P2 = mk3(__uint_as_float(p0.w), __uint_as_float(p1.x), __uint_as_float(p1.y));
    const f3 I  = mk3(__uint_as_float(p1.z), __uint_as_float(p1.w), __uint_as_float(p2.x));
    const bool o1 = p2.y != DDGI_NO_SLOT && w.sec_occluded[p2.y] != 0;
    const bool o2 = p2.z != DDGI_NO_SLOT && w.sec_occluded[p2.z] != 0;
    f3 L = o1 ? mk3(0.0f, 0.0f, 0.0f) : P1;
    if ((p2.w & 1u) && !o2) L = add3(L, P2);
    if (p2.w & 2u) L = add3(L, I);
    a.radiance[(size_t)probe * R + ray] = make_uint2(pack_h2(L.x, L.y), pack_h2(L.z, 0.0f));
}
#endif // HR_DEV_PATHS

// ------------------------------------------------------------------------------------------------
struct DDGIUpdateArgs
{
    DDGIU        d;
    const uint2* radiance;
    const uint2* dirdist;
    const uint2*    prev_irr;
    uint2*          out_irr;
    const uint32_t* prev_dep;
    uint32_t*       out_dep;
    int          first_frame;
    int          gy0;   // first probe z-slab of this launch
};

// An interior texel and the border texels that mirror it (gi_border_update.glsl:35-143, the copy table by formula, inverted: which
// border texels read THIS interior texel).  (sx, sy) in 1..S inside the probe's (S + 2)^2 cell whose corner is (cx, cy): an edge texel has
// one copy on the opposite half of the facing border, a corner texel two of those plus the diagonally opposite corner.
template <typename T>
HR_DEV void store_texel_and_borders(T* atlas, int tw, int cx, int cy, int S, int sx, int sy, T v)
{
    auto put = [&](int dx, int dy) { atlas[(size_t)(cy + dy) * tw + cx + dx] = v; };
    put(sx, sy);
    if (sy == 1) put(S - sx + 1, 0);
    if (sy == S) put(S - sx + 1, S + 1);
    if (sx == 1) put(0, S - sy + 1);
    if (sx == S) put(S + 1, S - sy + 1);
    if ((sx == 1 || sx == S) && (sy == 1 || sy == S)) put(sx == S ? 0 : S + 1, sy == S ? 0 : S + 1);
}

// probe_update() + border_update() of one frame in ONE launch (ddgi.cpp:829-939; gi_probe_update.glsl:136-184 in its irradiance and depth
// variants, gi_border_update.glsl:151-175).  One workgroup per probe (gx = x + y*cx, gy = z): the first depth_side^2 threads own the depth
// texels, the threads from the next wave boundary on the irradiance texels — with the default 16 / 8 sides four depth waves and one irradiance
// wave, which take about the same time (the irradiance texel does less per ray but reads two LDS vectors).  The probe's rays are staged through
// LDS once for both (gi_probe_update.glsl:73-84), with what is the same for every texel done at the staging: the fp16 decode, the ray's clamped
// distance and radiance * 0.95.  Every texel accumulates over the rays in the reference's order (r = 0 .. rays_per_probe - 1): the fp32 sums are
// the reference's bit for bit.  A ray a texel does not take (weight below 1e-8) enters as weight 0 — x + y * 0 == x for the finite distances /
// radiances the trace kernel writes — so the loop has ONE select per ray and no branch.  Each thread then writes its texel AND the border texels
// that mirror it: no border launches, no second pass over the atlas.
// Round 6, 16x8x16 probes x 256 rays: irradiance 34.1 + depth 67.3 + borders 2 x 5.0 us in four launches -> see DESIGN.md §5.
// SHARP50: depth_sharpness == 50 (ddgi.h default) as a compile-time fact: the multiplications det_powi performs for n = 50, written out.
template <bool SHARP50>
__global__ void k_ddgi_probe_update(DDGIUpdateArgs a)
{
    constexpr int CACHE = 256;
    __shared__ float4 s_dd[CACHE];    // direction, clamped distance
    __shared__ float4 s_rad[CACHE];   // radiance * 0.95
    const int sd = a.d.depth_probe_side_length, si = a.d.irradiance_probe_side_length;
    const int irr_base = (sd * sd + 63) & ~63;
    const bool depth_wave = (int)threadIdx.x < irr_base;   // wave-uniform
    const int  side = depth_wave ? sd : si;
    const int  k    = depth_wave ? (int)threadIdx.x : (int)threadIdx.x - irr_base;
    const bool live = k < side * side;
    const int gx = blockIdx.x, gy = blockIdx.y + a.gy0;
    const int lx = k % side, ly = k / side;
    const int probe = gx + (a.d.probe_counts[0] * a.d.probe_counts[1]) * gy; // == probe_id(current_coord, ...)
    const int R = a.d.rays_per_probe;
    const float ncx = ((float)lx + 0.5f) * __fdiv_rn(2.0f, (float)side) - 1.0f, ncy = ((float)ly + 0.5f) * __fdiv_rn(2.0f, (float)side) - 1.0f;
    const f3  texel_dir = gi_oct_decode(ncx, ncy);
    float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f, total_w = 0.0f;
    for (int offset = 0; offset < R; offset += CACHE)
    {
        const int num = (R - offset) < CACHE ? (R - offset) : CACHE;
        __syncthreads();
        for (int i = threadIdx.x; i < num; i += blockDim.x)
        {
            const uint2 q = a.dirdist[(size_t)probe * R + offset + i];
            float dist = min2(a.d.max_distance, h2f_hi(q.y) - 0.01f);
            if (dist == -1.0f) dist = a.d.max_distance;
            s_dd[i] = make_float4(h2f_lo(q.x), h2f_hi(q.x), h2f_lo(q.y), dist);
            const uint2 c = a.radiance[(size_t)probe * R + offset + i];
            s_rad[i] = make_float4(h2f_lo(c.x) * 0.95f, h2f_hi(c.x) * 0.95f, h2f_lo(c.y) * 0.95f, 0.0f);
        }
        __syncthreads();
        if (!live) continue;
        // one ray of the probe (gi_probe_update.glsl:150-181)
        if (depth_wave)
        {
            auto one_ray = [&](int r) {
                const float4 dd = s_dd[r];
                const float  dp = __builtin_fmaxf(0.0f, dot3(texel_dir, mk3(dd.x, dd.y, dd.z)));   // the sum is never NaN: one v_max
                float w;
                if (SHARP50) { const float b2 = dp * dp, b4 = b2 * b2, b8 = b4 * b4, b16 = b8 * b8, b32 = b16 * b16; w = ((1.0f * b2) * b16) * b32; }
                else w = det_pow_auto(dp, a.d.depth_sharpness);
                w = w >= 0.00000001f ? w : 0.0f;
                r0 += dd.w * w; r1 += (dd.w * dd.w) * w; total_w += w;
            };
            int r = 0;
            for (; r + 4 <= num; r += 4) { one_ray(r); one_ray(r + 1); one_ray(r + 2); one_ray(r + 3); }   // four rays' LDS reads in flight together
            for (; r < num; r++) one_ray(r);
        }
        else
        {
            auto one_ray = [&](int r) {
                const float4 dd = s_dd[r], c = s_rad[r];
                float dp = __builtin_fmaxf(0.0f, dot3(texel_dir, mk3(dd.x, dd.y, dd.z)));
                dp = dp >= 0.00000001f ? dp : 0.0f;
                r0 += c.x * dp; r1 += c.y * dp; r2 += c.z * dp; total_w += dp;
            };
            int r = 0;
            for (; r + 4 <= num; r += 4) { one_ray(r); one_ray(r + 1); one_ray(r + 2); one_ray(r + 3); }
            for (; r < num; r++) one_ray(r);
        }
    }
    if (!live) return;
    if (total_w > 0.00000001f) { r0 = __fdiv_rn(r0, total_w); r1 = __fdiv_rn(r1, total_w); r2 = __fdiv_rn(r2, total_w); }
    const int cx = gx * (side + 2) + 1, cy = gy * (side + 2) + 1;   // the corner of the probe's cell, border included
    if (depth_wave)
    {
        const int tw = a.d.depth_texture_width;
        if (a.first_frame == 0)
        {
            const uint32_t pv = a.prev_dep[(size_t)(cy + ly + 1) * tw + cx + lx + 1];
            r0 = mix1(r0, h2f_lo(pv), a.d.hysteresis); r1 = mix1(r1, h2f_hi(pv), a.d.hysteresis);
        }
        store_texel_and_borders<uint32_t>(a.out_dep, tw, cx, cy, side, lx + 1, ly + 1, pack_h2(r0, r1));
    }
    else
    {
        const int tw = a.d.irradiance_texture_width;
        if (a.first_frame == 0)
        {
            const uint2 pv = a.prev_irr[(size_t)(cy + ly + 1) * tw + cx + lx + 1];
            r0 = mix1(r0, h2f_lo(pv.x), a.d.hysteresis); r1 = mix1(r1, h2f_hi(pv.x), a.d.hysteresis); r2 = mix1(r2, h2f_lo(pv.y), a.d.hysteresis);
        }
        store_texel_and_borders<uint2>(a.out_irr, tw, cx, cy, side, lx + 1, ly + 1, make_uint2(pack_h2(r0, r1), pack_h2(r2, 1.0f)));
    }
}


__global__ __launch_bounds__(256) void k_ddgi_sample(DDGISampleArgs a)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = a.y0 + blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= a.w || y >= a.y1) return;
    const size_t o  = (size_t)y * a.w + x;
    const float  dp = a.depth[o];
    if (dp == 1.0f) { a.out[o] = make_uint2(0u, 0u); return; }
    const float tu = __fdiv_rn((float)x + 0.5f, (float)a.w), tv = __fdiv_rn((float)y + 0.5f, (float)a.h);
    const f3    P  = world_pos_from_depth(tu, tv, dp, a.vpi);
    const uint2 g2 = a.gb2[o];
    const f3    N  = oct_decode(h2f_lo(g2.x), h2f_hi(g2.x));
    const f3    Wo = normalize3(sub3(mk3(a.cam[0], a.cam[1], a.cam[2]), P));
    const f3    irr = scale3(sample_irradiance(a.d, P, N, Wo, a.irr, a.dep), a.gi_intensity);
    a.out[o] = make_uint2(pack_h2(irr.x, irr.y), pack_h2(irr.z, 1.0f));
}

// ------------------------------------------------------------------------------------------------
struct hr_ddgi
{
    hr_ctx* ctx = nullptr;
    int     full_w = 0, full_h = 0, w = 0, h = 0, scale = 0;
    DDGIU   d;
    int     n_probes = 0;
    DevBuf  radiance, dirdist, irr[2], dep[2], sample, counters, ray_slots;
    bool    want_stats = false;   // hr_ddgi_trace_stats: launch the instrumented trace kernel
    DevBuf  wf_rays, wf_hits, wf_sec_rays, wf_occluded, wf_part;   // wavefront ray_trace (trace_queue.h); counters + 32: queue words
    bool    wavefront = false;  // developer A/B (HR_DDGI_WAVEFRONT=1): measured slower than the single kernel, DESIGN.md §4
    bool    first_frame = true, ping_pong = false;
    int     z0 = 0, z1 = 0;     // probe z-slabs this instance traces and updates
    int     sy0 = 0, sy1 = 0;   // image rows this instance samples
    StageProfiler prof;
    hipStream_t   last_stream = nullptr;
};

bool hr::profiling_enabled(const hr_ddgi* p) { return p && p->prof.enabled; }

extern "C" {

void hr_ddgi_default_params(hr_ddgi_params* p)
{
    p->infinite_bounces = 1; p->infinite_bounce_intensity = 1.7f; p->gi_intensity = 1.0f;
    for (int i = 0; i < 9; i++) p->random_orientation[i] = (i % 4 == 0) ? 1.0f : 0.0f;
    p->exact = 1;
}

// DDGI::initialize_probe_grid (ddgi.cpp:150-169), create_images' atlas sizes (:197-201), update_properties_ubo (:738-763); member defaults ddgi.h:54-56,71-75,92-95
hr_status hr_ddgi_grid_from_extents(const float* lo, const float* hi, float probe_distance, int32_t rays_per_probe, hr_ddgi_uniforms* out)
{
    HR_CHECK_ARG(lo && hi && out && probe_distance > 0.0f && rays_per_probe > 0);
    hr_ddgi_uniforms u = {};
    for (int k = 0; k < 3; k++)
    {
        HR_CHECK_ARG(hi[k] >= lo[k] && hi[k] - lo[k] < 1e30f);
        const float q = (hi[k] - lo[k]) / probe_distance;   // glm: vec3 / float, then ivec3() truncates
        HR_CHECK_ARG(q < 65536.0f);
        u.probe_counts[k]        = (int32_t)q + 2;           // "Add 2 more probes to fully cover scene."
        u.grid_start_position[k] = lo[k];
        u.grid_step[k]           = probe_distance;
    }
    u.max_distance = probe_distance * 1.5f;
    u.depth_sharpness = 50.0f; u.hysteresis = 0.98f; u.normal_bias = 0.25f; u.energy_preservation = 0.85f;
    u.irradiance_probe_side_length = 8; u.depth_probe_side_length = 16;
    const long long cols = (long long)u.probe_counts[0] * u.probe_counts[1];
    HR_CHECK_ARG((u.depth_probe_side_length + 2) * cols + 2 < (1ll << 30));
    u.irradiance_texture_width  = (u.irradiance_probe_side_length + 2) * (int32_t)cols + 2;
    u.irradiance_texture_height = (u.irradiance_probe_side_length + 2) * u.probe_counts[2] + 2;
    u.depth_texture_width       = (u.depth_probe_side_length + 2) * (int32_t)cols + 2;
    u.depth_texture_height      = (u.depth_probe_side_length + 2) * u.probe_counts[2] + 2;
    u.rays_per_probe = rays_per_probe; u.visibility_test = 1;
    *out = u;
    return HR_OK;
}

hr_status hr_ddgi_create(hr_ctx* ctx, int32_t full_width, int32_t full_height, hr_scale scale, const hr_ddgi_uniforms* grid, hr_ddgi** out)
{
    HR_CHECK_ARG(ctx && out && grid && full_width > 0 && full_height > 0 && (int)scale >= 0 && (int)scale <= 2);
    const DDGIU& g = *grid;
    HR_CHECK_ARG(g.probe_counts[0] > 0 && g.probe_counts[1] > 0 && g.probe_counts[2] > 0 && g.rays_per_probe > 0);
    HR_CHECK_ARG(g.probe_counts[0] <= 1024 && g.probe_counts[1] <= 1024 && g.probe_counts[2] <= 1024);   // a grid cell travels as 3 x 10 bits (shading.h sample_irradiance_net_coop)
    HR_CHECK_ARG(g.irradiance_probe_side_length >= 2 && g.irradiance_probe_side_length <= 16 && g.depth_probe_side_length >= 2 && g.depth_probe_side_length <= 16);
    // atlas sizing of ddgi.cpp:197-201
    HR_CHECK_ARG(g.irradiance_texture_width == (g.irradiance_probe_side_length + 2) * g.probe_counts[0] * g.probe_counts[1] + 2);
    HR_CHECK_ARG(g.irradiance_texture_height == (g.irradiance_probe_side_length + 2) * g.probe_counts[2] + 2);
    HR_CHECK_ARG(g.depth_texture_width == (g.depth_probe_side_length + 2) * g.probe_counts[0] * g.probe_counts[1] + 2);
    HR_CHECK_ARG(g.depth_texture_height == (g.depth_probe_side_length + 2) * g.probe_counts[2] + 2);
    HR_HIP(hipSetDevice(ctx->device));
    hr_ddgi* p = new hr_ddgi();
    p->ctx = ctx; p->full_w = full_width; p->full_h = full_height; p->scale = (int)scale;
    p->w = full_width >> (int)scale; p->h = full_height >> (int)scale;
    p->d = g;
    p->n_probes = g.probe_counts[0] * g.probe_counts[1] * g.probe_counts[2];
    p->z0 = 0; p->z1 = g.probe_counts[2]; p->sy0 = 0; p->sy1 = p->h;
    hr_status s;
#define A(buf, n) if ((s = p->buf.alloc(n)) != HR_OK) { delete p; return s; }
    const size_t nr = (size_t)p->n_probes * g.rays_per_probe;
    A(radiance, nr * 8) A(dirdist, nr * 8)
    const size_t ib = (size_t)g.irradiance_texture_width * g.irradiance_texture_height * 8, db = (size_t)g.depth_texture_width * g.depth_texture_height * 4;
    A(irr[0], ib) A(irr[1], ib) A(dep[0], db) A(dep[1], db)
    A(sample, (size_t)p->w * p->h * 8)
    A(counters, 64)
    A(ray_slots, ((nr + 255) / 256) * 4 * 4)
#ifdef HR_DEV_PATHS
    if (const char* e = getenv("HR_DDGI_WAVEFRONT")) p->wavefront = atoi(e) != 0;   // developer A/B switch, read once
    if (p->wavefront)
    {
        A(wf_rays, nr * sizeof(RayRec)) A(wf_hits, nr * 16) A(wf_sec_rays, 2 * nr * sizeof(RayRec)) A(wf_occluded, 2 * nr) A(wf_part, nr * 48)
    }
#endif
#undef A
    for (int i = 0; i < 2; i++) { HR_HIP(hipMemset(p->irr[i].p, 0, ib)); HR_HIP(hipMemset(p->dep[i].p, 0, db)); }
    HR_HIP(hipMemset(p->counters.p, 0, 64));
    HR_HIP(hipMemset(p->ray_slots.p, 0, p->ray_slots.bytes));
    *out = p;
    return HR_OK;
}

hr_status hr_ddgi_destroy(hr_ddgi* p)
{
    if (!p) return HR_OK;
    (void)hipSetDevice(p->ctx->device);
    (void)hipDeviceSynchronize();
    delete p;
    return HR_OK;
}
hr_status hr_ddgi_set_shard(hr_ddgi* p, int32_t probe_z0, int32_t probe_z1, int32_t row_y0, int32_t row_y1)
{
    HR_CHECK_ARG(p && probe_z0 >= 0 && probe_z1 > probe_z0 && probe_z1 <= p->d.probe_counts[2]);
    HR_CHECK_ARG(row_y0 >= 0 && row_y1 > row_y0 && row_y1 <= p->h && (row_y0 & 7) == 0);
    p->z0 = probe_z0; p->z1 = probe_z1; p->sy0 = row_y0; p->sy1 = row_y1;
    HR_HIP(hipSetDevice(p->ctx->device));
    HR_HIP(hipMemset(p->ray_slots.p, 0, p->ray_slots.bytes));
    return HR_OK;
}

hr_status hr_ddgi_restart_accumulation(hr_ddgi* p) { HR_CHECK_ARG(p); p->first_frame = true; return HR_OK; }
hr_status hr_ddgi_set_normal_bias(hr_ddgi* p, float v) { HR_CHECK_ARG(p && v == v); p->d.normal_bias = v; return HR_OK; }
hr_status hr_ddgi_set_profiling(hr_ddgi* p, int32_t e) { HR_CHECK_ARG(p); p->prof.enabled = e != 0; return HR_OK; }
hr_status hr_ddgi_get_stage_times(hr_ddgi* p, hr_stage_times* out) { HR_CHECK_ARG(p && out); p->prof.collect(out); return HR_OK; }
hr_status hr_ddgi_get_uniforms(hr_ddgi* p, hr_ddgi_uniforms* out) { HR_CHECK_ARG(p && out); *out = p->d; return HR_OK; }
hr_status hr_ddgi_ray_count(hr_ddgi* p, uint64_t* rays)
{
    HR_CHECK_ARG(p && rays);
    HR_HIP(hipStreamSynchronize(p->last_stream));
    std::vector<uint32_t> slots(p->ray_slots.bytes / 4);
    HR_HIP(hipMemcpy(slots.data(), p->ray_slots.p, slots.size() * 4, hipMemcpyDeviceToHost));
    uint64_t total = 0;
    for (uint32_t v : slots) total += v;
    *rays = total;
    return HR_OK;
}

static void scene_shading(const hr_scene* scene, SceneShading& sh)
{
    scene_shading_from(scene, sh);
}

hr_status hr_ddgi_ray_trace(hr_ddgi* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_environment* env, const hr_ddgi_params* prm, void* stream_)
{
    HR_CHECK_ARG(p && scene && in && env && prm && env->sky && env->sky_size > 0);
    hipStream_t st = (hipStream_t)stream_;
    p->last_stream = st;
    const int rd = p->ping_pong ? 0 : 1; // read_idx = !m_ping_pong
    DDGITraceArgs a;
    a.d = p->d; a.light = in->ubo.light;
    for (int i = 0; i < 9; i++) a.orientation[i] = prm->random_orientation[i];
    a.nodes = (const Node8*)scene->nodes.p; a.tris = (const TriGPU*)scene->tris.p;
    scene_shading(scene, a.sh);
    a.sky = CubeMap { (const uint2*)env->sky, env->sky_size };
    a.prev_irr   = AtlasRGBA { (const uint2*)p->irr[rd].p, p->d.irradiance_texture_width, p->d.irradiance_texture_height };
    a.prev_depth = AtlasRG { (const uint32_t*)p->dep[rd].p, p->d.depth_texture_width, p->d.depth_texture_height };
    a.radiance = (uint2*)p->radiance.p; a.dirdist = (uint2*)p->dirdist.p; a.ray_slots = (uint32_t*)p->ray_slots.p;
    a.num_frames = in->num_frames;
    a.infinite_bounces = (prm->infinite_bounces && !p->first_frame) ? 1 : 0; // ddgi.cpp:790
    a.gi_intensity = prm->infinite_bounce_intensity;
    const int slab = p->d.probe_counts[0] * p->d.probe_counts[1];
    a.probe_begin = p->z0 * slab; a.n_probes = p->z1 * slab;
    a.stats = nullptr;
    const long long n = (long long)(a.n_probes - a.probe_begin) * p->d.rays_per_probe;
    if (p->want_stats)
    {
        // instrumented build of the same kernel (hr_ddgi_trace_stats): counters + 8 .. 32
        HR_HIP(hipMemsetAsync((char*)p->counters.p + 8, 0, 24, st));
        a.stats = (unsigned long long*)((char*)p->counters.p + 8);
        const int tb = 64 * DDGI_TRACE_WAVES;
        hipLaunchKernelGGL(k_ddgi_trace<true>, dim3((unsigned)((n + tb - 1) / tb)), dim3(tb), 0, st, a);
        HR_HIP(hipGetLastError());
        return HR_OK;
    }
    int ev = p->prof.begin("ray_trace", st, (uint64_t)n * 16);
#ifdef HR_DEV_PATHS
    if (p->wavefront)
    {
        // queue words (counters + 32): word 1 = length of the secondary queue (appended by k_ddgi_shade, read by the any-hit queue kernel)
        uint32_t* qw = (uint32_t*)((char*)p->counters.p + 32);
        HR_HIP(hipMemsetAsync(qw, 0, 16, st));
        DDGIWaveArgs w;
        w.t = a;
        w.rays = (RayRec*)p->wf_rays.p; w.hits = (const float4*)p->wf_hits.p; w.sec_rays = (RayRec*)p->wf_sec_rays.p;
        w.sec_count = qw + 1; w.sec_occluded = (const uint8_t*)p->wf_occluded.p; w.part = (uint4*)p->wf_part.p;
        const int persistent = trace_queue_grid(p->ctx->props.multiProcessorCount);
        hipLaunchKernelGGL(k_ddgi_gen, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w);
        TraceQueueArgs q;
        q.nodes = a.nodes; q.tris = a.tris;
        q.rays = w.rays; q.n_rays_dev = nullptr; q.n_rays = (uint32_t)n; q.hits = (float4*)p->wf_hits.p; q.occluded = nullptr;
        hipLaunchKernelGGL(k_trace_queue<false>, dim3(persistent), dim3(64), 0, st, q);
        hipLaunchKernelGGL(k_ddgi_shade, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, w);
        q.rays = w.sec_rays; q.n_rays_dev = qw + 1; q.n_rays = 0; q.hits = nullptr; q.occluded = (uint8_t*)p->wf_occluded.p;
        hipLaunchKernelGGL(k_trace_queue<true>, dim3(persistent), dim3(64), 0, st, q);
        hipLaunchKernelGGL(k_ddgi_combine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w);
    }
    else
#endif
    {
        const int tb = 64 * DDGI_TRACE_WAVES;
        hipLaunchKernelGGL(k_ddgi_trace<false>, dim3((unsigned)((n + tb - 1) / tb)), dim3(tb), 0, st, a);
    }
    p->prof.end(ev, st);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

// Instrumented ray trace (same rays, same results; the radiance / direction images are rewritten with the same values): out3 = rays
// traced (probe rays + light / sky rays of the hit points), BVH node steps, triangle tests — the BVH term of SURVEY §8d's algorithmic bytes.
hr_status hr_ddgi_trace_stats(hr_ddgi* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_environment* env, const hr_ddgi_params* prm, uint64_t* out3, void* stream)
{
    HR_CHECK_ARG(p && out3);
    p->want_stats = true;
    const hr_status s = hr_ddgi_ray_trace(p, scene, in, env, prm, stream);
    p->want_stats = false;
    if (s != HR_OK) return s;
    HR_HIP(hipStreamSynchronize((hipStream_t)stream));
    uint64_t host[3];
    HR_HIP(hipMemcpy(host, (char*)p->counters.p + 8, 24, hipMemcpyDeviceToHost));
    out3[0] = host[2]; out3[1] = host[0]; out3[2] = host[1];
    return HR_OK;
}

hr_status hr_ddgi_probe_update(hr_ddgi* p, void* stream_)
{
    HR_SCOPED_SAMPLE("Probe Update");   // ddgi.cpp:831
    HR_CHECK_ARG(p);
    hipStream_t st = (hipStream_t)stream_;
    p->last_stream = st;
    const int rd = p->ping_pong ? 0 : 1, wr = p->ping_pong ? 1 : 0;
    const dim3 grid(p->d.probe_counts[0] * p->d.probe_counts[1], p->z1 - p->z0);
    const uint64_t nr = (uint64_t)grid.x * grid.y * p->d.rays_per_probe;
    DDGIUpdateArgs a;
    a.d = p->d; a.radiance = (const uint2*)p->radiance.p; a.dirdist = (const uint2*)p->dirdist.p; a.first_frame = p->first_frame ? 1 : 0; a.gy0 = p->z0;
    a.prev_irr = (const uint2*)p->irr[rd].p; a.out_irr = (uint2*)p->irr[wr].p;
    a.prev_dep = (const uint32_t*)p->dep[rd].p; a.out_dep = (uint32_t*)p->dep[wr].p;
    // algorithmic bytes: both ray images once, each atlas read (history) and written (interior + borders) once
    const int ev = p->prof.begin("probe_update", st, nr * 16 + 2 * (p->irr[0].bytes + p->dep[0].bytes));
    const int sd2 = p->d.depth_probe_side_length * p->d.depth_probe_side_length, si2 = p->d.irradiance_probe_side_length * p->d.irradiance_probe_side_length;
    const dim3 block(cdiv(sd2, 64) * 64 + cdiv(si2, 64) * 64);   // <= 512: hr_ddgi_create bounds the sides by 16
    if (p->d.depth_sharpness == 50.0f) hipLaunchKernelGGL(k_ddgi_probe_update<true>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(k_ddgi_probe_update<false>, grid, block, 0, st, a);
    p->prof.end(ev, st);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

hr_status hr_ddgi_sample_probe_grid(hr_ddgi* p, const hr_frame_inputs* in, const hr_ddgi_params* prm, void* stream_)
{
    HR_CHECK_ARG(p && in && prm && in->cur.depth && in->cur.gb2 && in->cur.width == p->w && in->cur.height == p->h);
    hipStream_t st = (hipStream_t)stream_;
    p->last_stream = st;
    const int wr = p->ping_pong ? 1 : 0; // read_ds[m_ping_pong] (ddgi.cpp:971): the atlases just written
    DDGISampleArgs a;
    a.d = p->d;
    for (int i = 0; i < 16; i++) a.vpi[i] = in->ubo.view_proj_inverse[i];
    for (int i = 0; i < 3; i++) a.cam[i] = in->ubo.cam_pos[i];
    a.depth = in->cur.depth; a.gb2 = (const uint2*)in->cur.gb2;
    a.irr = AtlasRGBA { (const uint2*)p->irr[wr].p, p->d.irradiance_texture_width, p->d.irradiance_texture_height };
    a.dep = AtlasRG { (const uint32_t*)p->dep[wr].p, p->d.depth_texture_width, p->d.depth_texture_height };
    a.out = (uint2*)p->sample.p; a.w = p->w; a.h = p->h; a.y0 = p->sy0; a.y1 = p->sy1; a.gi_intensity = prm->gi_intensity;
    int ev = p->prof.begin("sample_probe_grid", st, (uint64_t)p->w * (p->sy1 - p->sy0) * 20);
    if (prm->exact) hipLaunchKernelGGL(k_ddgi_sample, dim3(cdiv(p->w, 32), cdiv(p->sy1 - p->sy0, 8)), dim3(256), 0, st, a);
    else launch_ddgi_sample_fast(a, st);
    p->prof.end(ev, st);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

hr_status hr_ddgi_end_frame(hr_ddgi* p)
{
    HR_CHECK_ARG(p);
    p->first_frame = false;
    p->ping_pong   = !p->ping_pong;
    return HR_OK;
}

hr_status hr_ddgi_render(hr_ddgi* p, const hr_scene* scene, const hr_frame_inputs* in, const hr_environment* env, const hr_ddgi_params* prm, void* stream)
{
    HR_SCOPED_SAMPLE("DDGI");
    HR_CHECK_ARG(p && scene && in && env && prm);
    HR_HIP(hipSetDevice(p->ctx->device));
    p->prof.begin_frame();
    hr_status s;
    if ((s = hr_ddgi_ray_trace(p, scene, in, env, prm, stream)) != HR_OK) return s;
    if ((s = hr_ddgi_probe_update(p, stream)) != HR_OK) return s;
    if ((s = hr_ddgi_sample_probe_grid(p, in, prm, stream)) != HR_OK) return s;
    return hr_ddgi_end_frame(p);
}

static void fill_view(hr_image_view* v, void* data, int w, int h, int bpp, hr_format f)
{
    v->data = data; v->width = w; v->height = h; v->row_pitch_bytes = w * bpp; v->format = f;
}

hr_status hr_ddgi_image(hr_ddgi* p, int32_t which, hr_image_view* v)
{
    HR_CHECK_ARG(p && v);
    const DDGIU& d = p->d;
    switch (which)
    {
        case 0: fill_view(v, p->radiance.p, d.rays_per_probe, p->n_probes, 8, HR_FORMAT_RGBA16F); break;
        case 1: fill_view(v, p->dirdist.p, d.rays_per_probe, p->n_probes, 8, HR_FORMAT_RGBA16F); break;
        case 2: case 3: fill_view(v, p->irr[which - 2].p, d.irradiance_texture_width, d.irradiance_texture_height, 8, HR_FORMAT_RGBA16F); break;
        case 4: case 5: fill_view(v, p->dep[which - 4].p, d.depth_texture_width, d.depth_texture_height, 4, HR_FORMAT_RG16F); break;
        case 6: fill_view(v, p->sample.p, p->w, p->h, 8, HR_FORMAT_RGBA16F); break;
        default: set_last_error("hr_ddgi_image: unknown image index"); return HR_ERR_INVALID_ARG;
    }
    return HR_OK;
}

hr_status hr_ddgi_output(hr_ddgi* p, hr_image_view* v) { return hr_ddgi_image(p, 6, v); }

hr_status hr_ddgi_current_write(hr_ddgi* p, hr_image_view* irradiance, hr_image_view* depth)
{
    HR_CHECK_ARG(p && irradiance && depth);
    const int i = p->ping_pong ? 1 : 0; // write_ds[m_ping_pong]: what probe_update fills and sample_probe_grid reads
    hr_status s = hr_ddgi_image(p, 2 + i, irradiance);
    if (s != HR_OK) return s;
    return hr_ddgi_image(p, 4 + i, depth);
}

hr_status hr_ddgi_current_read(hr_ddgi* p, hr_image_view* irradiance, hr_image_view* depth)
{
    HR_CHECK_ARG(p && irradiance && depth);
    const int i = p->ping_pong ? 0 : 1; // read_ds[!m_ping_pong]
    hr_status s = hr_ddgi_image(p, 2 + i, irradiance);
    if (s != HR_OK) return s;
    return hr_ddgi_image(p, 4 + i, depth);
}

} // extern "C"
