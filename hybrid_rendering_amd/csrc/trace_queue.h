// Wavefront ray queues for the DDGI hit-shading pass: the traversal of a ray batch as its own kernel — a developer A/B path
// (HR_DDGI_WAVEFRONT=1), NOT the default: measured slower than the single kernel (docs/EXPERIMENTS.md §4.3).
//
// The reference traces these rays from a ray-generation shader whose closest-hit shader traces further rays (gi_ray_trace.rgen:96 ->
// gi_ray_trace.rchit:95-128 -> ray_query.glsl): on RT cores the driver's scheduler regroups that work.  One HIP kernel doing the same
// per lane runs its loops at 25-50 % (node steps) and 5-7 % (triangle tests) lane utilisation (tools/divergence.py); this path splits
// it the way a wavefront path tracer does:
//     generate rays -> k_trace_queue<closest> -> shade densely, append secondary rays -> k_trace_queue<any-hit> -> combine.
// k_trace_queue is a persistent kernel: a lane whose ray is finished fetches the next ray of its wave's share of the queue, so a
// wave issues node steps for (sum of its rays' steps) / 64 rather than for its longest ray, and the triangle tests are the
// cooperative, redistributed ones of traverse.h (trace_coop).  Results are keyed by ray id: nothing depends on the order in which
// rays are fetched or queues are filled, and every hit decision is the same watertight test on the same operands — images are
// identical to the single-kernel path (tests/test_gpu_ddgi.py::test_ddgi_wavefront_variant compares both with the oracle).
#pragma once
#include "traverse.h"

namespace hr {

struct alignas(16) RayRec
{
    float ox, oy, oz, t_min;
    float dx, dy, dz, t_max;
};

struct TraceQueueArgs
{
    const Node8*    nodes;
    const TriGPU*   tris;
    const RayRec*   rays;
    const uint32_t* n_rays_dev;   // ray count in device memory (a queue filled by an earlier kernel), or nullptr: n_rays
    uint32_t        n_rays;
    float4*         hits;          // closest: (t, u, v, prim as bits; prim = -1: miss) per ray
    uint8_t*        occluded;      // any-hit: 1 / 0 per ray
};

#ifndef HR_QUEUE_REFILL
#define HR_QUEUE_REFILL 16   // refill the wave when this many lanes have no ray
#endif
#ifndef HR_QUEUE_DRAIN
#define HR_QUEUE_DRAIN 16    // flush a partly filled job ring when this many finished lanes wait for their last triangle tests
#endif

template <bool ANY>
__global__ __launch_bounds__(64) void k_trace_queue(TraceQueueArgs a)
{
    __shared__ uint32_t s_stack[HR_STACK_ENTRIES * 64];
    __shared__ CoopWave cw;
    const int      lane   = threadIdx.x;
    const uint32_t n_rays = a.n_rays_dev ? *a.n_rays_dev : a.n_rays;
    uint32_t  spill_array[HR_SPILL_ENTRIES];
    LaneStack st;
    st.init(s_stack, lane, spill_array);
    RayPre   r = ray_prepare(mk3(0.0f, 0.0f, 0.0f), mk3(0.0f, 0.0f, 1.0f));
    float    t_min = 0.0f, t_max = 0.0f, tfar = 0.0f;
    uint32_t ray_id = 0u, cur = 0u, pend = 0u, pend_base = 0u;
    bool     has_ray = false, alive = false, exhausted = false;
    uint32_t head = 0u, count = 0u;   // wave-uniform; head counts every job ever consumed (ring index = head & (RING - 1))
    uint32_t my_end = 0u;             // one past this lane's last job (absolute): the ray retires once head has passed it
    // static partition: this wave owns rays [next, end) — a shared fetch counter serialises on one L2 address (measured: ~11 ns per
    // atomic, 26 k refills = the whole kernel time)
    const uint32_t rpw  = (n_rays + gridDim.x - 1u) / gridDim.x;
    uint32_t       next = blockIdx.x * rpw;
    const uint32_t end  = next + rpw < n_rays ? next + rpw : n_rays;
    exhausted = next >= end;
    for (;;)
    {
        // retire: traversal over and every triangle job of the ray consumed
        if (has_ray && !alive && pend == 0u && (int32_t)(head - my_end) >= 0)
        {
            const unsigned long long k = cw.key[lane];
            if (ANY) a.occluded[ray_id] = k == 0ull ? 1 : 0;
            else
            {
                float4 h = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0xffffffffu));
                if (k != ~0ull)
                {
                    const float2 uv = cw.uv[lane];
                    h = make_float4(ordered_float((uint32_t)(k >> 32)), uv.x, uv.y, __uint_as_float((uint32_t)k));
                }
                a.hits[ray_id] = h;
            }
            has_ray = false;
        }
        const unsigned long long need = __ballot(!has_ray);
        const uint32_t           n_need = (uint32_t)__popcll(need);
        if (!exhausted && (n_need >= HR_QUEUE_REFILL || n_need == 64u))
        {
            const uint32_t base = next;
            next += n_need;
            if (!has_ray)
            {
                const uint32_t id = base + lanes_below(need);
                if (id < end)
                {
                    const float4* q = reinterpret_cast<const float4*>(a.rays + id);
                    const float4  q0 = q[0], q1 = q[1];
                    r = ray_prepare(mk3(q0.x, q0.y, q0.z), mk3(q1.x, q1.y, q1.z));
                    t_min = q0.w; t_max = q1.w; tfar = q1.w;
                    ray_id = id; cur = 1u; pend = 0u; st.sp = 0;
                    has_ray = true; alive = true; my_end = head;
                    cw.key[lane] = ~0ull;
                }
            }
            exhausted = next >= end;
        }
        if (__ballot(has_ray) == 0ull)
        {
            if (exhausted) break;
            continue;   // (cannot happen: an empty wave always refills)
        }
        if (alive && pend == 0u)
        {
            uint32_t ni;
            if (walk_next<(ANY ? HR_ANY_ORDER : HR_ORDER_NEAR) != HR_ORDER_SLOTS>(cur, st, ni))
            {
                const NodeHits h = test_node<ANY ? HR_ANY_ORDER : HR_ORDER_NEAR>(load_node(a.nodes, ni), r, t_min, tfar);
                pend      = walk_expand(h, cur, st);
                pend_base = h.tri_base;
            }
            else
                alive = false;
        }
        const uint32_t left = (uint32_t)__popc(pend);
        uint32_t       pos  = head + count;
#pragma unroll
        for (int k = 0; k < HR_COOP_PUSH; k++)
        {
            const bool               has = left > (uint32_t)k;
            const unsigned long long b   = __ballot(has);
            if (has)
            {
                const uint32_t i = (uint32_t)__builtin_ctz(pend);
                pend &= pend - 1u;
                const uint32_t at = pos + lanes_below(b);
                cw.jobs[at & (HR_COOP_RING - 1)] = ((uint32_t)lane << 26) | (pend_base + i);
                my_end = at + 1u;
            }
            pos += (uint32_t)__popcll(b);
        }
        count = pos - head;
        const bool     walking = __ballot(alive) != 0ull;
        const uint32_t waiting = (uint32_t)__popcll(__ballot(has_ray && !alive && pend == 0u && (int32_t)(head - my_end) < 0));
        bool           flushed = false;
        if (count >= 64u || (count > 0u && (!walking || waiting >= HR_QUEUE_DRAIN)))
        {
            do
            {
                const uint32_t n = count < 64u ? count : 64u;
                wave_fence();
                coop_flush<ANY>(cw, a.tris, r, t_min, t_max, (uint32_t)lane, head, n, lane);
                head += n; count -= n;
            } while (count >= 64u);
            flushed = true;
        }
        if (flushed && has_ray)
        {
            const unsigned long long k = cw.key[lane];
            if (ANY) { if (k == 0ull) { alive = false; pend = 0u; } }
            else if (k != ~0ull) tfar = ordered_float((uint32_t)(k >> 32)) * 1.0000005f;
        }
    }
}

// persistent grid: every SIMD of the device filled to the occupancy the kernel's registers and LDS allow
inline int trace_queue_grid(int n_cus) { return n_cus * 4 * 6; }

} // namespace hr
