// Software ray traversal of the compressed 8-wide BVH on CDNA4 — replaces rayQueryEXT /
// traceRayEXT (reference call sites: ray_query.glsl:13-27,42-56; reflections_ray_trace.rgen:150,165;
// gi_ray_trace.rgen:96).  Semantics kept from the reference's usage: all geometry opaque, no face
// culling, candidate iff t_min < t < t_max, any-hit = gl_RayFlagsTerminateOnFirstHitEXT.
//
// One ray per lane.  Per-lane traversal stack of internal-node indices lives in LDS, laid out
// [entry][lane] so a wave's pushes/pops hit 64 distinct banks.  Leaf children are not pushed: the
// hit leaves of a node are merged into a 32-bit triangle mask (their triangles are contiguous from
// tri_base) and tested right away.
//
// Box tests use FMAs and a conservative far-plane scale — they only have to be conservative.
// The triangle test is the watertight test of Woop/Benthin/Wald (JCGT 2013) in individually
// rounded fp32 ops (no FMA), so that hit decisions are reproducible bit for bit on a CPU.
#pragma once
#include "bvh.h"
#include "device_math.h"

namespace hr {

#define HR_STACK_ENTRIES 16   // LDS entries per lane (16 KB per 4-wave block => 8 waves/SIMD); deeper pushes spill to scratch
#define HR_SPILL_ENTRIES 48
static_assert(HR_STACK_ENTRIES + HR_SPILL_ENTRIES >= hr::kMaxTraversalDepth, "traversal stack shallower than the deepest BVH hr_scene_create accepts");

// Developer instrumentation (tools/divergence.py builds with -DHR_TRACE_DIVERGENCE): what a wave executes against what its lanes
// need.  lane_*: loop iterations THIS lane ran; wave_*: counted by one lane of every iteration the wave executed.
struct DivCounters { uint32_t lane_nodes, lane_pairs, wave_nodes, wave_pairs; };
#ifdef HR_TRACE_DIVERGENCE
#define HR_DIV(...) __VA_ARGS__
HR_DEV void div_count(uint32_t& lane_ctr, uint32_t& wave_ctr)
{
    lane_ctr++;
    const unsigned long long b = __ballot(1);
    if ((int)(threadIdx.x & 63) == __ffsll((long long)b) - 1) wave_ctr++;
}
// slots: 0 sum lane_nodes, 1 sum lane_pairs, 2 wave_nodes, 3 wave_pairs, 4 max-lane nodes, 5 max-lane pairs, 6 waves, 7 lanes that traced
HR_DEV void div_flush(const DivCounters& c, unsigned long long* g)
{
    uint32_t ln = c.lane_nodes, lp = c.lane_pairs, wn = c.wave_nodes, wp = c.wave_pairs, mn = c.lane_nodes, mp = c.lane_pairs, act = c.lane_nodes ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1)
    {
        ln += __shfl_xor(ln, o); lp += __shfl_xor(lp, o); wn += __shfl_xor(wn, o); wp += __shfl_xor(wp, o); act += __shfl_xor(act, o);
        const uint32_t a = __shfl_xor(mn, o), b = __shfl_xor(mp, o);
        mn = a > mn ? a : mn; mp = b > mp ? b : mp;
    }
    if ((threadIdx.x & 63) == 0)
    {
        atomicAdd(g + 0, (unsigned long long)ln); atomicAdd(g + 1, (unsigned long long)lp); atomicAdd(g + 2, (unsigned long long)wn); atomicAdd(g + 3, (unsigned long long)wp);
        atomicAdd(g + 4, (unsigned long long)mn); atomicAdd(g + 5, (unsigned long long)mp); atomicAdd(g + 6, 1ull); atomicAdd(g + 7, (unsigned long long)act);
    }
}
#else
#define HR_DIV(...)
#endif

struct RayPre
{
    f3    o;
    int   kx, ky, kz;
    float Sx, Sy, Sz;
    // box-test side
    float idx, idy, idz;     // 1/d with |d| clamped away from 0
    float ox, oy, oz;        // -o * id
    uint32_t sel;            // bit a set => d[a] < 0 (near plane is qhi)
};

HR_DEV float pick(f3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }

HR_DEV RayPre ray_prepare(f3 o, f3 d)
{
    RayPre r;
    r.o = o;
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    int   kz = 0;
    if (ay > ax) kz = 1;
    if (az > (kz == 0 ? ax : ay)) kz = 2;
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    if (pick(d, kz) < 0.0f) { int t = kx; kx = ky; ky = t; }
    r.kx = kx; r.ky = ky; r.kz = kz;
    float dz = pick(d, kz);
    r.Sx = __fdiv_rn(pick(d, kx), dz);
    r.Sy = __fdiv_rn(pick(d, ky), dz);
    r.Sz = __fdiv_rn(1.0f, dz);
    const float tiny = 1e-18f;
    float dx_ = fabsf(d.x) < tiny ? (d.x < 0.0f ? -tiny : tiny) : d.x;
    float dy_ = fabsf(d.y) < tiny ? (d.y < 0.0f ? -tiny : tiny) : d.y;
    float dz_ = fabsf(d.z) < tiny ? (d.z < 0.0f ? -tiny : tiny) : d.z;
    // box-test side: only has to be conservative (padded boxes, scaled far plane), so the 1-ulp hardware reciprocal will do
    r.idx = __builtin_amdgcn_rcpf(dx_); r.idy = __builtin_amdgcn_rcpf(dy_); r.idz = __builtin_amdgcn_rcpf(dz_);
    r.sel = (dx_ < 0.0f ? 1u : 0u) | (dy_ < 0.0f ? 2u : 0u) | (dz_ < 0.0f ? 4u : 0u);
    return r;
}

// Watertight ray/triangle test.  true iff t_min < t < t_max.  Optionally returns t,u,v.
// kx, ky, kz: the ray's axis permutation — r.kx / r.ky / r.kz, or the same values as compile-time constants when a whole wave shares
// them (ray_tri_uniform below): the 18 v_cndmask of the component picks then fold away; the arithmetic is the same either way.
template <bool WANT_TUV>
HR_DEV bool ray_tri_perm(const RayPre& r, const int kx, const int ky, const int kz, f3 v0, f3 v1, f3 v2, float t_min, float t_max, float& t_out, float& u_out, float& v_out)
{
    f3 A = sub3(v0, r.o), B = sub3(v1, r.o), C = sub3(v2, r.o);
    float Akz = pick(A, kz), Bkz = pick(B, kz), Ckz = pick(C, kz);
    float Ax = pick(A, kx) - r.Sx * Akz, Ay = pick(A, ky) - r.Sy * Akz;
    float Bx = pick(B, kx) - r.Sx * Bkz, By = pick(B, ky) - r.Sy * Bkz;
    float Cx = pick(C, kx) - r.Sx * Ckz, Cy = pick(C, ky) - r.Sy * Ckz;
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    float det = (U + V) + W;
    if (det == 0.0f) return false;
    float Az = r.Sz * Akz, Bz = r.Sz * Bkz, Cz = r.Sz * Ckz;
    float T  = (U * Az + V * Bz) + W * Cz;
    float sg = det < 0.0f ? -1.0f : 1.0f;
    float Ts = T * sg, ad = det * sg;
    if (!(Ts > t_min * ad && Ts < t_max * ad)) return false;
    if (WANT_TUV)
    {
        float inv = __fdiv_rn(1.0f, det);
        t_out = T * inv;
        u_out = V * inv;
        v_out = W * inv;
    }
    return true;
}
template <bool WANT_TUV>
HR_DEV bool ray_tri(const RayPre& r, f3 v0, f3 v1, f3 v2, float t_min, float t_max, float& t_out, float& u_out, float& v_out)
{
    return ray_tri_perm<WANT_TUV>(r, r.kx, r.ky, r.kz, v0, v1, v2, t_min, t_max, t_out, u_out, v_out);
}
// permutation code of a ray: kz * 2 + (kx, ky swapped); -1 from wave_perm_code when the lanes of the wave disagree
HR_DEV int perm_code(const RayPre& r) { return r.kz * 2 + ((r.kx != (r.kz == 2 ? 0 : r.kz + 1)) ? 1 : 0); }
HR_DEV int wave_perm_code(const RayPre& r)
{
    const int c = perm_code(r), c0 = __builtin_amdgcn_readfirstlane(c);
    return __all(c == c0) ? c0 : -1;
}

HR_DEV float ubyte(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xffu); }

struct NodeHits
{
    uint32_t child_base, tri_base;
    uint32_t meta_lo, meta_hi;
    uint32_t hit8;        // bit i set => child i's box is hit
    uint32_t n_internal;  // slots 0..n_internal-1 are internal nodes child_base + i
    uint32_t rev;         // 1: the ray runs against the axis the internal children are sorted along -> visit high slots first
};

// Tests the 8 quantised child boxes of a node.  t_far is the current far limit.  (Packed v_pk_fma_f32 for the
// near/far pair was tried: same speed per FMA here, 5-8 more VGPRs, 5-7% slower AO trace — scalar FMAs stay.)
struct NodeRaw { uint4 q0, q1, q2, q3, q4; };   // the 80 bytes of a node, as loaded

HR_DEV NodeRaw load_node(const Node8* __restrict__ nodes, uint32_t ni)
{
    const uint4* p = reinterpret_cast<const uint4*>(nodes + ni);
    NodeRaw n;
    n.q0 = p[0]; n.q1 = p[1]; n.q2 = p[2]; n.q3 = p[3]; n.q4 = p[4];
    return n;
}

// ORDER: in which order the walk visits the hit internal children of a node (they are sorted along one axis of the node,
// bvh.h).  HR_ORDER_NEAR: near to far — closest-hit queries, the far limit shrinks soonest.  HR_ORDER_FAR: far to near — any-hit
// queries: their rays run from a surface towards a light or the sky, and what blocks those is mostly the building's envelope
// (roof slabs, walls, hanging fabric) at the far end, not the geometry around the origin: on the CPU replay of the bench frame
// (tools/bvh_eval.cpp) the wave-level node steps of the hard tier's shadow rays fall by 55 %, the hit shaders' light / sky rays by
// 12-60 %, the standard tier is unchanged.  HR_ORDER_SLOTS: ascending slots whatever the direction.  The answer of a query does
// not depend on the order (any-hit: a function of the geometry; closest hit: smallest t, ties to the smallest index).
#define HR_ORDER_SLOTS 0
#define HR_ORDER_NEAR 1
#define HR_ORDER_FAR 2
#ifndef HR_ANY_ORDER
#define HR_ANY_ORDER HR_ORDER_FAR   // developer A/B: HR_CFLAGS=-DHR_ANY_ORDER=0
#endif
template <int ORDER>
HR_DEV NodeHits test_node(const NodeRaw& n, const RayPre& r, float t_near, float t_far)
{
    const uint4 q0 = n.q0, q1 = n.q1, q2 = n.q2, q3 = n.q3, q4 = n.q4;
    const float  nox = __uint_as_float(q0.x), noy = __uint_as_float(q0.y), noz = __uint_as_float(q0.z);
    const float  sx = __uint_as_float((q0.w & 0xffu) << 23), sy = __uint_as_float(((q0.w >> 8) & 0xffu) << 23), sz = __uint_as_float(((q0.w >> 16) & 0xffu) << 23);
    const float  ax = sx * r.idx, ay = sy * r.idy, az = sz * r.idz;
    const float  bx = (nox - r.o.x) * r.idx, by = (noy - r.o.y) * r.idy, bz = (noz - r.o.z) * r.idz;
    // per axis: words holding the near / far plane bytes for children 0-3 and 4-7
    const bool   nx = r.sel & 1u, ny = r.sel & 2u, nz = r.sel & 4u;
    const uint32_t lox0 = q2.x, lox1 = q2.y, loy0 = q2.z, loy1 = q2.w, loz0 = q3.x, loz1 = q3.y;
    const uint32_t hix0 = q3.z, hix1 = q3.w, hiy0 = q4.x, hiy1 = q4.y, hiz0 = q4.z, hiz1 = q4.w;
    const uint32_t nX[2] = { nx ? hix0 : lox0, nx ? hix1 : lox1 }, fX[2] = { nx ? lox0 : hix0, nx ? lox1 : hix1 };
    const uint32_t nY[2] = { ny ? hiy0 : loy0, ny ? hiy1 : loy1 }, fY[2] = { ny ? loy0 : hiy0, ny ? loy1 : hiy1 };
    const uint32_t nZ[2] = { nz ? hiz0 : loz0, nz ? hiz1 : loz1 }, fZ[2] = { nz ? loz0 : hiz0, nz ? loz1 : hiz1 };
    NodeHits h;
    h.child_base = q1.x; h.tri_base = q1.y; h.meta_lo = q1.z; h.meta_hi = q1.w;
    h.n_internal = (q0.w >> 24) & 15u;
    h.rev = 0u;
    if (ORDER != HR_ORDER_SLOTS)
    {
        // the builder's sort axis of the internal children sits in the low bits of slot 0's meta byte
        const uint32_t along = (r.sel >> (q1.z & 3u)) & 1u;   // the ray runs against the axis
        h.rev = ORDER == HR_ORDER_NEAR ? along : along ^ 1u;
    }
    uint32_t hits = 0;
#pragma unroll
    for (int half = 0; half < 2; half++)
    {
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const float tnx = hr_fma(ubyte(nX[half], k), ax, bx), tfx = hr_fma(ubyte(fX[half], k), ax, bx);
            const float tny = hr_fma(ubyte(nY[half], k), ay, by), tfy = hr_fma(ubyte(fY[half], k), ay, by);
            const float tnz = hr_fma(ubyte(nZ[half], k), az, bz), tfz = hr_fma(ubyte(fZ[half], k), az, bz);
            const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, t_near));
            const float tf = fminf(fminf(tfx, tfy), fminf(tfz, t_far)) * 1.0000005f;
            hits |= (tn <= tf) ? (1u << (half * 4 + k)) : 0u;
        }
    }
    h.hit8 = hits & ((1u << (q0.w >> 28)) - 1u);   // empty slots never hit
    return h;
}

// Per-lane stack: LDS part [HR_STACK_ENTRIES][64] per wave; deeper pushes go to a per-lane private array.
// The private array is a SEPARATE object (only its address is kept here): dynamic indexing pins an object to
// scratch memory, and when `sp` lived in the same struct every push/pop of the hot loop became a scratch
// round trip (found in the ISA: scratch_load/store of sp around each of the 8 child pushes).
struct LaneStack
{
    uint32_t* lds;   // &wave_region[lane]
    uint32_t* spill; // private overflow array, HR_SPILL_ENTRIES entries
    int       sp;
    HR_DEV void init(uint32_t* wave_region, int lane, uint32_t* spill_array) { lds = wave_region + lane; spill = spill_array; sp = 0; }
    HR_DEV bool empty() const { return sp == 0; }
    // The walk keeps one entry per BVH level and hr_scene_create refuses trees deeper than kMaxTraversalDepth (bvh.h; the
    // builder caps the depth, bvh_build.cpp kSahDepth), so `sp` never reaches the capacity.  Belt and braces: a push beyond
    // it is dropped WITHOUT advancing sp — push and pop stay paired and no index ever leaves the arrays.
    HR_DEV void push(uint32_t v)
    {
        if (sp < HR_STACK_ENTRIES) lds[sp * 64] = v;
        else if (sp < HR_STACK_ENTRIES + HR_SPILL_ENTRIES) spill[sp - HR_STACK_ENTRIES] = v;
        else return;
        sp++;
    }
    HR_DEV uint32_t pop()
    {
        sp--;
        if (sp < HR_STACK_ENTRIES) return lds[sp * 64];
        return spill[sp - HR_STACK_ENTRIES];
    }
};

HR_DEV void load_tri(const TriGPU* __restrict__ tris, uint32_t i, f3& v0, f3& v1, f3& v2, uint32_t& prim)
{
    const uint4* p = reinterpret_cast<const uint4*>(tris + i);
    uint4 a = p[0], b = p[1], c = p[2];
    v0 = mk3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z));
    v1 = mk3(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z));
    v2 = mk3(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z));
    prim = a.w;
}

struct TriRaw { uint4 a, b, c; };
HR_DEV TriRaw load_tri_raw(const TriGPU* __restrict__ tris, uint32_t i)
{
    const uint4* p = reinterpret_cast<const uint4*>(tris + i);
    TriRaw t;
    t.a = p[0]; t.b = p[1]; t.c = p[2];
    return t;
}
// `code`: wave_perm_code() of the rays being traced (wave-uniform, so the switch is a scalar branch).  Shadow rays towards a
// directional light and the hit shaders' light rays share one permutation per wave almost always.
template <bool WANT_TUV>
HR_DEV bool ray_tri_raw_uniform(const RayPre& r, int code, const TriRaw& q, float t_min, float t_max, float& t, float& u, float& v)
{
    const f3 v0 = mk3(__uint_as_float(q.a.x), __uint_as_float(q.a.y), __uint_as_float(q.a.z)), v1 = mk3(__uint_as_float(q.b.x), __uint_as_float(q.b.y), __uint_as_float(q.b.z)),
             v2 = mk3(__uint_as_float(q.c.x), __uint_as_float(q.c.y), __uint_as_float(q.c.z));
    switch (code)
    {
    case 0: return ray_tri_perm<WANT_TUV>(r, 1, 2, 0, v0, v1, v2, t_min, t_max, t, u, v);
    case 1: return ray_tri_perm<WANT_TUV>(r, 2, 1, 0, v0, v1, v2, t_min, t_max, t, u, v);
    case 2: return ray_tri_perm<WANT_TUV>(r, 2, 0, 1, v0, v1, v2, t_min, t_max, t, u, v);
    case 3: return ray_tri_perm<WANT_TUV>(r, 0, 2, 1, v0, v1, v2, t_min, t_max, t, u, v);
    case 4: return ray_tri_perm<WANT_TUV>(r, 0, 1, 2, v0, v1, v2, t_min, t_max, t, u, v);
    case 5: return ray_tri_perm<WANT_TUV>(r, 1, 0, 2, v0, v1, v2, t_min, t_max, t, u, v);
    default: return ray_tri_perm<WANT_TUV>(r, r.kx, r.ky, r.kz, v0, v1, v2, t_min, t_max, t, u, v);
    }
}
template <bool WANT_TUV>
HR_DEV bool ray_tri_raw(const RayPre& r, const TriRaw& q, float t_min, float t_max, float& t, float& u, float& v)
{
    return ray_tri<WANT_TUV>(r, mk3(__uint_as_float(q.a.x), __uint_as_float(q.a.y), __uint_as_float(q.a.z)),
                             mk3(__uint_as_float(q.b.x), __uint_as_float(q.b.y), __uint_as_float(q.b.z)),
                             mk3(__uint_as_float(q.c.x), __uint_as_float(q.c.y), __uint_as_float(q.c.z)), t_min, t_max, t, u, v);
}

// Depth-first walk with one stack entry per NODE: entry = child_base << 9 | rev << 8 | mask of its hit internal children
// that are still to be visited (rev: take the highest slot first — the children are sorted along the node's longest axis
// and the ray runs against it).  The entry being consumed stays in a register (`cur`); it goes to the stack only when a
// newly tested node has internal hits of its own while `cur` still has siblings left.
// ORDERED: the entry's `rev` bit picks the end of the mask to start from (test_node's ORDER).
template <bool ORDERED>
HR_DEV bool walk_next(uint32_t& cur, LaneStack& st, uint32_t& ni)
{
    if ((cur & 0xffu) == 0u)
    {
        if (st.empty()) return false;
        cur = st.pop();
    }
    const uint32_t m = cur & 0xffu;
    const uint32_t i = (ORDERED && (cur & 0x100u)) ? 31u - (uint32_t)__builtin_clz(m) : (uint32_t)__builtin_ctz(m);
    cur &= ~(1u << i);
    ni = (cur >> 9) + i;
    return true;
}

// after test_node: schedule the hit internal children, return the triangle mask of the hit leaves
HR_DEV uint32_t walk_expand(const NodeHits& h, uint32_t& cur, LaneStack& st)
{
    const uint32_t imask = (1u << h.n_internal) - 1u;
    const uint32_t ih    = h.hit8 & imask;
    if (ih)
    {
        if (cur & 0xffu) st.push(cur);
        cur = (h.child_base << 9) | (h.rev << 8) | ih;
    }
    uint32_t lh = h.hit8 & ~imask, trimask = 0;
    while (lh)
    {
        const uint32_t i = (uint32_t)__builtin_ctz(lh);
        lh &= lh - 1u;
        const uint32_t m = ((i < 4 ? h.meta_lo : h.meta_hi) >> (8 * (i & 3))) & 0xffu;
        trimask |= ((1u << (m >> 5)) - 1u) << (m & 31u);
    }
    return trimask;
}

// Any-hit query (query_distance / query_visibility).  Returns true if occluded.
// Deepest node whose subtree holds every triangle that can touch the box [lo, hi]: walk down from the root while exactly
// one child box overlaps the query box and that child is internal.  Rays that live inside the box (short AO rays around a
// pixel: all spp share it) can start their traversal there instead of at the root — the hit set is unchanged, because a
// child whose (conservative) box misses the query box cannot hold a triangle the rays reach.  HR_NO_ENTRY: nothing overlaps.
#define HR_NO_ENTRY 0xffffffffu
HR_DEV uint32_t entry_node_for_box(const Node8* __restrict__ nodes, f3 lo, f3 hi)
{
    uint32_t ni = 0;
    for (int depth = 0; depth < 24; depth++)
    {
        const NodeRaw n = load_node(nodes, ni);
        const float nox = __uint_as_float(n.q0.x), noy = __uint_as_float(n.q0.y), noz = __uint_as_float(n.q0.z);
        const float sx = __uint_as_float((n.q0.w & 0xffu) << 23), sy = __uint_as_float(((n.q0.w >> 8) & 0xffu) << 23), sz = __uint_as_float(((n.q0.w >> 16) & 0xffu) << 23);
        const uint32_t wlo[6] = { n.q2.x, n.q2.y, n.q2.z, n.q2.w, n.q3.x, n.q3.y };   // lo x0 x1 y0 y1 z0 z1
        const uint32_t whi[6] = { n.q3.z, n.q3.w, n.q4.x, n.q4.y, n.q4.z, n.q4.w };
        uint32_t ov = 0;
#pragma unroll
        for (int half = 0; half < 2; half++)
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const bool o = hr_fma(ubyte(wlo[0 + half], k), sx, nox) <= hi.x && hr_fma(ubyte(whi[0 + half], k), sx, nox) >= lo.x &&
                               hr_fma(ubyte(wlo[2 + half], k), sy, noy) <= hi.y && hr_fma(ubyte(whi[2 + half], k), sy, noy) >= lo.y &&
                               hr_fma(ubyte(wlo[4 + half], k), sz, noz) <= hi.z && hr_fma(ubyte(whi[4 + half], k), sz, noz) >= lo.z;
                ov |= o ? (1u << (half * 4 + k)) : 0u;
            }
        ov &= (1u << (n.q0.w >> 28)) - 1u;
        if (ov == 0u) return HR_NO_ENTRY;
        const uint32_t imask = (1u << ((n.q0.w >> 24) & 15u)) - 1u;
        if ((ov & (ov - 1u)) != 0u || (ov & ~imask) != 0u) break;   // several children, or a leaf: start here
        ni = n.q1.x + (uint32_t)__builtin_ctz(ov);
    }
    return ni;
}

// hit_tri (optional): index into `tris` of the triangle that occluded the ray (untouched on a miss) — the next frame's first guess
// (shadow trace: occluder cache).
template <bool STATS, int ORDER = HR_ANY_ORDER>
HR_DEV bool trace_any(const Node8* __restrict__ nodes, const TriGPU* __restrict__ tris, f3 o, f3 d, float t_min, float t_max,
                      uint32_t* wave_stack, int lane, uint32_t& n_nodes, uint32_t& n_tris, uint32_t entry = 0u, DivCounters* dv = nullptr,
                      uint32_t* hit_tri = nullptr)
{
    if (entry == HR_NO_ENTRY) return false;
    RayPre    r = ray_prepare(o, d);
    uint32_t  spill_array[HR_SPILL_ENTRIES];
    LaneStack st;
    st.init(wave_stack, lane, spill_array);
    uint32_t cur = (entry << 9) | 1u, ni;   // the entry node (root = 0) = "child 0 of child_base entry"
    bool     hit = false;
    const int pcode = wave_perm_code(r);
    while (walk_next<ORDER != HR_ORDER_SLOTS>(cur, st, ni))
    {
        const NodeHits h = test_node<ORDER>(load_node(nodes, ni), r, t_min, t_max);
        if (STATS) n_nodes++;
        HR_DIV(if (dv) div_count(dv->lane_nodes, dv->wave_nodes);)
        uint32_t trimask = walk_expand(h, cur, st);
        while (trimask)
        {
            HR_DIV(if (dv) div_count(dv->lane_pairs, dv->wave_pairs);)
            // two triangles per iteration: both fetches in flight together, both tests back to back (AO trace -2.5%)
            const uint32_t i0 = (uint32_t)__builtin_ctz(trimask);
            trimask &= trimask - 1u;
            const bool     two = trimask != 0u;
            const uint32_t i1  = two ? (uint32_t)__builtin_ctz(trimask) : i0;
            trimask &= trimask - 1u;   // no-op on 0
            const TriRaw ta = load_tri_raw(tris, h.tri_base + i0), tb = load_tri_raw(tris, h.tri_base + i1);
            if (STATS) n_tris += two ? 2u : 1u;
            float t, u, v;
            const bool ha = ray_tri_raw_uniform<false>(r, pcode, ta, t_min, t_max, t, u, v);
            const bool hb = ray_tri_raw_uniform<false>(r, pcode, tb, t_min, t_max, t, u, v);
            if (ha || hb) { hit = true; if (hit_tri) *hit_tri = h.tri_base + (ha ? i0 : i1); break; }
        }
        if (hit) break;
    }
    return hit;
}


HR_DEV void     wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
HR_DEV uint32_t lanes_below(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

#ifdef HR_DEV_PATHS   // A/B paths that lost (docs/EXPERIMENTS.md 4.3): -DAO_SEQ / -DDDGI_SEQ need -DHR_DEV_PATHS
// ---- lane-sequential any-hit rays ---------------------------------------------------------------------------------------------
// NB rays per lane (the sample rays of one AO pixel), walked back to back INSIDE one wave-level loop: a lane whose ray is done
// (occluded, or its stack ran empty) switches to its next ray at once instead of idling until the slowest lane of the wave has
// finished the current sample.  A wave then waits once for max_lanes(sum of its rays' steps) instead of NB times for
// sum_rays(max_lanes(steps)) — the ray LENGTHS are what idles lanes (docs/EXPERIMENTS.md §4.3), and a sum of NB lengths spreads less than
// NB maxima.  No queue, no atomics, no cross-lane traffic: the rays of a lane are prepared up front and live in registers; a switch
// is a handful of v_cndmask.  Decisions per ray are those of trace_any (same node / triangle tests, any-hit), so the masks are
// bit-identical.  Returns the bit mask of OCCLUDED rays.
struct RaySeq { float Sx, Sy, Sz, idx, idy, idz, t_max; uint32_t code; };   // code: kx | ky << 2 | kz << 4 | sel << 6

HR_DEV RaySeq rayseq_pack(const RayPre& r, float t_max)
{
    RaySeq q;
    q.t_max = t_max;
    q.Sx = r.Sx; q.Sy = r.Sy; q.Sz = r.Sz; q.idx = r.idx; q.idy = r.idy; q.idz = r.idz;
    q.code = (uint32_t)r.kx | ((uint32_t)r.ky << 2) | ((uint32_t)r.kz << 4) | (r.sel << 6);
    return q;
}
HR_DEV void rayseq_unpack(RayPre& r, const RaySeq& q)
{
    r.Sx = q.Sx; r.Sy = q.Sy; r.Sz = q.Sz; r.idx = q.idx; r.idy = q.idy; r.idz = q.idz;
    r.kx = (int)(q.code & 3u); r.ky = (int)((q.code >> 2) & 3u); r.kz = (int)((q.code >> 4) & 3u); r.sel = q.code >> 6;
}
template <int NB>
HR_DEV RaySeq rayseq_select(const RaySeq (&q)[NB], int s)
{
    RaySeq o = q[0];
#pragma unroll
    for (int k = 1; k < NB; k++)
        if (s == k) o = q[k];   // v_cndmask chains: `s` differs between the lanes
    return o;
}

template <int NB, int ORDER = HR_ANY_ORDER>
HR_DEV uint32_t trace_any_seq(bool active, int n_rays, const Node8* __restrict__ nodes, const TriGPU* __restrict__ tris, f3 o, const f3 (&dir)[NB],
                              float t_min, const float (&t_maxs)[NB], uint32_t* wave_stack, int lane, uint32_t entry, DivCounters* dv = nullptr)
{
    RaySeq q[NB];
#pragma unroll
    for (int k = 0; k < NB; k++) q[k] = rayseq_pack(ray_prepare(o, dir[k]), t_maxs[k]);
    RayPre r;
    r.o = o;
    rayseq_unpack(r, q[0]);
    float t_max = q[0].t_max;
    uint32_t  spill_array[HR_SPILL_ENTRIES];
    LaneStack st;
    st.init(wave_stack, lane, spill_array);
    bool      alive = active && entry != HR_NO_ENTRY && n_rays > 0;
    const uint32_t first = (entry << 9) | 1u;
    uint32_t  cur = first, occluded = 0u;
    int       s = 0;
    while (__any(alive))
    {
        if (alive)
        {
            uint32_t ni;
            bool     go = walk_next<ORDER != HR_ORDER_SLOTS>(cur, st, ni);
            if (!go)
            {
                // this ray's stack ran empty: not occluded; next ray of the lane (walk_next on `first` always yields the entry node)
                s++;
                alive = s < n_rays;
                if (alive)
                {
                    const RaySeq nx = rayseq_select<NB>(q, s);
                    rayseq_unpack(r, nx); t_max = nx.t_max;
                    cur = first; st.sp = 0;
                    go  = walk_next<ORDER != HR_ORDER_SLOTS>(cur, st, ni);
                }
            }
            if (go)
            {
                const NodeHits h = test_node<ORDER>(load_node(nodes, ni), r, t_min, t_max);
                HR_DIV(if (dv) div_count(dv->lane_nodes, dv->wave_nodes);)
                uint32_t trimask = walk_expand(h, cur, st);
                bool     hit = false;
                while (trimask)
                {
                    HR_DIV(if (dv) div_count(dv->lane_pairs, dv->wave_pairs);)
                    const uint32_t i0 = (uint32_t)__builtin_ctz(trimask);
                    trimask &= trimask - 1u;
                    const bool     two = trimask != 0u;
                    const uint32_t i1  = two ? (uint32_t)__builtin_ctz(trimask) : i0;
                    trimask &= trimask - 1u;   // no-op on 0
                    const TriRaw ta = load_tri_raw(tris, h.tri_base + i0), tb = load_tri_raw(tris, h.tri_base + i1);
                    float t, u, v;
                    const bool ha = ray_tri_raw<false>(r, ta, t_min, t_max, t, u, v);
                    const bool hb = ray_tri_raw<false>(r, tb, t_min, t_max, t, u, v);
                    if (ha || hb) { hit = true; break; }
                }
                if (hit)
                {
                    occluded |= 1u << s;
                    s++;
                    alive = s < n_rays;
                    if (alive)
                    {
                        const RaySeq nx = rayseq_select<NB>(q, s);
                        rayseq_unpack(r, nx); t_max = nx.t_max;
                        cur = first; st.sp = 0;
                    }
                }
            }
        }
    }
    return occluded;
}
#endif // HR_DEV_PATHS

#ifdef HR_DEV_PATHS   // the persistent-wave shadow trace (shadows.hip k_shadows_trace_pw) and the wavefront queue kernels (trace_queue.h)
// ---- step-wise any-hit traversal (for persistent waves that refill idle lanes from a ray queue) ------------------
struct AnyHitLane
{
    RayPre    r;
    float     t_min, t_max;
    uint32_t  cur;
    LaneStack st;
};

HR_DEV void anyhit_begin(AnyHitLane& s, f3 o, f3 d, float t_min, float t_max, uint32_t* wave_stack, int lane, uint32_t* spill_array)
{
    s.r = ray_prepare(o, d);
    s.t_min = t_min;
    s.t_max = t_max;
    s.st.init(wave_stack, lane, spill_array);
    s.cur = 1u;
}

// One node (box tests + the triangles of its hit leaves).  Returns 0 = keep going, 1 = occluded, 2 = done, no hit.
template <bool STATS>
HR_DEV int anyhit_step(AnyHitLane& s, const Node8* __restrict__ nodes, const TriGPU* __restrict__ tris, uint32_t& n_nodes, uint32_t& n_tris)
{
    uint32_t ni;
    if (!walk_next<HR_ANY_ORDER != HR_ORDER_SLOTS>(s.cur, s.st, ni)) return 2;
    const NodeHits h  = test_node<HR_ANY_ORDER>(load_node(nodes, ni), s.r, s.t_min, s.t_max);
    if (STATS) n_nodes++;
    uint32_t trimask = walk_expand(h, s.cur, s.st);
    while (trimask)
    {
        const uint32_t i = (uint32_t)__builtin_ctz(trimask);
        trimask &= trimask - 1u;
        f3       v0, v1, v2;
        uint32_t prim;
        load_tri(tris, h.tri_base + i, v0, v1, v2, prim);
        if (STATS) n_tris++;
        float t, u, v;
        if (ray_tri<false>(s.r, v0, v1, v2, s.t_min, s.t_max, t, u, v)) return 1;
    }
    return ((s.cur & 0xffu) != 0u || s.st.sp > 0) ? 0 : 2;
}
#endif // HR_DEV_PATHS

struct HitRec
{
    float   t, u, v;
    int32_t prim; // -1 = miss
};

// Closest hit: smallest t, ties broken by the smallest original triangle index (so the answer
// does not depend on traversal order).
// STATS: count node steps / triangle tests into *n_nodes / *n_tris (the instrumented builds behind hr_*_trace_stats).
template <bool STATS = false>
HR_DEV HitRec trace_closest(const Node8* __restrict__ nodes, const TriGPU* __restrict__ tris, f3 o, f3 d, float t_min, float t_max,
                            uint32_t* wave_stack, int lane, DivCounters* dv = nullptr, uint32_t* n_nodes = nullptr, uint32_t* n_tris = nullptr)
{
    RayPre    r = ray_prepare(o, d);
    uint32_t  spill_array[HR_SPILL_ENTRIES];
    LaneStack st;
    st.init(wave_stack, lane, spill_array);
    uint32_t cur = 1u, ni;
    HitRec best;
    best.t = t_max; best.u = 0.0f; best.v = 0.0f; best.prim = -1;
    while (walk_next<true>(cur, st, ni))
    {
        const float    tfar = best.prim < 0 ? t_max : best.t * 1.0000005f;
        const NodeHits h    = test_node<HR_ORDER_NEAR>(load_node(nodes, ni), r, t_min, tfar);
        if (STATS) (*n_nodes)++;
        HR_DIV(if (dv) div_count(dv->lane_nodes, dv->wave_nodes);)
        uint32_t trimask = walk_expand(h, cur, st);
        while (trimask)
        {
            HR_DIV(if (dv) div_count(dv->lane_pairs, dv->wave_pairs);)
            const uint32_t i = (uint32_t)__builtin_ctz(trimask);
            trimask &= trimask - 1u;
            const TriRaw tri = load_tri_raw(tris, h.tri_base + i);
            if (STATS) (*n_tris)++;
            float t, u, v;
            if (ray_tri_raw<true>(r, tri, t_min, t_max, t, u, v))
            {
                const int32_t prim = (int32_t)tri.a.w;
                if (best.prim < 0 || t < best.t || (t == best.t && prim < best.prim)) { best.t = t; best.u = u; best.v = v; best.prim = prim; }
            }
        }
    }
    return best;
}

// ---- wave-cooperative traversal: deferred, redistributed triangle tests -----------------------------------------------------
// tools/divergence.py on the hit-shading passes: the node loop of a wave runs at 25-50 % lane utilisation, the triangle loop
// nested in it at 5-7 % (a handful of lanes have leaf hits in any one step, the other lanes wait) — and it was half of the
// instructions a wave issued.  Here a lane does not test its own triangles: it appends (lane, triangle) jobs to a ring in LDS and
// keeps walking; whenever the ring holds a job for every lane of the wave, ALL lanes — including those whose ray has finished or
// that never had one — each take one job, fetch the owner's ray through ds_bpermute, run the same watertight test (bit-identical
// decisions: same code, same operands), and hand the result back through LDS (closest hit: 64-bit ds_min of (t, prim), the
// reference's tie rule).  The far limit a lane culls nodes with lags by at most one flush; that only costs node visits, never a hit.
// The caller keeps the wave converged around the call (inactive lanes pass active = false and serve as job lanes).
#ifndef HR_COOP_PUSH
#define HR_COOP_PUSH 3     // jobs a lane may append per step (a lane with more pending leaf triangles skips node steps until drained).  Round 4, with the
                           // new tree: AO trace 371 / 364 / 351 us for 2 / 3 / 4 (4 needs the 512-entry ring), DDGI and reflections -1 %; round 2: 356 / 333 / 330 (DDGI) for 1 / 2 / 3
#endif
#ifndef HR_COOP_RING
#define HR_COOP_RING (HR_COOP_PUSH <= 3 ? 256 : 512)   // jobs, a power of two; must hold 63 queued + 64 x HR_COOP_PUSH appended in one step
#endif
static_assert((HR_COOP_RING & (HR_COOP_RING - 1)) == 0 && HR_COOP_RING >= 63 + 64 * HR_COOP_PUSH + 1, "cooperative job ring too small for HR_COOP_PUSH");
struct CoopWave
{
    uint32_t           jobs[HR_COOP_RING];   // owner lane << 26 | triangle index (hr_scene_create bounds the triangle count)
    unsigned long long key[64];              // closest: ordered(t) << 32 | prim, ~0 = miss;  any-hit: 0 = occluded
    float2             uv[64];
};
constexpr uint32_t kCoopMaxTriangles = 1u << 26;

HR_DEV uint32_t float_ordered(float f) { const uint32_t b = __float_as_uint(f); return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u); }
HR_DEV float    ordered_float(uint32_t k) { return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu)); }

template <bool ANY>
HR_DEV void coop_flush(CoopWave& cw, const TriGPU* __restrict__ tris, const RayPre& r, float t_min, float t_max, uint32_t rank, uint32_t head, uint32_t n, int lane)
{
    const bool     mine  = rank < n;
    const uint32_t job   = mine ? cw.jobs[(head + rank) & (HR_COOP_RING - 1)] : ((uint32_t)lane << 26);
    const int      owner = (int)(job >> 26);
    RayPre q;
    q.o.x = __shfl(r.o.x, owner); q.o.y = __shfl(r.o.y, owner); q.o.z = __shfl(r.o.z, owner);
    q.Sx  = __shfl(r.Sx, owner);  q.Sy  = __shfl(r.Sy, owner);  q.Sz  = __shfl(r.Sz, owner);
    const int kp = __shfl(r.kx | (r.ky << 2) | (r.kz << 4), owner);
    q.kx = kp & 3; q.ky = (kp >> 2) & 3; q.kz = kp >> 4;
    const float q_min = __shfl(t_min, owner), q_max = __shfl(t_max, owner);
    bool     hit = false;
    float    t = 0.0f, u = 0.0f, v = 0.0f;
    uint32_t prim = 0u;
    if (mine)
    {
        const TriRaw tr = load_tri_raw(tris, job & (kCoopMaxTriangles - 1u));
        hit  = ray_tri_raw<!ANY>(q, tr, q_min, q_max, t, u, v);
        prim = tr.a.w;
    }
    if (ANY)
    {
        if (hit) cw.key[owner] = 0ull;
    }
    else
    {
        const unsigned long long k = ((unsigned long long)float_ordered(t) << 32) | prim;
        if (hit) atomicMin(&cw.key[owner], k);
        wave_fence();
        if (hit && cw.key[owner] == k) cw.uv[owner] = make_float2(u, v);
    }
    wave_fence();
}

// Closest hit (ANY = false: smallest t, ties to the smallest original triangle index — trace_closest's answer bit for bit) or
// any-hit (ANY = true: HitRec.prim = 0 if occluded, -1 if not; t, u, v unset).
template <bool ANY, int ORDER = (ANY ? HR_ANY_ORDER : HR_ORDER_NEAR)>
HR_DEV HitRec trace_coop(bool active, const Node8* __restrict__ nodes, const TriGPU* __restrict__ tris, f3 o, f3 d, float t_min, float t_max,
                         uint32_t* wave_stack, CoopWave& cw, int lane, uint32_t entry = 0u, DivCounters* dv = nullptr)
{
    const unsigned long long exec  = __ballot(1);
    const uint32_t           nproc = (uint32_t)__popcll(exec), rank = lanes_below(exec);
    RayPre    r = ray_prepare(o, d);
    uint32_t  spill_array[HR_SPILL_ENTRIES];
    LaneStack st;
    st.init(wave_stack, lane, spill_array);
    bool     alive = active && entry != HR_NO_ENTRY;
    uint32_t cur   = alive ? ((entry << 9) | 1u) : 0u, pend = 0u, pend_base = 0u;
    uint32_t head = 0u, count = 0u;   // wave-uniform
    float    tfar = t_max;
    cw.key[lane] = ~0ull;
    wave_fence();
    for (;;)
    {
        if (alive && pend == 0u)
        {
            uint32_t ni;
            if (walk_next<ORDER != HR_ORDER_SLOTS>(cur, st, ni))
            {
                const NodeHits h = test_node<ORDER>(load_node(nodes, ni), r, t_min, tfar);
                HR_DIV(if (dv) div_count(dv->lane_nodes, dv->wave_nodes);)
                pend      = walk_expand(h, cur, st);
                pend_base = h.tri_base;
            }
            else
                alive = false;
        }
        uint32_t left = (uint32_t)__popc(pend);
        uint32_t pos  = head + count;
#pragma unroll
        for (int k = 0; k < HR_COOP_PUSH; k++)
        {
            const bool               has = left > (uint32_t)k;
            const unsigned long long b   = __ballot(has);
            if (has)
            {
                const uint32_t i = (uint32_t)__builtin_ctz(pend);
                pend &= pend - 1u;
                cw.jobs[(pos + lanes_below(b)) & (HR_COOP_RING - 1)] = ((uint32_t)lane << 26) | (pend_base + i);
            }
            pos += (uint32_t)__popcll(b);
        }
        count = pos - head;
        const bool walking = __ballot(alive) != 0ull;
        bool       flushed = false;
        while (count >= nproc || (!walking && count > 0u))
        {
            const uint32_t n = count < nproc ? count : nproc;
            wave_fence();
            HR_DIV(if (dv) div_count(dv->lane_pairs, dv->wave_pairs);)
            coop_flush<ANY>(cw, tris, r, t_min, t_max, rank, head, n, lane);
            head += n; count -= n;
            flushed = true;
        }
        if (flushed)
        {
            const unsigned long long k = cw.key[lane];
            if (ANY) { if (k == 0ull) { alive = false; pend = 0u; } }
            else if (k != ~0ull) tfar = ordered_float((uint32_t)(k >> 32)) * 1.0000005f;
        }
        if (!walking && count == 0u) break;
    }
    HitRec best;
    best.t = t_max; best.u = 0.0f; best.v = 0.0f; best.prim = -1;
    const unsigned long long k = cw.key[lane];
    if (ANY) { if (k == 0ull) best.prim = 0; }
    else if (k != ~0ull)
    {
        const float2 uv = cw.uv[lane];
        best.t = ordered_float((uint32_t)(k >> 32)); best.u = uv.x; best.v = uv.y; best.prim = (int32_t)(uint32_t)k;
    }
    wave_fence();   // the next call re-initialises key[]
    return best;
}

} // namespace hr
