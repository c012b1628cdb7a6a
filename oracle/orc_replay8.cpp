// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header): nothing in hybrid_rendering_amd/ or include/ links or calls this file.
//
// CPU replay of the PRODUCT's acceleration structure for bench.py's `cpu_baseline.trace_replay_same_tree` (BASELINE.json north_star: "a CPU
// replay of the same BVH + ray batches on the box's own host cores"; VERDICT r4 "what's missing" #5: the oracle's own replay walks the oracle's
// BVH2, ~53 node visits per ray, not the tree the GPU walks).
//
//   * the tree is built by the product's host builder, compiled into THIS library from its source where it lies
//     (hybrid_rendering_amd/csrc/bvh_build.cpp: SBVH + reinsertion + 8-wide collapse, 80-byte quantised nodes, 48-byte triangles — bvh.h);
//   * the walk restates the discipline of csrc/traverse.h on the host: one stack entry per node (child base | reverse flag | hit mask),
//     internal children before leaves, any-hit children in slot order, the conservative quantised slab test with the far plane scaled by
//     1 + 4e-7;
//   * the ray / triangle decision is the oracle's watertight test (orc_bvh.h ray_tri, Woop-Benthin-Wald in individually rounded fp32), so
//     the any-hit answers are THE answers — a pure function of (ray, triangle set) — and are compared with the oracle's BVH2 answers and
//     with the GPU masks (tests/test_oracle_bvh.py::test_replay8_*, tests/test_gpu_trace.py).
//
// The reference has no counterpart: its traversal is the Vulkan driver's (ray_query.glsl:13-27).
#include "bvh.h"        // hybrid_rendering_amd/csrc (the product's node / triangle layout and builder entry point)
#include "orc_bvh.h"
#include <cmath>
#include <cstring>

namespace {

struct Replay8
{
    hr::BuiltBVH bvh;
};

struct Pre8
{
    float    o[3], id[3];
    uint32_t sel;
};

inline Pre8 prepare8(orc::vec3 o, orc::vec3 d)
{
    Pre8 p;
    const float tiny = 1e-18f;
    const float dd[3] = { d.x, d.y, d.z };
    p.o[0] = o.x; p.o[1] = o.y; p.o[2] = o.z;
    p.sel = 0;
    for (int a = 0; a < 3; a++)
    {
        const float v = std::fabs(dd[a]) < tiny ? (dd[a] < 0.0f ? -tiny : tiny) : dd[a];
        p.id[a] = 1.0f / v;
        if (v < 0.0f) p.sel |= 1u << a;
    }
    return p;
}

// hit mask over the node's (up to 8) children.  Eight children = eight SIMD lanes: the loop over slots is written for the vectoriser and the
// function is cloned for AVX2 + FMA hosts (resolved once at load time); the arithmetic per slot is the same either way
// (fma(q, A, B), max / min against the ray interval), so the mask does not depend on the clone.
__attribute__((target_clones("avx2,fma", "default")))
uint32_t slab8(const hr::Node8& n, const Pre8& r, float t_min, float t_max)
{
    const uint8_t e[3] = { n.ex, n.ey, n.ez };
    const float   org[3] = { n.ox, n.oy, n.oz };
    float tn[8], tf[8];
    for (int i = 0; i < 8; i++) { tn[i] = t_min; tf[i] = t_max; }
    for (int a = 0; a < 3; a++)
    {
        uint32_t bits = (uint32_t)e[a] << 23;   // 2^(e-127)
        float    s;
        std::memcpy(&s, &bits, 4);
        const float A = s * r.id[a], B = (org[a] - r.o[a]) * r.id[a];
        const uint8_t* qn = ((r.sel >> a) & 1u) ? n.qhi[a] : n.qlo[a];
        const uint8_t* qf = ((r.sel >> a) & 1u) ? n.qlo[a] : n.qhi[a];
#pragma omp simd
        for (int i = 0; i < 8; i++)
        {
            const float a_ = __builtin_fmaf((float)qn[i], A, B), b_ = __builtin_fmaf((float)qf[i], A, B);
            tn[i] = a_ > tn[i] ? a_ : tn[i];
            tf[i] = b_ < tf[i] ? b_ : tf[i];
        }
    }
    uint32_t hits = 0;
    for (int i = 0; i < 8; i++) hits |= (tn[i] <= tf[i] * 1.0000005f ? 1u : 0u) << i;
    return hits & ((1u << (n.counts >> 4)) - 1u);
}

inline bool any_hit8(const hr::BuiltBVH& b, orc::vec3 o, orc::vec3 d, float t_min, float t_max, uint64_t* n_nodes, uint64_t* n_tris)
{
    const Pre8        p  = prepare8(o, d);
    const orc::RayPre rp = orc::ray_prepare(o, d);
    uint32_t stack[hr::kMaxTraversalDepth + 8];
    int      sp  = 0;
    uint32_t cur = 1u;   // node 0, mask bit 0
    for (;;)
    {
        if ((cur & 0xffu) == 0u)
        {
            if (sp == 0) return false;
            cur = stack[--sp];
        }
        const uint32_t i = (uint32_t)__builtin_ctz(cur & 0xffu);
        cur &= ~(1u << i);
        const hr::Node8& n = b.nodes[(cur >> 9) + i];
        const uint32_t   hits = slab8(n, p, t_min, t_max);
        if (n_nodes) ++*n_nodes;
        const uint32_t imask = (1u << (n.counts & 15u)) - 1u;
        // leaves of this node first (a hit ends the ray), then descend
        uint32_t lh = hits & ~imask;
        while (lh)
        {
            const uint32_t s = (uint32_t)__builtin_ctz(lh);
            lh &= lh - 1u;
            const uint32_t m = n.meta[s];
            for (uint32_t k = 0; k < (m >> 5); k++)
            {
                const hr::TriGPU& t = b.tris[n.tri_base + (m & 31u) + k];
                const orc::Tri    tr { orc::v3(t.v0[0], t.v0[1], t.v0[2]), orc::v3(t.v1[0], t.v1[1], t.v1[2]), orc::v3(t.v2[0], t.v2[1], t.v2[2]) };
                if (n_tris) ++*n_tris;
                if (orc::ray_tri(rp, tr, t_min, t_max, nullptr, nullptr, nullptr)) return true;
            }
        }
        const uint32_t ih = hits & imask;
        if (ih)
        {
            if (cur & 0xffu) stack[sp++] = cur;
            cur = (n.child_base << 9) | ih;
        }
    }
}

} // namespace

extern "C" {

void* orc_replay8_create(const float* verts, int n_tris)
{
    Replay8* r = new Replay8;
    hr::build_bvh8(verts, n_tris, r->bvh);
    return r;
}

void orc_replay8_destroy(void* h) { delete (Replay8*)h; }

int orc_replay8_num_nodes(const void* h) { return (int)((const Replay8*)h)->bvh.nodes.size(); }
int orc_replay8_num_refs(const void* h) { return (int)((const Replay8*)h)->bvh.tris.size(); }

// rays: [n][8] = origin.xyz, t_max, direction.xyz, t_min (the layout of orc_any_hit_batch).  stats (nullable): {node steps, triangle tests}
void orc_replay8_any_hit_batch(const void* h, int n, const float* rays, uint8_t* out, uint64_t* stats)
{
    const hr::BuiltBVH& b = ((const Replay8*)h)->bvh;
    uint64_t nn = 0, nt = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : nn, nt)
    for (int i = 0; i < n; i++)
    {
        const float* r = rays + (size_t)i * 8;
        uint64_t a = 0, c = 0;
        out[i] = any_hit8(b, orc::v3(r[0], r[1], r[2]), orc::v3(r[4], r[5], r[6]), r[7], r[3], stats ? &a : nullptr, stats ? &c : nullptr) ? 1 : 0;
        nn += a; nt += c;
    }
    if (stats) { stats[0] = nn; stats[1] = nt; }
}

} // extern "C"
