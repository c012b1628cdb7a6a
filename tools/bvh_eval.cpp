// CPU developer tool: quality of the 8-wide BVH that csrc/bvh_build.cpp builds, WITHOUT a GPU.
//
//   g++ -O2 -std=c++17 -fopenmp -I hybrid_rendering_amd/csrc tools/bvh_eval.cpp hybrid_rendering_amd/csrc/bvh_build.cpp -o tools/_build/bvh_eval
//   tools/_build/bvh_eval scene.f32 [--hard-light]          (scene.f32 = SceneData.verts.tofile(): [n][3][3] float32)
//
// Builds the tree (the builder's HR_BVH_* environment switches apply), then replays the walk of csrc/traverse.h on the host — the same
// stack discipline (one entry per node, internal children first, any-hit in slot order, closest-hit near-to-far along the sort axis),
// the same conservative box test — over ray sets shaped like the four trace passes of the 1080p bench frame:
//   shadows   one any-hit ray per lit pixel of every 4th 8x8 tile towards the directional light (t 0.01 .. 10000)
//   ao        4 cosine-hemisphere any-hit rays per pixel, length 7, started at entry_node_for_box
//   refl      one closest-hit mirror ray per pixel of the half-resolution tiles
//   ddgi      256 spherical-Fibonacci closest-hit rays from every 8th probe of the 16x8x16 grid
// AO_CANDIDATE_STUDY=1 / DDGI_MAP_STUDY=1 in the environment run one of the offline studies quoted in docs/EXPERIMENTS.md R4.4 instead.
// and prints per set: node steps / ray, triangle tests / ray, and the same two summed as max-over-the-64-lanes-of-a-wave (what a SIMD
// issues: the slowest lane of a wave sets its step count), plus the SAH cost of the tree and the build time.
#include "bvh.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace hr;

struct V3 { float x, y, z; };
static inline V3 mk(float x, float y, float z) { return { x, y, z }; }
static inline V3 operator+(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
static inline V3 operator-(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static inline V3 operator*(V3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
static inline V3 norm(V3 a) { float l = std::sqrt(dot(a, a)); return l > 0 ? a * (1.0f / l) : a; }

struct Ray { V3 o, d; float tmin, tmax; };
struct Counts { uint32_t nodes = 0, tris = 0; };

struct Pre { V3 o, id; uint32_t sel; V3 d; };
static Pre prepare(const Ray& r)
{
    Pre p; p.o = r.o; p.d = r.d;
    const float tiny = 1e-18f;
    float dx = std::fabs(r.d.x) < tiny ? (r.d.x < 0 ? -tiny : tiny) : r.d.x;
    float dy = std::fabs(r.d.y) < tiny ? (r.d.y < 0 ? -tiny : tiny) : r.d.y;
    float dz = std::fabs(r.d.z) < tiny ? (r.d.z < 0 ? -tiny : tiny) : r.d.z;
    p.id = mk(1.0f / dx, 1.0f / dy, 1.0f / dz);
    p.sel = (dx < 0 ? 1u : 0u) | (dy < 0 ? 2u : 0u) | (dz < 0 ? 4u : 0u);
    return p;
}

struct Hits { uint32_t hit8, n_internal, rev; };
static Hits test_node(const Node8& n, const Pre& r, float tn0, float tf0, bool ordered)
{
    const float s[3] = { std::ldexp(1.0f, (int)n.ex - 127), std::ldexp(1.0f, (int)n.ey - 127), std::ldexp(1.0f, (int)n.ez - 127) };
    const float o[3] = { n.ox, n.oy, n.oz };
    const float ro[3] = { r.o.x, r.o.y, r.o.z }, id[3] = { r.id.x, r.id.y, r.id.z };
    Hits h; h.n_internal = n.counts & 15u; h.rev = 0;
    if (ordered)
    {
        const uint32_t ax = n.meta[0] & 3u;
        h.rev = (r.sel >> ax) & 1u;
    }
    const int nk = n.counts >> 4;
    uint32_t hits = 0;
    for (int i = 0; i < nk; i++)
    {
        float tn = tn0, tf = tf0;
        for (int a = 0; a < 3; a++)
        {
            const bool neg = (r.sel >> a) & 1u;
            const float qn = neg ? n.qhi[a][i] : n.qlo[a][i], qf = neg ? n.qlo[a][i] : n.qhi[a][i];
            const float A = s[a] * id[a], B = (o[a] - ro[a]) * id[a];
            tn = std::fmax(tn, std::fma(qn, A, B));
            tf = std::fmin(tf, std::fma(qf, A, B));
        }
        if (tn <= tf * 1.0000005f) hits |= 1u << i;
    }
    h.hit8 = hits;
    return h;
}

// Moeller-Trumbore in double: the evaluation only needs hit / t, not the product's bit-exact watertight decisions
static bool tri_hit(const TriGPU& t, const Ray& r, float tmin, float tmax, float& tout)
{
    const double e1[3] = { (double)t.v1[0] - t.v0[0], (double)t.v1[1] - t.v0[1], (double)t.v1[2] - t.v0[2] };
    const double e2[3] = { (double)t.v2[0] - t.v0[0], (double)t.v2[1] - t.v0[1], (double)t.v2[2] - t.v0[2] };
    const double d[3] = { r.d.x, r.d.y, r.d.z };
    const double p[3] = { d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0] };
    const double det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
    if (det == 0.0) return false;
    const double inv = 1.0 / det;
    const double s[3] = { (double)r.o.x - t.v0[0], (double)r.o.y - t.v0[1], (double)r.o.z - t.v0[2] };
    const double u = (s[0] * p[0] + s[1] * p[1] + s[2] * p[2]) * inv;
    if (u < 0.0 || u > 1.0) return false;
    const double q[3] = { s[1] * e1[2] - s[2] * e1[1], s[2] * e1[0] - s[0] * e1[2], s[0] * e1[1] - s[1] * e1[0] };
    const double v = (d[0] * q[0] + d[1] * q[1] + d[2] * q[2]) * inv;
    if (v < 0.0 || u + v > 1.0) return false;
    const double tt = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) * inv;
    if (!(tt > tmin && tt < tmax)) return false;
    tout = (float)tt;
    return true;
}

struct Walk
{
    uint32_t cur, stack[96]; int sp = 0;
    bool next(bool ordered, uint32_t& ni)
    {
        if ((cur & 0xffu) == 0u) { if (sp == 0) return false; cur = stack[--sp]; }
        const uint32_t m = cur & 0xffu;
        const uint32_t i = (ordered && (cur & 0x100u)) ? 31u - (uint32_t)__builtin_clz(m) : (uint32_t)__builtin_ctz(m);
        cur &= ~(1u << i);
        ni = (cur >> 9) + i;
        return true;
    }
    uint32_t expand(const Node8& n, const Hits& h)
    {
        const uint32_t imask = (1u << h.n_internal) - 1u, ih = h.hit8 & imask;
        if (ih) { if (cur & 0xffu) stack[sp++] = cur; cur = (n.child_base << 9) | (h.rev << 8) | ih; }
        uint32_t lh = h.hit8 & ~imask, trimask = 0;
        while (lh)
        {
            const uint32_t i = (uint32_t)__builtin_ctz(lh); lh &= lh - 1u;
            const uint32_t m = n.meta[i];
            trimask |= ((1u << (m >> 5)) - 1u) << (m & 31u);
        }
        return trimask;
    }
};

static bool trace_any(const BuiltBVH& b, const Ray& r, uint32_t entry, Counts& c)
{
    if (entry == 0xffffffffu) return false;
    const Pre p = prepare(r);
    Walk w; w.cur = (entry << 9) | 1u;
    uint32_t ni;
    while (w.next(false, ni))
    {
        const Node8& n = b.nodes[ni];
        const Hits h = test_node(n, p, r.tmin, r.tmax, false);
        c.nodes++;
        uint32_t tm = w.expand(n, h);
        while (tm)
        {
            const uint32_t i = (uint32_t)__builtin_ctz(tm); tm &= tm - 1u;
            c.tris++;
            float t;
            if (tri_hit(b.tris[n.tri_base + i], r, r.tmin, r.tmax, t)) return true;
        }
    }
    return false;
}

// experiment: any-hit order heuristics.  g_blk[ni] = (sum of the triangle areas under node ni) / (half area of its box): how much of
// a ray through the box the subtree is expected to block.  mode 1: every node carries one flag — walk its internal children from
// the high slot down when the blockier half sits there.  mode 2 (upper bound, not implementable with one mask entry per node): the hit
// internal children in descending blockiness.
static std::vector<float> g_blk;
static std::vector<uint8_t> g_anyrev;
static int g_any_mode = 0;
static bool trace_any_x(const BuiltBVH& b, const Ray& r, uint32_t entry, Counts& c)
{
    if (entry == 0xffffffffu) return false;
    const Pre p = prepare(r);
    if (g_any_mode == 1 || g_any_mode == 3 || g_any_mode == 4)
    {
        Walk w; w.cur = (entry << 9) | 1u;
        uint32_t ni;
        while (w.next(true, ni))
        {
            const Node8& n = b.nodes[ni];
            Hits h = test_node(n, p, r.tmin, r.tmax, g_any_mode >= 3);
            if (g_any_mode == 1) h.rev = g_anyrev[ni];
            if (g_any_mode == 4) h.rev ^= 1u;   // far to near
            c.nodes++;
            uint32_t tm = w.expand(n, h);
            while (tm)
            {
                const uint32_t i = (uint32_t)__builtin_ctz(tm); tm &= tm - 1u;
                c.tris++;
                float t;
                if (tri_hit(b.tris[n.tri_base + i], r, r.tmin, r.tmax, t)) return true;
            }
        }
        return false;
    }
    uint32_t stack[512]; int sp = 0;
    stack[sp++] = entry;
    while (sp)
    {
        const uint32_t ni = stack[--sp];
        const Node8& n = b.nodes[ni];
        const Hits h = test_node(n, p, r.tmin, r.tmax, false);
        c.nodes++;
        const uint32_t imask = (1u << h.n_internal) - 1u;
        uint32_t lh = h.hit8 & ~imask;
        while (lh)
        {
            const uint32_t i = (uint32_t)__builtin_ctz(lh); lh &= lh - 1u;
            const uint32_t m = n.meta[i];
            for (uint32_t k = 0; k < (m >> 5); k++)
            {
                c.tris++;
                float t;
                if (tri_hit(b.tris[n.tri_base + (m & 31u) + k], r, r.tmin, r.tmax, t)) return true;
            }
        }
        uint32_t ih = h.hit8 & imask, kids[8]; int nk = 0;
        while (ih) { const uint32_t i = (uint32_t)__builtin_ctz(ih); ih &= ih - 1u; kids[nk++] = n.child_base + i; }
        std::sort(kids, kids + nk, [&](uint32_t x, uint32_t y) { return g_blk[x] < g_blk[y]; });   // ascending: the blockiest is pushed last, popped first
        for (int k = 0; k < nk; k++) stack[sp++] = kids[k];
    }
    return false;
}

static int trace_closest(const BuiltBVH& b, const Ray& r, Counts& c, float& tbest)
{
    const Pre p = prepare(r);
    Walk w; w.cur = 1u;
    uint32_t ni; int best = -1; tbest = r.tmax;
    while (w.next(true, ni))
    {
        const Node8& n = b.nodes[ni];
        const float tf = best < 0 ? r.tmax : tbest * 1.0000005f;
        const Hits h = test_node(n, p, r.tmin, tf, true);
        c.nodes++;
        uint32_t tm = w.expand(n, h);
        while (tm)
        {
            const uint32_t i = (uint32_t)__builtin_ctz(tm); tm &= tm - 1u;
            c.tris++;
            float t;
            const TriGPU& tg = b.tris[n.tri_base + i];
            if (tri_hit(tg, r, r.tmin, r.tmax, t) && (best < 0 || t < tbest || (t == tbest && (int)tg.prim < best))) { tbest = t; best = (int)tg.prim; }
        }
    }
    return best;
}

static double g_lit_nodes = 0, g_occ_nodes = 0, g_lit_rays = 0, g_occ_rays = 0, g_wave_cost = 0, g_wave_cost_no_lit = 0, g_waves_lit_slowest = 0, g_waves_n = 0;   // SHADOW_LIT_STUDY
static double g_entry_levels = 0, g_entry_calls = 0;   // AO entry study (round 5): node tests of the per-pixel descent
static uint32_t entry_node_for_box(const BuiltBVH& b, V3 lo, V3 hi)
{
    uint32_t ni = 0;
#pragma omp atomic
    g_entry_calls += 1;
    for (int depth = 0; depth < 24; depth++)
    {
#pragma omp atomic
        g_entry_levels += 1;
        const Node8& n = b.nodes[ni];
        const float s[3] = { std::ldexp(1.0f, (int)n.ex - 127), std::ldexp(1.0f, (int)n.ey - 127), std::ldexp(1.0f, (int)n.ez - 127) };
        const float o[3] = { n.ox, n.oy, n.oz }, l[3] = { lo.x, lo.y, lo.z }, h[3] = { hi.x, hi.y, hi.z };
        uint32_t ov = 0; const int nk = n.counts >> 4;
        for (int i = 0; i < nk; i++)
        {
            bool in = true;
            for (int a = 0; a < 3; a++) in = in && std::fma((float)n.qlo[a][i], s[a], o[a]) <= h[a] && std::fma((float)n.qhi[a][i], s[a], o[a]) >= l[a];
            if (in) ov |= 1u << i;
        }
        if (!ov) return 0xffffffffu;
        const uint32_t imask = (1u << (n.counts & 15u)) - 1u;
        if ((ov & (ov - 1u)) != 0u || (ov & ~imask) != 0u) break;
        ni = n.child_base + (uint32_t)__builtin_ctz(ov);
    }
    return ni;
}

struct SetStats { double rays = 0, nodes = 0, tris = 0, wnodes = 0, wtris = 0, waves = 0, hits = 0, worst = 0; std::vector<float> wc; };
static void add_wave(SetStats& s, const Counts* c, int n_lanes, int rays_per_lane = 1)
{
    uint32_t mn = 0, mt = 0;
    for (int i = 0; i < n_lanes; i++)
    {
        s.nodes += c[i].nodes; s.tris += c[i].tris;
        mn = std::max(mn, c[i].nodes); mt = std::max(mt, c[i].tris);
    }
    s.wnodes += mn; s.wtris += mt; s.waves += 1;
    // critical path of the wave: its slowest lane's steps (a lane's node steps and triangle tests are serial)
    float cp = 0; for (int i = 0; i < n_lanes; i++) cp = std::max(cp, c[i].nodes * 230.0f + c[i].tris * 80.0f);
    s.wc.push_back(cp); s.worst = std::max(s.worst, (double)cp);
}
static void report(const char* name, const SetStats& s)
{
    std::vector<float> w = s.wc; std::sort(w.begin(), w.end());
    const float p99 = w.empty() ? 0 : w[(size_t)(w.size() * 0.99)], p999 = w.empty() ? 0 : w[(size_t)(w.size() * 0.999)];
    printf("%-8s rays %9.0f  hit %.3f  nodes/ray %7.3f  tris/ray %7.3f  | wave-max: nodes %8.2f tris %8.2f per wave   cost/ray %8.1f  wavecost %10.0f  slowest-lane path: p99 %6.0f p99.9 %6.0f max %6.0f\n", name, s.rays,
           s.hits / std::max(1.0, s.rays), s.nodes / std::max(1.0, s.rays), s.tris / std::max(1.0, s.rays), s.wnodes / std::max(1.0, s.waves), s.wtris / std::max(1.0, s.waves),
           (s.nodes * 230.0 + s.tris * 80.0) / std::max(1.0, s.rays), (s.wnodes * 230.0 + s.wtris * 80.0) / std::max(1.0, s.waves), p99, p999, s.worst);
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: bvh_eval scene.f32 [--hard-light] [--w 1920 --h 1080]\n"); return 2; }
    bool hard_light = false; int W = 1920, H = 1080;
    for (int i = 2; i < argc; i++)
    {
        if (!strcmp(argv[i], "--hard-light")) hard_light = true;
        if (!strcmp(argv[i], "--w") && i + 1 < argc) W = atoi(argv[++i]);
        if (!strcmp(argv[i], "--h") && i + 1 < argc) H = atoi(argv[++i]);
    }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END); const long sz = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<float> pos((size_t)sz / 4);
    if (fread(pos.data(), 1, (size_t)sz, f) != (size_t)sz) return 1;
    fclose(f);
    const int n_tris = (int)(pos.size() / 9);
    BuiltBVH b;
    const auto t0 = std::chrono::steady_clock::now();
    build_bvh8(pos.data(), n_tris, b);
    const double build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    // SAH of the 8-wide tree: sum over nodes of area(node box) (one node step each) + sum over leaf children of area * count
    {
        double root_area = 0, cn = 0, cl = 0; long nleaf = 0, nleaftri = 0;
        std::vector<double> area(b.nodes.size(), 0.0);
        auto ha = [](double x, double y, double z) { return x * y + y * z + z * x; };
        root_area = ha((double)b.hi[0] - b.lo[0], (double)b.hi[1] - b.lo[1], (double)b.hi[2] - b.lo[2]);
        area[0] = root_area;
        for (size_t ni = 0; ni < b.nodes.size(); ni++)
        {
            const Node8& n = b.nodes[ni];
            const double s[3] = { std::ldexp(1.0, (int)n.ex - 127), std::ldexp(1.0, (int)n.ey - 127), std::ldexp(1.0, (int)n.ez - 127) };
            const int nk = n.counts >> 4, nin = n.counts & 15;
            cn += area[ni];
            for (int i = 0; i < nk; i++)
            {
                const double a = ha((n.qhi[0][i] - n.qlo[0][i]) * s[0], (n.qhi[1][i] - n.qlo[1][i]) * s[1], (n.qhi[2][i] - n.qlo[2][i]) * s[2]);
                if (i < nin) area[n.child_base + i] = a;
                else { cl += a * (n.meta[i] >> 5); nleaf++; nleaftri += n.meta[i] >> 5; }
            }
        }
        unsigned long long hsh = 1469598103934665603ull;
        auto mix = [&](const void* ptr, size_t n) { const unsigned char* c = (const unsigned char*)ptr; for (size_t i = 0; i < n; i++) { hsh ^= c[i]; hsh *= 1099511628211ull; } };
        mix(b.nodes.data(), b.nodes.size() * sizeof(Node8)); mix(b.tris.data(), b.tris.size() * sizeof(TriGPU));
        printf("bvh checksum %016llx\n", hsh);
        printf("tris %d refs %zu nodes %zu depth %d  build %.2f s   SAH: nodes %.3f  leaves %.3f  (x root area)  leaves %ld avg %.2f tris\n", n_tris, b.tris.size(), b.nodes.size(), b.max_depth,
               build_s, cn / root_area, cl / root_area, nleaf, (double)nleaftri / std::max(1L, nleaf));
    }

    if (const char* e = getenv("EVAL_ANY_MODE")) g_any_mode = atoi(e);
    if (g_any_mode)
    {
        const size_t nn = b.nodes.size();
        std::vector<double> tarea(nn, 0.0);
        g_blk.assign(nn, 0.f); g_anyrev.assign(nn, 0);
        auto tri_area = [&](const TriGPU& t) { V3 e1 = mk(t.v1[0] - t.v0[0], t.v1[1] - t.v0[1], t.v1[2] - t.v0[2]), e2 = mk(t.v2[0] - t.v0[0], t.v2[1] - t.v0[1], t.v2[2] - t.v0[2]); V3 c = cross(e1, e2); return 0.5 * std::sqrt((double)dot(c, c)); };
        for (size_t ni = nn; ni-- > 0;)   // breadth-first layout: children come after their parent
        {
            const Node8& n = b.nodes[ni];
            const int nk = n.counts >> 4, nin = n.counts & 15;
            double a = 0;
            for (int i = nin; i < nk; i++) for (uint32_t k = 0; k < (uint32_t)(n.meta[i] >> 5); k++) a += tri_area(b.tris[n.tri_base + (n.meta[i] & 31u) + k]);
            for (int i = 0; i < nin; i++) a += tarea[n.child_base + i];
            tarea[ni] = a;
        }
        auto ha = [](double x, double y, double z) { return x * y + y * z + z * x; };
        g_blk[0] = (float)(tarea[0] / ha((double)b.hi[0] - b.lo[0], (double)b.hi[1] - b.lo[1], (double)b.hi[2] - b.lo[2]));
        for (size_t ni = 0; ni < nn; ni++)
        {
            const Node8& n = b.nodes[ni];
            const double s[3] = { std::ldexp(1.0, (int)n.ex - 127), std::ldexp(1.0, (int)n.ey - 127), std::ldexp(1.0, (int)n.ez - 127) };
            const int nin = n.counts & 15;
            double wsum = 0, wpos = 0;
            for (int i = 0; i < nin; i++)
            {
                const double a = ha((n.qhi[0][i] - n.qlo[0][i]) * s[0], (n.qhi[1][i] - n.qlo[1][i]) * s[1], (n.qhi[2][i] - n.qlo[2][i]) * s[2]);
                const double blk = tarea[n.child_base + i] / std::max(a, 1e-30);
                g_blk[n.child_base + i] = (float)blk;
                wsum += blk; wpos += blk * i;
            }
            if (nin > 1 && wpos / wsum > 0.5 * (nin - 1)) g_anyrev[ni] = 1;
        }
    }
    // camera of synth.sponza_camera (frame 0), fov 60, 16:9
    const V3 eye = mk(279.5372f, 75.164913f, -20.101242f);
    const V3 fwd = norm(mk(-1.0f, 0.12f, 0.08f));
    const V3 right = norm(cross(fwd, mk(0, 1, 0))), up = cross(right, fwd);
    const float tanh_ = std::tan(30.0f * 3.14159265f / 180.0f), aspect = (float)W / (float)H;
    V3 L;
    if (hard_light) L = norm(mk(0.82f, 0.36f, 0.44f));
    else
    {
        const double a = 30.0 * M_PI / 180.0, bx = -10.0 * M_PI / 180.0;
        // (Rz Rx)(0,-1,0), negated
        const double v[3] = { 0.0, -std::cos(bx), -std::sin(bx) };   // Rx * (0,-1,0)
        const double d[3] = { std::cos(a) * v[0] - std::sin(a) * v[1], std::sin(a) * v[0] + std::cos(a) * v[1], v[2] };
        L = norm(mk((float)-d[0], (float)-d[1], (float)-d[2]));
    }
    auto primary = [&](float px, float py) {
        const float u = (2.0f * (px + 0.5f) / W - 1.0f) * tanh_ * aspect, v = (1.0f - 2.0f * (py + 0.5f) / H) * tanh_;
        Ray r; r.o = eye; r.d = norm(fwd + right * u + up * v); r.tmin = 1.0f; r.tmax = 1000.0f;
        return r;
    };
    auto tri_normal = [&](int prim, V3 towards) {
        const float* p = pos.data() + (size_t)prim * 9;
        V3 n = norm(cross(mk(p[3] - p[0], p[4] - p[1], p[5] - p[2]), mk(p[6] - p[0], p[7] - p[1], p[8] - p[2])));
        return dot(n, towards) < 0 ? n * -1.0f : n;
    };
    auto hash = [](uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; };
    auto rnd = [&](uint32_t a, uint32_t bq) { return (hash(a * 9781u + bq * 6271u + 17u) >> 8) * (1.0f / 16777216.0f); };

    SetStats sh, ao, rf, gi, pr, r2, g2;
    const int tiles_x = W / 8, tiles_y = H / 8;
    std::vector<int> tile_list;
    for (int ty = 0; ty < tiles_y; ty++)
        for (int tx = 0; tx < tiles_x; tx++)
            if (((tx + 2 * ty) & 3) == 0) tile_list.push_back(ty * tiles_x + tx);
#pragma omp parallel
    {
        SetStats lsh, lao, lrf, lpr, l2;
#pragma omp for schedule(dynamic, 16)
        for (size_t ti = 0; ti < tile_list.size(); ti++)
        {
            const int tx = tile_list[ti] % tiles_x, ty = tile_list[ti] / tiles_x;
            Counts cp[64], cs[64], ca[64], cr[64], c2[64];
            bool   lit_lane[64] = {};
            for (int l = 0; l < 64; l++)
            {
                const int x = tx * 8 + (l & 7), y = ty * 8 + (l >> 3);
                const Ray pr_ = primary((float)x, (float)y);
                float t;
                const int prim = trace_closest(b, pr_, cp[l], t);
                lpr.rays++;
                if (prim < 0) continue;
                lpr.hits++;
                const V3 P = pr_.o + pr_.d * t, N = tri_normal(prim, pr_.d * -1.0f);
                if (dot(N, L) > 0)
                {
                    // soft-shadow disk jitter of radius 0.08 (lighting.glsl:44-47)
                    const float r1 = 0.08f * std::sqrt(rnd(x, y)), a1 = 6.2831853f * rnd(y + 7777u, x);
                    V3 T = norm(cross(L, std::fabs(L.y) < 0.99f ? mk(0, 1, 0) : mk(1, 0, 0))), B = cross(L, T);
                    Ray s; s.o = P + N * 0.5f; s.d = norm(L + T * (r1 * std::cos(a1)) + B * (r1 * std::sin(a1))); s.tmin = 0.01f; s.tmax = 10000.0f;
                    lsh.rays++;
                    const bool occ = g_any_mode ? trace_any_x(b, s, 0u, cs[l]) : trace_any(b, s, 0u, cs[l]);
                    lsh.hits += occ;
                    // round-5 study (SHADOW_LIT_STUDY): how much of the walk belongs to rays that turn out LIT (they walk to the end of the scene)
                    if (getenv("SHADOW_LIT_STUDY"))
                    {
#pragma omp atomic
                        (occ ? g_occ_nodes : g_lit_nodes) += cs[l].nodes;
#pragma omp atomic
                        (occ ? g_occ_rays : g_lit_rays) += 1;
                        lit_lane[l] = !occ;
                    }
                }
                {
                    const V3 o = P + N * 0.3f;
                    // AO_ENTRY_CELL=<c> (round 5 study): the entry node of the grid cell of edge c that holds the origin, grown by the ray length —
                    // what a per-scene table of entry nodes would hand every pixel of the cell instead of the per-pixel descent
                    static const float cell = getenv("AO_ENTRY_CELL") ? (float)atof(getenv("AO_ENTRY_CELL")) : 0.0f;
                    V3 blo = mk(o.x - 7, o.y - 7, o.z - 7), bhi = mk(o.x + 7, o.y + 7, o.z + 7);
                    if (cell > 0.0f)
                    {
                        const V3 c0 = mk(std::floor(o.x / cell) * cell, std::floor(o.y / cell) * cell, std::floor(o.z / cell) * cell);
                        blo = mk(c0.x - 7, c0.y - 7, c0.z - 7); bhi = mk(c0.x + cell + 7, c0.y + cell + 7, c0.z + cell + 7);
                    }
                    const uint32_t entry = entry_node_for_box(b, blo, bhi);
                    V3 T = norm(cross(N, std::fabs(N.y) < 0.99f ? mk(0, 1, 0) : mk(1, 0, 0))), B = cross(N, T);
                    for (int s4 = 0; s4 < 4; s4++)
                    {
                        const float u1 = rnd(x * 4 + s4, y), u2 = rnd(y * 4 + s4 + 999u, x);
                        const float rr = std::sqrt(1.0f - u1), ph = 6.2831853f * u2;
                        Ray s; s.o = o; s.d = norm(T * (rr * std::cos(ph)) + B * (rr * std::sin(ph)) + N * std::sqrt(u1)); s.tmin = 0.01f; s.tmax = 7.0f;
                        lao.rays++;
                        lao.hits += g_any_mode ? trace_any_x(b, s, entry, ca[l]) : trace_any(b, s, entry, ca[l]);
                    }
                }
                if (((x | y) & 1) == 0)   // half resolution: a quarter of the pixels — grouped below as they come
                {
                    Ray s; s.o = P + N * 0.1f; s.d = norm(pr_.d - N * (2.0f * dot(pr_.d, N))); s.tmin = 0.01f; s.tmax = 10000.0f;
                    float t2;
                    lrf.rays++;
                    const int hp = trace_closest(b, s, cr[l], t2);
                    lrf.hits += hp >= 0;
                    if (hp >= 0)
                    {
                        // the hit shader's light ray (lighting.glsl:117-196: origin P + N * 0.1)
                        const V3 hP = s.o + s.d * t2, hN = tri_normal(hp, s.d * -1.0f);
                        if (dot(hN, L) > 0)
                        {
                            Ray q; q.o = hP + hN * 0.1f; q.d = L; q.tmin = 0.01f; q.tmax = 10000.0f;
                            l2.rays++;
                            l2.hits += g_any_mode ? trace_any_x(b, q, 0u, c2[l]) : trace_any(b, q, 0u, c2[l]);
                        }
                    }
                }
            }
            if (getenv("SHADOW_LIT_STUDY"))
            {
                // the wave's slowest lane: a lit ray or an occluded one?  and what the wave would cost if its lit rays were free
                uint32_t mx = 0, mx_occ = 0; bool mx_lit = false;
                for (int l = 0; l < 64; l++)
                {
                    const uint32_t c = cs[l].nodes * 230u + cs[l].tris * 80u;
                    if (c > mx) { mx = c; mx_lit = lit_lane[l]; }
                    if (!lit_lane[l] && c > mx_occ) mx_occ = c;
                }
#pragma omp atomic
                g_wave_cost += mx;
#pragma omp atomic
                g_wave_cost_no_lit += mx_occ;
#pragma omp atomic
                g_waves_lit_slowest += mx_lit ? 1 : 0;
#pragma omp atomic
                g_waves_n += mx > 0 ? 1 : 0;
            }
            add_wave(lpr, cp, 64); add_wave(lsh, cs, 64); add_wave(lao, ca, 64); add_wave(lrf, cr, 64); add_wave(l2, c2, 64);
        }
#pragma omp critical
        {
            for (auto pp : { std::make_pair(&sh, &lsh), std::make_pair(&ao, &lao), std::make_pair(&rf, &lrf), std::make_pair(&pr, &lpr), std::make_pair(&r2, &l2) })
            {
                pp.first->rays += pp.second->rays; pp.first->nodes += pp.second->nodes; pp.first->tris += pp.second->tris; pp.first->wnodes += pp.second->wnodes;
                pp.first->wtris += pp.second->wtris; pp.first->waves += pp.second->waves; pp.first->hits += pp.second->hits;
                pp.first->worst = std::max(pp.first->worst, pp.second->worst); pp.first->wc.insert(pp.first->wc.end(), pp.second->wc.begin(), pp.second->wc.end());
            }
        }
    }
    if (getenv("AO_CANDIDATE_STUDY"))
    {
        // How much geometry can the AO rays of one 8x8 tile reach?  Box of the tile's ray origins grown by the ray length, intersected with
        // the tree: leaves / triangles inside (a per-tile candidate list would replace the per-ray walk: 3.1 node steps + 1.1 triangle tests).
        std::vector<float> leaves, tris, visits;
#pragma omp parallel
        {
            std::vector<float> ll, lt, lv;
#pragma omp for schedule(dynamic, 16)
            for (size_t ti = 0; ti < tile_list.size(); ti++)
            {
                const int tx = tile_list[ti] % tiles_x, ty = tile_list[ti] / tiles_x;
                V3 lo = mk(1e30f, 1e30f, 1e30f), hi = mk(-1e30f, -1e30f, -1e30f);
                int live = 0;
                for (int l = 0; l < 64; l++)
                {
                    const Ray pr_ = primary((float)(tx * 8 + (l & 7)), (float)(ty * 8 + (l >> 3)));
                    float t; Counts c;
                    const int prim = trace_closest(b, pr_, c, t);
                    if (prim < 0) continue;
                    const V3 P = pr_.o + pr_.d * t, N = tri_normal(prim, pr_.d * -1.0f), o = P + N * 0.3f;
                    lo = mk(std::fmin(lo.x, o.x), std::fmin(lo.y, o.y), std::fmin(lo.z, o.z)); hi = mk(std::fmax(hi.x, o.x), std::fmax(hi.y, o.y), std::fmax(hi.z, o.z));
                    live++;
                }
                if (!live) continue;
                lo = lo - mk(7, 7, 7); hi = hi + mk(7, 7, 7);
                // box query
                uint32_t stack[256]; int sp = 0; stack[sp++] = 0;
                int nl = 0, nt = 0, nv = 0;
                while (sp)
                {
                    const Node8& n = b.nodes[stack[--sp]];
                    nv++;
                    const float sc[3] = { std::ldexp(1.0f, (int)n.ex - 127), std::ldexp(1.0f, (int)n.ey - 127), std::ldexp(1.0f, (int)n.ez - 127) };
                    const float oo[3] = { n.ox, n.oy, n.oz }, l3[3] = { lo.x, lo.y, lo.z }, h3[3] = { hi.x, hi.y, hi.z };
                    const int nk = n.counts >> 4, ni = n.counts & 15;
                    for (int i = 0; i < nk; i++)
                    {
                        bool in = true;
                        for (int a = 0; a < 3; a++) in = in && std::fma((float)n.qlo[a][i], sc[a], oo[a]) <= h3[a] && std::fma((float)n.qhi[a][i], sc[a], oo[a]) >= l3[a];
                        if (!in) continue;
                        if (i < ni) { if (sp < 256) stack[sp++] = n.child_base + i; }
                        else { nl++; nt += n.meta[i] >> 5; }
                    }
                }
                ll.push_back((float)nl); lt.push_back((float)nt); lv.push_back((float)nv);
            }
#pragma omp critical
            { leaves.insert(leaves.end(), ll.begin(), ll.end()); tris.insert(tris.end(), lt.begin(), lt.end()); visits.insert(visits.end(), lv.begin(), lv.end()); }
        }
        auto pct = [](std::vector<float> v, double q) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0f : v[(size_t)((v.size() - 1) * q)]; };
        printf("AO candidate study: %zu tiles; per tile (origins' box + ray length 7):\n", leaves.size());
        printf("  leaves     p10 %6.0f  p50 %6.0f  p90 %6.0f  p99 %6.0f  max %6.0f\n", pct(leaves, 0.1), pct(leaves, 0.5), pct(leaves, 0.9), pct(leaves, 0.99), pct(leaves, 1.0));
        printf("  triangles  p10 %6.0f  p50 %6.0f  p90 %6.0f  p99 %6.0f  max %6.0f\n", pct(tris, 0.1), pct(tris, 0.5), pct(tris, 0.9), pct(tris, 0.99), pct(tris, 1.0));
        printf("  nodes the query visits  p50 %6.0f  p90 %6.0f  max %6.0f\n", pct(visits, 0.5), pct(visits, 0.9), pct(visits, 1.0));
        return 0;
    }
    if (getenv("DDGI_MAP_STUDY"))
    {
        // Which 64 rays should share a wave?  All 16x8x16 probes x 256 rays are walked once (probe rays + the hit points' light / sky rays),
        // then the per-ray step counts are summed as max-over-lanes under several thread -> (probe, ray) mappings.
        const int gx = 16, gy = 8, gz = 16, NP = gx * gy * gz, R = 256;
        std::vector<Counts> c1((size_t)NP * R), c2((size_t)NP * R);
#pragma omp parallel for schedule(dynamic, 4)
        for (int p = 0; p < NP; p++)
        {
            const int ix = p % gx, iy = (p / gx) % gy, iz = p / (gx * gy);
            const V3 o = mk(b.lo[0] + (b.hi[0] - b.lo[0]) * (ix + 0.5f) / gx, b.lo[1] + (b.hi[1] - b.lo[1]) * (iy + 0.5f) / gy, b.lo[2] + (b.hi[2] - b.lo[2]) * (iz + 0.5f) / gz);
            for (int i = 0; i < R; i++)
            {
                const float phi = 6.2831853f * std::fmod(i * 0.61803398875f, 1.0f), ct = 1.0f - (2.0f * i + 1.0f) / 256.0f, st = std::sqrt(std::fmax(0.0f, 1.0f - ct * ct));
                Ray s; s.o = o; s.d = mk(std::cos(phi) * st, std::sin(phi) * st, ct); s.tmin = 0.001f; s.tmax = 10000.0f;
                float t;
                Counts& a1 = c1[(size_t)p * R + i]; Counts& a2 = c2[(size_t)p * R + i];
                const int hp = trace_closest(b, s, a1, t);
                if (hp >= 0)
                {
                    const V3 hP = s.o + s.d * t, hN = tri_normal(hp, s.d * -1.0f);
                    if (dot(hN, L) > 0)
                    {
                        Ray q; q.o = hP + hN * 0.1f; q.d = L; q.tmin = 0.01f; q.tmax = 10000.0f;
                        g_any_mode ? trace_any_x(b, q, 0u, a2) : trace_any(b, q, 0u, a2);
                    }
                    V3 T = norm(cross(hN, std::fabs(hN.y) < 0.99f ? mk(0, 1, 0) : mk(1, 0, 0))), B = cross(hN, T);
                    const float u1 = rnd(i * 31 + p, 5u), u2 = rnd(p, i + 77u), rr = std::sqrt(1.0f - u1), ph = 6.2831853f * u2;
                    Ray q; q.o = hP + hN * 0.1f; q.d = norm(T * (rr * std::cos(ph)) + B * (rr * std::sin(ph)) + hN * std::sqrt(u1)); q.tmin = 0.01f; q.tmax = 10000.0f;
                    Counts a3;
                    g_any_mode ? trace_any_x(b, q, 0u, a3) : trace_any(b, q, 0u, a3);
                    // the two any-hit traversals of a lane run one after the other, each as a wave-level loop: keep them apart
                    a2.nodes |= a3.nodes << 16; a2.tris |= a3.tris << 16;
                }
            }
        }
        auto eval = [&](const char* name, auto&& lane_of) {
            // lane_of(wave, lane) -> index into c1 / c2
            double w1n = 0, w1t = 0, w2n = 0, w2t = 0, w3n = 0, w3t = 0;
            const int n_waves = NP * R / 64;
            for (int w = 0; w < n_waves; w++)
            {
                uint32_t m1n = 0, m1t = 0, m2n = 0, m2t = 0, m3n = 0, m3t = 0;
                for (int l = 0; l < 64; l++)
                {
                    const size_t k = lane_of(w, l);
                    m1n = std::max(m1n, c1[k].nodes); m1t = std::max(m1t, c1[k].tris);
                    m2n = std::max(m2n, c2[k].nodes & 0xffffu); m2t = std::max(m2t, c2[k].tris & 0xffffu);
                    m3n = std::max(m3n, c2[k].nodes >> 16); m3t = std::max(m3t, c2[k].tris >> 16);
                }
                w1n += m1n; w1t += m1t; w2n += m2n; w2t += m2t; w3n += m3n; w3t += m3t;
            }
            printf("%-28s probe rays: wave-max nodes %6.2f tris %6.2f | light rays %6.2f %6.2f | sky rays %6.2f %6.2f | wave cost %8.0f\n", name, w1n / n_waves, w1t / n_waves,
                   w2n / n_waves, w2t / n_waves, w3n / n_waves, w3t / n_waves, ((w1n + w2n + w3n) * 230.0 + (w1t + w2t + w3t) * 80.0) / n_waves);
        };
        eval("probe-major (shipping)", [&](int w, int l) { return (size_t)w * 64 + l; });
        eval("ray-major, 64 linear probes", [&](int w, int l) { const int i = w % R, pb = w / R; return (size_t)(pb * 64 + l) * R + i; });
        eval("ray-major, 4x4x4 probes", [&](int w, int l) {
            const int i = w % R, blk = w / R, bx = blk % (gx / 4), by = (blk / (gx / 4)) % (gy / 4), bz = blk / ((gx / 4) * (gy / 4));
            const int px = bx * 4 + (l & 3), py = by * 4 + ((l >> 2) & 3), pz = bz * 4 + (l >> 4);
            return (size_t)(px + gx * (py + gy * pz)) * R + i; });
        eval("ray-major, 8x8x1 probes (xz)", [&](int w, int l) {
            const int i = w % R, blk = w / R, bx = blk % (gx / 8), bz = (blk / (gx / 8)) % (gz / 8), py = blk / ((gx / 8) * (gz / 8));
            const int px = bx * 8 + (l & 7), pz = bz * 8 + (l >> 3);
            return (size_t)(px + gx * (py + gy * pz)) * R + i; });
        eval("16 probes x 4 adjacent rays", [&](int w, int l) {
            const int ig = w % (R / 4), pb = w / (R / 4); return (size_t)(pb * 16 + (l >> 2)) * R + ig * 4 + (l & 3); });
        {
            // lower bound: all rays sorted by cost
            std::vector<size_t> idx((size_t)NP * R); for (size_t k = 0; k < idx.size(); k++) idx[k] = k;
            std::sort(idx.begin(), idx.end(), [&](size_t a_, size_t b_) { return c1[a_].nodes * 230 + c1[a_].tris * 80 < c1[b_].nodes * 230 + c1[b_].tris * 80; });
            eval("all rays sorted by probe-ray cost", [&](int w, int l) { return idx[(size_t)w * 64 + l]; });
        }
        return 0;
    }
    // DDGI: 16x8x16 probes over the scene box (ddgi.cpp:173-237 derives the grid from the extents), every 8th probe, 256 rays
    {
        const int gx = 16, gy = 8, gz = 16;
        std::vector<int> probes;
        for (int p = 0; p < gx * gy * gz; p += 8) probes.push_back(p);
#pragma omp parallel
        {
            SetStats l, lg2;
#pragma omp for schedule(dynamic, 4)
            for (size_t pi = 0; pi < probes.size(); pi++)
            {
                const int p = probes[pi], ix = p % gx, iy = (p / gx) % gy, iz = p / (gx * gy);
                const V3 o = mk(b.lo[0] + (b.hi[0] - b.lo[0]) * (ix + 0.5f) / gx, b.lo[1] + (b.hi[1] - b.lo[1]) * (iy + 0.5f) / gy, b.lo[2] + (b.hi[2] - b.lo[2]) * (iz + 0.5f) / gz);
                for (int w = 0; w < 4; w++)
                {
                    Counts c[64], cg[64];
                    for (int k = 0; k < 64; k++)
                    {
                        const int i = w * 64 + k;
                        const float phi = 6.2831853f * std::fmod(i * 0.61803398875f, 1.0f), ct = 1.0f - (2.0f * i + 1.0f) / 256.0f, st = std::sqrt(std::fmax(0.0f, 1.0f - ct * ct));
                        Ray s; s.o = o; s.d = mk(std::cos(phi) * st, std::sin(phi) * st, ct); s.tmin = 0.001f; s.tmax = 10000.0f;
                        float t;
                        l.rays++;
                        const int hp = trace_closest(b, s, c[k], t);
                        l.hits += hp >= 0;
                        if (hp >= 0)
                        {
                            const V3 hP = s.o + s.d * t, hN = tri_normal(hp, s.d * -1.0f);
                            if (dot(hN, L) > 0)
                            {
                                Ray q; q.o = hP + hN * 0.1f; q.d = L; q.tmin = 0.01f; q.tmax = 10000.0f;
                                lg2.rays++;
                                lg2.hits += g_any_mode ? trace_any_x(b, q, 0u, cg[k]) : trace_any(b, q, 0u, cg[k]);
                            }
                            {   // sky visibility ray, cosine-sampled around the normal
                                V3 T = norm(cross(hN, std::fabs(hN.y) < 0.99f ? mk(0, 1, 0) : mk(1, 0, 0))), B = cross(hN, T);
                                const float u1 = rnd(i * 31 + p, 5u), u2 = rnd(p, i + 77u), rr = std::sqrt(1.0f - u1), ph = 6.2831853f * u2;
                                Ray q; q.o = hP + hN * 0.1f; q.d = norm(T * (rr * std::cos(ph)) + B * (rr * std::sin(ph)) + hN * std::sqrt(u1)); q.tmin = 0.01f; q.tmax = 10000.0f;
                                lg2.rays++;
                                lg2.hits += g_any_mode ? trace_any_x(b, q, 0u, cg[k]) : trace_any(b, q, 0u, cg[k]);
                            }
                        }
                    }
                    add_wave(l, c, 64); add_wave(lg2, cg, 64);
                }
            }
#pragma omp critical
            { g2.wc.insert(g2.wc.end(), lg2.wc.begin(), lg2.wc.end()); gi.wc.insert(gi.wc.end(), l.wc.begin(), l.wc.end()); g2.worst = std::max(g2.worst, lg2.worst); gi.worst = std::max(gi.worst, l.worst);
              g2.rays += lg2.rays; g2.nodes += lg2.nodes; g2.tris += lg2.tris; g2.wnodes += lg2.wnodes; g2.wtris += lg2.wtris; g2.waves += lg2.waves; g2.hits += lg2.hits;
              gi.rays += l.rays; gi.nodes += l.nodes; gi.tris += l.tris; gi.wnodes += l.wnodes; gi.wtris += l.wtris; gi.waves += l.waves; gi.hits += l.hits; }
        }
    }
    if (getenv("SHADOW_LIT_STUDY"))
        printf("shadow rays: lit %.0f rays, %.2f nodes/ray; occluded %.0f rays, %.2f nodes/ray; waves whose slowest lane is a LIT ray: %.1f %%; wave cost if lit rays were free: %.1f %% of today's\n",
               g_lit_rays, g_lit_nodes / std::max(1.0, g_lit_rays), g_occ_rays, g_occ_nodes / std::max(1.0, g_occ_rays), 100.0 * g_waves_lit_slowest / std::max(1.0, g_waves_n),
               100.0 * g_wave_cost_no_lit / std::max(1.0, g_wave_cost));
    printf("AO entry descent: %.2f node tests per pixel (%.0f pixels)\n", g_entry_levels / std::max(1.0, g_entry_calls), g_entry_calls);
    report("primary", pr); report("shadows", sh); report("ao", ao); report("refl", rf); report("ddgi", gi);
    report("refl-vis", r2); report("ddgi-vis", g2);
    return 0;
}
