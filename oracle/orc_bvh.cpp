// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_bvh.h header).
#include "orc_bvh.h"
#include <algorithm>
#include <cfloat>

namespace orc {

static thread_local TraversalStats tl_stats;
TraversalStats& traversal_stats() { return tl_stats; }

namespace {
struct AABB
{
    float lo[3], hi[3];
    void  reset()
    {
        for (int a = 0; a < 3; a++) { lo[a] = FLT_MAX; hi[a] = -FLT_MAX; }
    }
    void grow(const float* p)
    {
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); }
    }
    void grow(const AABB& b)
    {
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); }
    }
    float area() const
    {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0) return 0.0f;
        return 2.0f * (dx * dy + dy * dz + dz * dx);
    }
};
} // namespace

void Scene::build(const float* verts, int n_tris)
{
    tris.resize(n_tris);
    order.clear();
    order.reserve(n_tris);
    std::vector<AABB>  tb(n_tris);
    std::vector<float> cen((size_t)n_tris * 3);
    AABB               all;
    all.reset();
    for (int i = 0; i < n_tris; i++)
    {
        const float* p = verts + (size_t)i * 9;
        tris[i].v0     = v3(p[0], p[1], p[2]);
        tris[i].v1     = v3(p[3], p[4], p[5]);
        tris[i].v2     = v3(p[6], p[7], p[8]);
        tb[i].reset();
        tb[i].grow(p);
        tb[i].grow(p + 3);
        tb[i].grow(p + 6);
        for (int a = 0; a < 3; a++) cen[(size_t)i * 3 + a] = 0.5f * (tb[i].lo[a] + tb[i].hi[a]);
        // a triangle with a NaN / infinite coordinate is never hit (the watertight test fails on NaN): it stays in `tris` (indices are
        // stable) but gets no place in the tree — its box would turn the bin index below into an out-of-range integer
        bool finite = true;
        for (int k = 0; k < 9; k++) finite = finite && std::isfinite(p[k]);
        if (finite) { all.grow(tb[i]); order.push_back(i); }
    }
    if (order.empty()) { const float z[3] = { 0.0f, 0.0f, 0.0f }; all.grow(z); }
    for (int a = 0; a < 3; a++) { lo[a] = all.lo[a]; hi[a] = all.hi[a]; }
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    pad      = 3e-5f * std::sqrt(dx * dx + dy * dy + dz * dz);
    for (int i = 0; i < n_tris; i++)
        for (int a = 0; a < 3; a++) { tb[i].lo[a] -= pad; tb[i].hi[a] += pad; }

    nodes.clear();
    nodes.reserve((size_t)n_tris * 2);
    nodes.push_back(BVH2Node {});

    struct Job { int node, first, count; };
    std::vector<Job> stack;
    stack.push_back({ 0, 0, (int)order.size() });
    const int NB = 16, MAX_LEAF = 4;
    while (!stack.empty())
    {
        Job j = stack.back();
        stack.pop_back();
        AABB nb, cb;
        nb.reset();
        cb.reset();
        for (int i = j.first; i < j.first + j.count; i++)
        {
            nb.grow(tb[order[i]]);
            cb.grow(&cen[(size_t)order[i] * 3]);
        }
        BVH2Node& N = nodes[j.node];
        for (int a = 0; a < 3; a++) { N.lo[a] = nb.lo[a]; N.hi[a] = nb.hi[a]; }
        if (j.count <= MAX_LEAF)
        {
            N.left  = j.first;
            N.count = j.count;
            continue;
        }
        // binned SAH over the 3 axes
        float best_cost = FLT_MAX;
        int   best_axis = -1, best_split = -1;
        for (int a = 0; a < 3; a++)
        {
            float ext = cb.hi[a] - cb.lo[a];
            if (!(ext > 0.0f)) continue;
            AABB bb[NB];
            int  bc[NB];
            for (int b = 0; b < NB; b++) { bb[b].reset(); bc[b] = 0; }
            float scale = (float)NB / ext;
            for (int i = j.first; i < j.first + j.count; i++)
            {
                int t = order[i];
                int b = std::min(NB - 1, (int)((cen[(size_t)t * 3 + a] - cb.lo[a]) * scale));
                bb[b].grow(tb[t]);
                bc[b]++;
            }
            float la[NB - 1], ra[NB - 1];
            int   lc[NB - 1], rc[NB - 1];
            AABB  acc;
            acc.reset();
            int c = 0;
            for (int b = 0; b < NB - 1; b++) { acc.grow(bb[b]); c += bc[b]; la[b] = acc.area(); lc[b] = c; }
            acc.reset();
            c = 0;
            for (int b = NB - 1; b > 0; b--) { acc.grow(bb[b]); c += bc[b]; ra[b - 1] = acc.area(); rc[b - 1] = c; }
            for (int b = 0; b < NB - 1; b++)
            {
                if (lc[b] == 0 || rc[b] == 0) continue;
                float cost = la[b] * lc[b] + ra[b] * rc[b];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = b; }
            }
        }
        int mid;
        if (best_axis < 0)
        {
            mid = j.first + j.count / 2; // all centroids coincide: median split
        }
        else
        {
            float ext   = cb.hi[best_axis] - cb.lo[best_axis];
            float scale = (float)NB / ext;
            auto  it    = std::partition(order.begin() + j.first, order.begin() + j.first + j.count, [&](int t) {
                int b = std::min(NB - 1, (int)((cen[(size_t)t * 3 + best_axis] - cb.lo[best_axis]) * scale));
                return b <= best_split;
            });
            mid = (int)(it - order.begin());
            if (mid == j.first || mid == j.first + j.count) mid = j.first + j.count / 2;
        }
        int l = (int)nodes.size();
        nodes.push_back(BVH2Node {});
        nodes.push_back(BVH2Node {});
        nodes[j.node].left  = l;
        nodes[j.node].count = 0;
        stack.push_back({ l + 1, mid, j.first + j.count - mid });
        stack.push_back({ l, j.first, mid - j.first });
    }
}

// conservative slab test: returns true if [tn,tf] overlaps [t_min,t_max]
static inline bool slab(const BVH2Node& n, vec3 o, vec3 id, float t_min, float t_max, float* tnear)
{
    float t0 = t_min, t1 = t_max;
    const float oo[3] = { o.x, o.y, o.z }, ii[3] = { id.x, id.y, id.z };
    for (int a = 0; a < 3; a++)
    {
        float ta = (n.lo[a] - oo[a]) * ii[a];
        float tb = (n.hi[a] - oo[a]) * ii[a];
        float tn = ta < tb ? ta : tb;
        float tf = ta < tb ? tb : ta;
        // NaN (0 * inf) => treat the axis as non-restrictive
        if (tn != tn) tn = -INFINITY;
        if (tf != tf) tf = INFINITY;
        tf = tf * 1.0000004f;
        if (tn > t0) t0 = tn;
        if (tf < t1) t1 = tf;
    }
    *tnear = t0;
    return t0 <= t1;
}

bool Scene::any_hit(vec3 o, vec3 d, float t_min, float t_max) const
{
    if (nodes.empty()) return false;
    RayPre r  = ray_prepare(o, d);
    vec3   id = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    int    stk[128];
    int    sp = 0;
    stk[sp++] = 0;
    while (sp)
    {
        const BVH2Node& n = nodes[stk[--sp]];
        float           tn;
        tl_stats.nodes++;
        if (!slab(n, o, id, t_min, t_max, &tn)) continue;
        if (n.count)
        {
            for (int i = 0; i < n.count; i++)
            {
                tl_stats.tris++;
                if (ray_tri(r, tris[order[n.left + i]], t_min, t_max, nullptr, nullptr, nullptr)) return true;
            }
        }
        else
        {
            stk[sp++] = n.left + 1;
            stk[sp++] = n.left;
        }
    }
    return false;
}

bool Scene::any_hit_brute(vec3 o, vec3 d, float t_min, float t_max) const
{
    RayPre r = ray_prepare(o, d);
    for (size_t i = 0; i < tris.size(); i++)
        if (ray_tri(r, tris[i], t_min, t_max, nullptr, nullptr, nullptr)) return true;
    return false;
}

// closest hit: smallest t; ties broken by the smallest original triangle index.
Hit Scene::closest_hit(vec3 o, vec3 d, float t_min, float t_max) const
{
    Hit best { t_max, 0, 0, -1 };
    if (nodes.empty()) return best;
    RayPre r  = ray_prepare(o, d);
    vec3   id = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    int    stk[128];
    int    sp = 0;
    stk[sp++] = 0;
    while (sp)
    {
        const BVH2Node& n = nodes[stk[--sp]];
        float           tn;
        tl_stats.nodes++;
        // cull against the current best (inclusive, so equal-t candidates are still visited)
        if (!slab(n, o, id, t_min, best.prim < 0 ? t_max : best.t * 1.0000004f, &tn)) continue;
        if (n.count)
        {
            for (int i = 0; i < n.count; i++)
            {
                int   p = order[n.left + i];
                float t, u, v;
                tl_stats.tris++;
                if (ray_tri(r, tris[p], t_min, t_max, &t, &u, &v))
                {
                    if (best.prim < 0 || t < best.t || (t == best.t && p < best.prim)) best = Hit { t, u, v, p };
                }
            }
        }
        else
        {
            stk[sp++] = n.left + 1;
            stk[sp++] = n.left;
        }
    }
    return best;
}

Hit Scene::closest_hit_brute(vec3 o, vec3 d, float t_min, float t_max) const
{
    Hit    best { t_max, 0, 0, -1 };
    RayPre r = ray_prepare(o, d);
    for (size_t i = 0; i < tris.size(); i++)
    {
        float t, u, v;
        if (ray_tri(r, tris[i], t_min, t_max, &t, &u, &v))
            if (best.prim < 0 || t < best.t || (t == best.t && (int)i < best.prim)) best = Hit { t, u, v, (int)i };
    }
    return best;
}

} // namespace orc
