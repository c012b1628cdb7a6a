// Host-side builder of the compressed 8-wide BVH (see bvh.h).  Replaces the driver's
// acceleration-structure build behind dw::RayTracedScene (reference: main.cpp:74,
// common.cpp:355-521).  Steps: binned-SAH binary tree down to single triangles -> SAH-optimal
// collapse to 8-wide nodes with leaf children of <= 4 triangles (dynamic programme) ->
// breadth-first layout with contiguous children / leaf triangles -> 8-bit conservative quantisation.
#include "bvh.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <queue>

namespace hr {
namespace {

struct Box
{
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX };
    float hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    void  add(const float* p)
    {
        for (int a = 0; a < 3; a++)
        {
            if (p[a] < lo[a]) lo[a] = p[a];
            if (p[a] > hi[a]) hi[a] = p[a];
        }
    }
    void add(const Box& b)
    {
        for (int a = 0; a < 3; a++)
        {
            if (b.lo[a] < lo[a]) lo[a] = b.lo[a];
            if (b.hi[a] > hi[a]) hi[a] = b.hi[a];
        }
    }
    double half_area() const
    {
        double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
        if (x < 0) return 0.0;
        return x * y + y * z + z * x;
    }
};

struct Bin2
{
    Box     box;
    int32_t a = -1, b = -1; // children, or -1 for leaf
    int32_t first = 0, count = 0;
};

constexpr int kBins    = 64;   // 32 -> 64: nodes per shadow ray 6.38 -> 6.30, frame -1.3% (8 bins: 7.25, +5%)
constexpr int kMaxLeaf = 4;   // triangles per leaf CHILD of an 8-wide node (count field of the meta byte, 8 x 4 = 32-bit mask)

// SAH bin of a centroid.  `k` = kBins / extent overflows to +inf when the extent is subnormal and the product is then inf or
// NaN (0 * inf): compare in a way that sends both to a valid bin instead of converting them to int (undefined).
static inline int bin_of(float c, float lo, float k)
{
    const float f = (c - lo) * k;
    if (!(f > 0.0f)) return 0;               // also NaN
    if (f >= (float)(kBins - 1)) return kBins - 1;
    return (int)f;
}

struct Builder
{
    const float*          pos;
    std::vector<Box>      tbox;
    std::vector<float>    tcen;
    std::vector<int32_t>  idx;
    std::vector<int32_t>  prim;   // reference -> original triangle
    std::vector<Bin2>     n2;
    int                   bvh2_leaf = kMaxLeaf;   // binary-tree leaf size (1 for the optimal collapse: it forms the leaves)

    // SAH splits down to binary depth kSahDepth, object-median splits below it: a median split halves the count, so the
    // binary tree is never deeper than kSahDepth + ceil(log2 n) <= 40 + 24 = 64 levels.  The 8-wide collapse only removes
    // levels, and the traversal keeps ONE stack entry per level (traverse.h: walk_expand), so HR_STACK_ENTRIES +
    // HR_SPILL_ENTRIES = 64 entries always suffice — also for adversarial input (a chain of slivers of geometrically growing
    // size makes the SAH peel one triangle per level, n levels deep; tests/test_gpu_trace.py::test_degenerate_sliver_chain).
    static constexpr int kSahDepth = 40;
    int sah_depth = kSahDepth;   // HR_BVH_SAH_DEPTH lowers it (developer switch: exercises the median fallback in tests)
    int32_t split(int32_t first, int32_t count, int depth = 0)
    {
        int32_t me = (int32_t)n2.size();
        n2.emplace_back();
        Box nb, cb;
        for (int32_t i = first; i < first + count; i++)
        {
            nb.add(tbox[idx[i]]);
            cb.add(&tcen[(size_t)idx[i] * 3]);
        }
        n2[me].box   = nb;
        n2[me].first = first;
        n2[me].count = count;
        if (count <= bvh2_leaf) return me;

        double  best     = DBL_MAX;
        int     bax      = -1;
        int     bsplit   = 0;
        if (depth >= sah_depth)
        {
            int ax = 0;
            for (int k = 1; k < 3; k++)
                if (cb.hi[k] - cb.lo[k] > cb.hi[ax] - cb.lo[ax]) ax = k;
            const int32_t mid = first + count / 2;
            std::nth_element(idx.begin() + first, idx.begin() + mid, idx.begin() + first + count, [&](int32_t a, int32_t b) {
                const float ca = tcen[(size_t)a * 3 + ax], cb_ = tcen[(size_t)b * 3 + ax];
                return ca < cb_ || (ca == cb_ && a < b);
            });
            const int32_t l = split(first, mid - first, depth + 1);
            const int32_t r = split(mid, first + count - mid, depth + 1);
            n2[me].a = l;
            n2[me].b = r;
            return me;
        }
        for (int ax = 0; ax < 3; ax++)
        {
            float ext = cb.hi[ax] - cb.lo[ax];
            if (!(ext > 0.0f)) continue;
            Box   bbox[kBins];
            int   bcnt[kBins] = { 0 };
            float k           = (float)kBins / ext;
            for (int32_t i = first; i < first + count; i++)
            {
                int t = idx[i];
                const int b = bin_of(tcen[(size_t)t * 3 + ax], cb.lo[ax], k);
                bbox[b].add(tbox[t]);
                bcnt[b]++;
            }
            double rarea[kBins];
            int    rcnt[kBins];
            Box    acc;
            int    c = 0;
            for (int b = kBins - 1; b >= 1; b--)
            {
                acc.add(bbox[b]);
                c += bcnt[b];
                rarea[b] = acc.half_area();
                rcnt[b]  = c;
            }
            Box lacc;
            int lc = 0;
            for (int b = 0; b < kBins - 1; b++)
            {
                lacc.add(bbox[b]);
                lc += bcnt[b];
                if (lc == 0 || rcnt[b + 1] == 0) continue;
                double cost = lacc.half_area() * lc + rarea[b + 1] * rcnt[b + 1];
                if (cost < best) { best = cost; bax = ax; bsplit = b; }
            }
        }
        int32_t mid;
        if (bax < 0) mid = first + count / 2;
        else
        {
            float ext = cb.hi[bax] - cb.lo[bax];
            float k   = (float)kBins / ext;
            float lo  = cb.lo[bax];
            auto  it  = std::partition(idx.begin() + first, idx.begin() + first + count, [&](int32_t t) {
                return bin_of(tcen[(size_t)t * 3 + bax], lo, k) <= bsplit;
            });
            mid = (int32_t)(it - idx.begin());
            if (mid == first || mid == first + count) mid = first + count / 2;
        }
        int32_t l = split(first, mid - first, depth + 1);
        int32_t r = split(mid, first + count - mid, depth + 1);
        n2[me].a  = l;
        n2[me].b  = r;
        return me;
    }
};

// ---- triangle reference splitting ("early split clipping") -----------------------------------------------------------
// Large triangles (walls, floors) drag their whole bounding box into every node above them, and a ray that travels along
// such a surface visits all of those nodes.  Before the SAH build each triangle whose box is longer than `limit` is cut by
// axis-aligned planes into pieces with tight boxes; the BVH is built over the pieces ("references"), every piece points at
// the ORIGINAL triangle, so the ray/triangle test and its results are untouched — only the boxes get tighter.
struct Ref
{
    Box     box;
    int32_t prim;
};

struct Poly
{
    double v[10][3];
    int    n = 0;
};

inline void clip_poly(const Poly& in, int ax, double plane, bool keep_below, Poly& out)
{
    out.n = 0;
    for (int i = 0; i < in.n; i++)
    {
        const double* a = in.v[i];
        const double* b = in.v[(i + 1) % in.n];
        const bool ina = keep_below ? a[ax] <= plane : a[ax] >= plane;
        const bool inb = keep_below ? b[ax] <= plane : b[ax] >= plane;
        if (ina && out.n < 10) { std::memcpy(out.v[out.n++], a, 24); }
        if (ina != inb && out.n < 10)
        {
            const double t = (plane - a[ax]) / (b[ax] - a[ax]);
            double*      q = out.v[out.n++];
            for (int k = 0; k < 3; k++) q[k] = a[k] + t * (b[k] - a[k]);
            q[ax] = plane;
        }
    }
}

inline Box poly_box(const Poly& p, const Box& parent)
{
    Box b;
    for (int i = 0; i < p.n; i++)
    {
        float f[3];
        for (int k = 0; k < 3; k++)
        {
            // outward-rounded float bounds of the double coordinate, clamped to the parent piece
            float lo = (float)p.v[i][k], hi = lo;
            if ((double)lo > p.v[i][k]) lo = std::nextafter(lo, -FLT_MAX);
            if ((double)hi < p.v[i][k]) hi = std::nextafter(hi, FLT_MAX);
            if (lo < b.lo[k]) b.lo[k] = lo;
            if (hi > b.hi[k]) b.hi[k] = hi;
            f[k] = lo;
        }
        (void)f;
    }
    for (int k = 0; k < 3; k++)
    {
        if (b.lo[k] < parent.lo[k]) b.lo[k] = parent.lo[k];
        if (b.hi[k] > parent.hi[k]) b.hi[k] = parent.hi[k];
    }
    return b;
}

void split_refs(const Poly& poly, const Box& box, int32_t prim, float limit, int depth, std::vector<Ref>& out)
{
    int ax = 0;
    for (int k = 1; k < 3; k++)
        if (box.hi[k] - box.lo[k] > box.hi[ax] - box.lo[ax]) ax = k;
    const float ext = box.hi[ax] - box.lo[ax];
    if (!(ext > limit) || depth >= 6 || poly.n < 3)
    {
        out.push_back(Ref { box, prim });
        return;
    }
    const double plane = 0.5 * ((double)box.lo[ax] + (double)box.hi[ax]);
    Poly l, r;
    clip_poly(poly, ax, plane, true, l);
    clip_poly(poly, ax, plane, false, r);
    if (l.n < 3 || r.n < 3)
    {
        out.push_back(Ref { box, prim });
        return;
    }
    split_refs(l, poly_box(l, box), prim, limit, depth + 1, out);
    split_refs(r, poly_box(r, box), prim, limit, depth + 1, out);
}

inline uint8_t exponent_for(float extent)
{
    // smallest e with extent <= 255 * 2^(e-127)
    if (!(extent > 0.0f)) return 1;
    int   ex;
    float m = std::frexp(extent / 255.0f, &ex); // extent/255 = m * 2^ex, m in [0.5,1)
    (void)m;
    int e = ex + 127; // 2^ex >= extent/255
    while (e > 1 && std::ldexp(255.0, e - 1 - 127) >= (double)extent) e--;
    while (std::ldexp(255.0, e - 127) < (double)extent) e++;
    if (e < 1) e = 1;
    if (e > 254) e = 254;
    return (uint8_t)e;
}

} // namespace

void build_bvh8(const float* positions, int n_tris, BuiltBVH& out)
{
    out.nodes.clear();
    out.tris.clear();
    out.max_depth = 0;
    Builder B;
    B.pos = positions;
    Box all;
    for (int i = 0; i < n_tris; i++)
    {
        const float* p = positions + (size_t)i * 9;
        all.add(p); all.add(p + 3); all.add(p + 6);
    }
    if (n_tris == 0)
    {
        for (int a = 0; a < 3; a++) { all.lo[a] = 0; all.hi[a] = 0; }
    }
    for (int a = 0; a < 3; a++) { out.lo[a] = all.lo[a]; out.hi[a] = all.hi[a]; }
    double diag;
    {
        double dx = (double)all.hi[0] - all.lo[0], dy = (double)all.hi[1] - all.lo[1], dz = (double)all.hi[2] - all.lo[2];
        diag = std::sqrt(dx * dx + dy * dy + dz * dz);
        // Boxes are padded well above the fp32 error of the triangle test so that box culling can
        // never reject a triangle the test would accept.
        out.pad = (float)(3e-5 * diag);
        if (!(out.pad > 0.0f)) out.pad = 1e-6f;
    }
    // references: one per triangle, or several tight pieces for triangles longer than diag * split_fraction
    // Off by default: the bench scene is finely tessellated and splitting at diag/16 .. diag/100 changed nodes/ray by
    // < 2% and the trace time by < 1.5% (tools/passbench.py).  HR_BVH_SPLIT=<fraction of the scene diagonal> enables it
    // for scenes with wall-sized triangles (original Sponza: two triangles per wall).
    double split_fraction = 0.0;
    if (const char* e = getenv("HR_BVH_SPLIT")) split_fraction = atof(e);
    const float limit = split_fraction > 0.0 ? (float)(diag * split_fraction) : FLT_MAX;
    std::vector<Ref> refs;
    refs.reserve((size_t)n_tris + n_tris / 4);
    for (int i = 0; i < n_tris; i++)
    {
        const float* p = positions + (size_t)i * 9;
        Box tb;
        tb.add(p); tb.add(p + 3); tb.add(p + 6);
        Poly poly;
        poly.n = 3;
        for (int v = 0; v < 3; v++)
            for (int k = 0; k < 3; k++) poly.v[v][k] = p[v * 3 + k];
        split_refs(poly, tb, i, limit, 0, refs);
    }
    const int n_refs = (int)refs.size();
    out.n_refs = n_refs;
    B.tbox.resize(n_refs);
    B.tcen.resize((size_t)n_refs * 3);
    B.idx.resize(n_refs);
    B.prim.resize(n_refs);
    for (int i = 0; i < n_refs; i++)
    {
        B.tbox[i] = refs[i].box;
        B.prim[i] = refs[i].prim;
        for (int a = 0; a < 3; a++)
        {
            B.tcen[(size_t)i * 3 + a] = 0.5f * (B.tbox[i].lo[a] + B.tbox[i].hi[a]);
            B.tbox[i].lo[a] -= out.pad; B.tbox[i].hi[a] += out.pad;
        }
        B.idx[i] = i;
    }

    if (n_tris == 0)
    {
        Node8 n;
        std::memset(&n, 0, sizeof(n));
        n.ex = n.ey = n.ez = 1;
        out.nodes.push_back(n);
        return;
    }
    B.bvh2_leaf = getenv("HR_BVH_GREEDY") ? kMaxLeaf : 1;
    if (const char* e = getenv("HR_BVH_SAH_DEPTH")) { const int v = atoi(e); if (v >= 0 && v < Builder::kSahDepth) B.sah_depth = v; }
    B.n2.reserve((size_t)n_refs * 2);
    int32_t root2 = B.split(0, n_refs);

    // ---- optimal collapse (Ylitie, Karras, Laine 2017, sec. 3.1): dynamic programme over the binary tree -----------------
    // c(n, i) = least SAH cost of representing the subtree of n by a forest of at most i roots, a root being either a leaf
    // child (<= kMaxLeaf triangles: area * count * C_prim) or an 8-wide node (area * C_node + the best forest of <= 8 roots
    // below it).  The greedy largest-area-first collapse left 45% of the nodes with two children (bottom nodes holding two
    // leaves): every slot of a node is box-tested anyway, so half-empty nodes are pure overhead.
    const bool optimal = !getenv("HR_BVH_GREEDY");   // developer switch: the previous greedy collapse
    std::vector<double>  cost;   // double: areas of a scene spanning many orders of magnitude overflow a float (inf <= inf made over-full leaves)
    std::vector<uint8_t> best_k, use_split, as_leaf;
    if (optimal)
    {
        const size_t nn = B.n2.size();
        // one node step ~230 VALU + 80 B, one triangle test ~80 VALU + 48 B; the result is flat in C_prim (0.15 .. 1.2: 0.253-0.256 ms)
        const double C_node = 1.0, C_prim = 0.35;
        cost.assign(nn * 8, 0.0); best_k.assign(nn * 8, 0); use_split.assign(nn * 8, 0); as_leaf.assign(nn, 0);
        for (size_t r = nn; r-- > 0;)   // children are allocated after their parent: reverse order is bottom-up
        {
            const Bin2& c = B.n2[r];
            const double area = c.box.half_area();
            const double leaf = c.count <= kMaxLeaf ? area * c.count * C_prim : 1e300;
            if (c.a < 0)
            {
                for (int i = 0; i < 8; i++) cost[r * 8 + i] = leaf;
                as_leaf[r] = 1;
                continue;
            }
            double dist[9];   // dist[j]: children of r as a forest of <= j roots, j = 2..8
            for (int j = 2; j <= 8; j++)
            {
                double b = 1e300; int bk = 1;
                for (int k = 1; k < j; k++)
                {
                    const double v = cost[(size_t)c.a * 8 + (k - 1)] + cost[(size_t)c.b * 8 + (j - k - 1)];
                    if (v < b) { b = v; bk = k; }
                }
                dist[j] = b;
                best_k[r * 8 + (j - 1)] = (uint8_t)bk;
            }
            const double internal = dist[8] + area * C_node;
            as_leaf[r] = c.count <= kMaxLeaf && leaf <= internal;
            double prev = as_leaf[r] ? leaf : internal;
            cost[r * 8 + 0] = prev;
            for (int i = 2; i <= 7; i++)
            {
                if (dist[i] < prev) { prev = dist[i]; use_split[r * 8 + (i - 1)] = 1; }
                cost[r * 8 + (i - 1)] = prev;
            }
            cost[r * 8 + 7] = cost[r * 8 + 6];   // budget 8 only ever splits (used for the children of a node: best_k[.][7])
            use_split[r * 8 + 7] = 1;
        }
    }

    // ---- collapse to 8-wide, breadth-first -----------------------------------------------
    struct Pending { int32_t n2; int32_t n8; int depth; };
    std::queue<Pending> q;
    out.nodes.emplace_back();
    q.push({ root2, 0, 1 });
    out.tris.reserve(n_refs);
    while (!q.empty())
    {
        Pending pd = q.front();
        q.pop();
        if (pd.depth > out.max_depth) out.max_depth = pd.depth;
        int32_t kids[8];
        bool    kid_internal[8];
        int     nk = 0;
        if (optimal)
        {
            // children of this 8-wide node = the minimum-cost forest of at most 8 roots under the binary node (dp below)
            if (B.n2[pd.n2].a < 0) { kids[nk] = pd.n2; kid_internal[nk++] = false; }
            else
            {
                struct Item { int32_t n; int budget; };
                Item stack[32];
                int  sp = 0;
                auto push_split = [&](int32_t n, int budget) {
                    const int k = best_k[(size_t)n * 8 + (budget - 1)];
                    stack[sp++] = { B.n2[n].b, budget - k };
                    stack[sp++] = { B.n2[n].a, k };
                };
                push_split(pd.n2, 8);
                while (sp > 0)
                {
                    Item it = stack[--sp];
                    const Bin2& c = B.n2[it.n];
                    int budget = it.budget;
                    if (c.a < 0) { kids[nk] = it.n; kid_internal[nk++] = false; continue; }
                    while (budget > 1 && !use_split[(size_t)it.n * 8 + (budget - 1)]) budget--;   // c(n,i) == c(n,i-1)
                    if (budget == 1)
                    {
                        kids[nk] = it.n;
                        kid_internal[nk++] = !as_leaf[it.n];
                    }
                    else push_split(it.n, budget);
                }
            }
        }
        else
        {
        if (B.n2[pd.n2].a < 0) kids[nk++] = pd.n2; // root is itself a leaf
        else { kids[nk++] = B.n2[pd.n2].a; kids[nk++] = B.n2[pd.n2].b; }
        while (nk < 8)
        {
            int    best = -1;
            double ba   = -1.0;
            for (int i = 0; i < nk; i++)
                if (B.n2[kids[i]].a >= 0)
                {
                    double ar = B.n2[kids[i]].box.half_area();
                    if (ar > ba) { ba = ar; best = i; }
                }
            if (best < 0) break;
            int32_t k   = kids[best];
            kids[best]  = B.n2[k].a;
            kids[nk++]  = B.n2[k].b;
        }
        for (int i = 0; i < nk; i++) kid_internal[i] = B.n2[kids[i]].a >= 0;
        }
        // kids[] and kid_internal[] are permuted together below
        {
            int32_t ki[8], kl[8];
            int     ni_ = 0, nl_ = 0;
            for (int i = 0; i < nk; i++) (kid_internal[i] ? ki[ni_++] : kl[nl_++]) = kids[i];
            for (int i = 0; i < ni_; i++) { kids[i] = ki[i]; kid_internal[i] = true; }
            for (int i = 0; i < nl_; i++) { kids[ni_ + i] = kl[i]; kid_internal[ni_ + i] = false; }
        }
        const int n_int_kids = (int)std::count(kid_internal, kid_internal + nk, true);
        // slots: internal children first (slot i <-> node child_base + i), then the leaves — the traversal keeps
        // one (child_base, hit mask) stack entry per node instead of one entry per child
        int32_t* kids_mid = kids + n_int_kids;
        Box nb;
        for (int i = 0; i < nk; i++) nb.add(B.n2[kids[i]].box);
        Node8 n;
        std::memset(&n, 0, sizeof(n));
        n.ox = nb.lo[0]; n.oy = nb.lo[1]; n.oz = nb.lo[2];
        n.ex = exponent_for(nb.hi[0] - nb.lo[0]);
        n.ey = exponent_for(nb.hi[1] - nb.lo[1]);
        n.ez = exponent_for(nb.hi[2] - nb.lo[2]);
        const uint8_t eb[3] = { n.ex, n.ey, n.ez };
        {
            // Internal children are ordered along the node's longest axis — the axis with the largest scale exponent, first
            // one on ties, which the traversal re-derives from the exponent bytes — so that a ray can walk them near to far
            // (ascending slots for a positive direction component, descending for a negative one).
            int ax = 0;
            if (eb[1] > eb[0]) ax = 1;
            if (eb[2] > eb[ax]) ax = 2;
            std::stable_sort(kids, kids_mid, [&](int32_t x, int32_t y) {
                const Box& bx = B.n2[x].box; const Box& by = B.n2[y].box;
                return (double)bx.lo[ax] + bx.hi[ax] < (double)by.lo[ax] + by.hi[ax];
            });
        }
        n.child_base = (uint32_t)out.nodes.size();
        n.tri_base   = (uint32_t)out.tris.size();
        int n_internal = 0;
        uint32_t tri_off = 0;
        for (int i = 0; i < nk; i++)
        {
            const Bin2& c = B.n2[kids[i]];
            for (int a = 0; a < 3; a++)
            {
                double s  = std::ldexp(1.0, (int)eb[a] - 127);
                double o  = (a == 0 ? n.ox : (a == 1 ? n.oy : n.oz));
                double ql = std::floor(((double)c.box.lo[a] - o) / s);
                double qh = std::ceil(((double)c.box.hi[a] - o) / s);
                if (ql < 0) ql = 0;
                if (ql > 255) ql = 255;
                if (qh > 255) qh = 255;
                if (qh < ql) qh = ql;
                n.qlo[a][i] = (uint8_t)ql;
                n.qhi[a][i] = (uint8_t)qh;
            }
            if (i < n_int_kids)
            {
                n.meta[i] = (uint8_t)(0x10 | n_internal);
                n_internal++;
            }
            else
            {
                n.meta[i] = (uint8_t)((c.count << 5) | tri_off);
                for (int t = 0; t < c.count; t++)
                {
                    int32_t      prim = B.prim[B.idx[c.first + t]];
                    const float* p    = positions + (size_t)prim * 9;
                    TriGPU       tg;
                    std::memset(&tg, 0, sizeof(tg));
                    std::memcpy(tg.v0, p, 12);
                    std::memcpy(tg.v1, p + 3, 12);
                    std::memcpy(tg.v2, p + 6, 12);
                    tg.prim = (uint32_t)prim;
                    out.tris.push_back(tg);
                }
                tri_off += (uint32_t)c.count;
            }
        }
        n.counts = (uint8_t)(n_internal | (nk << 4));
        // reserve the internal children contiguously, then enqueue them in slot order
        size_t base = out.nodes.size();
        out.nodes.resize(base + n_internal);
        out.nodes[pd.n8] = n;
        int slot = 0;
        for (int i = 0; i < nk; i++)
            if (i < n_int_kids) q.push({ kids[i], (int32_t)(base + slot++), pd.depth + 1 });
    }
    if (getenv("HR_BVH_STATS"))
    {
        long hist[9] = { 0 }, ihist[9] = { 0 };
        for (const Node8& n : out.nodes) { hist[n.counts >> 4]++; ihist[n.counts & 15]++; }
        fprintf(stderr, "bvh8: %zu nodes, %zu tri refs, depth %d; children/node histogram:", out.nodes.size(), out.tris.size(), out.max_depth);
        for (int i = 0; i <= 8; i++) fprintf(stderr, " %d:%ld", i, hist[i]);
        fprintf(stderr, "; internal children/node:");
        for (int i = 0; i <= 8; i++) fprintf(stderr, " %d:%ld", i, ihist[i]);
        fprintf(stderr, "\n");
    }
}

} // namespace hr
