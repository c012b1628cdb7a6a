// Host-side builder of the compressed 8-wide BVH (see bvh.h).  Replaces the driver's acceleration-structure build behind
// dw::RayTracedScene (reference: main.cpp:74 build_tlas, common.cpp:355-521 initialize_for_ray_tracing).  Steps:
//   1. binary tree over triangle REFERENCES down to single references: binned-SAH object splits against chopped-binning
//      SPATIAL splits (Stich, Friedrich, Dietrich 2009, "Spatial splits in bounding volume hierarchies") — a triangle that
//      straddles the chosen plane is referenced from both sides with its box clipped to each side (or kept whole on one side when
//      that is cheaper: reference unsplitting), within a duplication budget;
//   2. insertion-based optimisation of that tree (Bittner, Hapala, Havran 2013): subtrees are removed and re-inserted at the
//      position that minimises the surface-area cost, largest nodes first;
//   3. SAH-optimal collapse to 8-wide nodes with leaf children of <= 4 triangles (dynamic programme, Ylitie et al. 2017);
//   4. breadth-first layout with contiguous children / leaf triangles, 8-bit conservative quantisation.
// Every reference points at its ORIGINAL triangle and the leaves store the unmodified vertices, so the ray/triangle test and its
// results — any-hit: a function of the geometry; closest hit: of (t, original index) — do not depend on any of this: only the boxes
// get tighter.  Deterministic: no RNG, and the subtrees that are built on several threads (step 1) are merged in left-right order, so
// the tree does not depend on the number of threads (HR_BVH_THREADS) or on scheduling.
#include "bvh.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <future>
#include <queue>
#include <system_error>
#include <thread>

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
    void clip_to(const Box& b)
    {
        for (int a = 0; a < 3; a++)
        {
            if (b.lo[a] > lo[a]) lo[a] = b.lo[a];
            if (b.hi[a] < hi[a]) hi[a] = b.hi[a];
        }
    }
    bool valid() const { return lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]; }
    bool same(const Box& b) const { return std::memcmp(lo, b.lo, 12) == 0 && std::memcmp(hi, b.hi, 12) == 0; }
    double half_area() const
    {
        double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
        if (x < 0 || y < 0 || z < 0) return 0.0;
        return x * y + y * z + z * x;
    }
};
inline Box merged(const Box& a, const Box& b) { Box r = a; r.add(b); return r; }

struct Bin2
{
    Box     box;
    int32_t a = -1, b = -1; // children, or -1 for leaf
    int32_t parent = -1;
    int32_t first = 0, count = 0;   // references of the subtree: Builder::leaves[first .. first + count)
};

constexpr int kBins        = 64;   // 32 -> 64: nodes per shadow ray 6.38 -> 6.30, frame -1.3% (8 bins: 7.25, +5%)
constexpr int kSpatialBins = 32;
constexpr int kMaxLeaf     = 4;    // triangles per leaf CHILD of an 8-wide node (count field of the meta byte, 8 x 4 = 32-bit mask)

// SAH bin of a coordinate.  `k` = bins / extent overflows to +inf when the extent is subnormal and the product is then inf or
// NaN (0 * inf): compare in a way that sends both to a valid bin instead of converting them to int (undefined).
static inline int bin_of(float c, float lo, float k, int bins = kBins)
{
    const float f = (c - lo) * k;
    if (!(f > 0.0f)) return 0;               // also NaN
    if (f >= (float)(bins - 1)) return bins - 1;
    return (int)f;
}

// A triangle reference: the box of (triangle ∩ the region its ancestors' spatial splits left it), and the original triangle.
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

// outward-rounded float bounds of a polygon with double coordinates, clamped to the piece it was cut from
inline Box poly_box(const Poly& p, const Box& parent)
{
    Box b;
    for (int i = 0; i < p.n; i++)
        for (int k = 0; k < 3; k++)
        {
            float lo = (float)p.v[i][k], hi = lo;
            if ((double)lo > p.v[i][k]) lo = std::nextafter(lo, -FLT_MAX);
            if ((double)hi < p.v[i][k]) hi = std::nextafter(hi, FLT_MAX);
            if (lo < b.lo[k]) b.lo[k] = lo;
            if (hi > b.hi[k]) b.hi[k] = hi;
        }
    b.clip_to(parent);
    return b;
}

inline void tri_poly(const float* positions, int32_t prim, Poly& poly)
{
    const float* p = positions + (size_t)prim * 9;
    poly.n = 3;
    for (int v = 0; v < 3; v++)
        for (int k = 0; k < 3; k++) poly.v[v][k] = p[v * 3 + k];
}

// ---- early split clipping (HR_BVH_SPLIT=<fraction of the scene diagonal>, developer switch; the spatial splits of the SAH build
// below do the same job adaptively): every triangle whose box is longer than `limit` is cut in halves before the build.
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

struct Builder
{
    const float*      pos;
    std::vector<Bin2> n2;
    std::vector<Ref>  leaves;         // references in leaf order
    int               bvh2_leaf = 1;  // binary-tree leaf size (1 for the optimal collapse: it forms the leaves)
    bool              spatial   = true;
    double            alpha     = 1e-3;   // spatial splits are tried when the object split's children overlap by more than alpha x root area
    double            root_area = 0.0;
    long              n_spatial = 0, n_unsplit = 0;

    // SAH splits down to binary depth kSahDepth, object-median splits below it: a median split halves the count, so the
    // binary tree is never deeper than kSahDepth + ceil(log2 n_refs) <= 36 + 26 = 62 levels (hr_scene_create refuses 2^26
    // references).  The 8-wide collapse only removes levels, and the traversal keeps ONE stack entry per level (traverse.h:
    // walk_expand), so HR_STACK_ENTRIES + HR_SPILL_ENTRIES = 64 entries always suffice — also for adversarial input (a chain of
    // slivers of geometrically growing size makes the SAH peel one triangle per level, n levels deep;
    // tests/test_gpu_trace.py::test_degenerate_sliver_chain).
    static constexpr int kSahDepth = 36;
    int sah_depth = kSahDepth;   // HR_BVH_SAH_DEPTH lowers it (developer switch: exercises the median fallback in tests)

    static inline float centre(const Ref& r, int ax) { return 0.5f * (r.box.lo[ax] + r.box.hi[ax]); }

    // Builds the subtree over A[first .. first + count).  Object and median splits partition that range in place; a spatial split
    // writes its two (longer) reference lists into vectors of their own and recurses into those.
    // `budget`: references the spatial splits of THIS subtree may still add (split between the children in proportion to their
    // reference counts, so the tree does not depend on the order in which subtrees are built).
    // Subtrees of more than kParallelMin references are built by two threads — each child into a Builder of its own, merged back in
    // left-right order with shifted indices — as long as the thread allowance (HR_BVH_THREADS, default min(hardware, 16)) lasts: the
    // node numbering, the leaf order and therefore the whole BVH are those of the single-threaded build.
    static constexpr size_t kParallelMin = 16384;
    std::atomic<int>* threads_left = nullptr;

    void adopt(Builder& c, int32_t parent_node)
    {
        // append the nodes / leaves of a child context (root = its node 0) behind ours
        const int32_t no = (int32_t)n2.size(), lo = (int32_t)leaves.size();
        for (Bin2 n : c.n2)
        {
            if (n.a >= 0) { n.a += no; n.b += no; }
            else n.first += lo;
            n.parent = n.parent >= 0 ? n.parent + no : parent_node;
            n2.push_back(n);
        }
        leaves.insert(leaves.end(), c.leaves.begin(), c.leaves.end());
        n_spatial += c.n_spatial; n_unsplit += c.n_unsplit;
    }
    Builder child_context() const
    {
        Builder c;
        c.pos = pos; c.bvh2_leaf = bvh2_leaf; c.spatial = spatial; c.alpha = alpha; c.root_area = root_area; c.sah_depth = sah_depth;
        c.threads_left = threads_left; c.max_area_fraction = max_area_fraction;
        return c;
    }
    // builds the two children of node `me` over (AL, fl, cl) and (AR, fr, cr); returns their node indices
    void build_children(int32_t me, std::vector<Ref>& AL, size_t fl, size_t cl, std::vector<Ref>& AR, size_t fr, size_t cr, int depth, long budget, int32_t& l, int32_t& r)
    {
        const long bl = (cl + cr) ? (long)((double)budget * (double)cl / (double)(cl + cr)) : 0, br = budget - bl;
        bool parallel = false;
        if (threads_left && cl >= kParallelMin && cr >= kParallelMin)
        {
            int have = threads_left->load();
            while (have > 0 && !threads_left->compare_exchange_weak(have, have - 1)) {}
            parallel = have > 0;
        }
        if (!parallel)
        {
            l = build(AL, fl, cl, depth + 1, me, bl);
            r = build(AR, fr, cr, depth + 1, me, br);
            return;
        }
        Builder lc = child_context(), rc = child_context();
        std::future<void> fut;
        try
        {
            fut = std::async(std::launch::async, [&] { lc.build(AL, fl, cl, depth + 1, -1, bl); });
        }
        catch (const std::system_error&)   // no thread to be had: build the left subtree here (same result)
        {
            lc.build(AL, fl, cl, depth + 1, -1, bl);
        }
        rc.build(AR, fr, cr, depth + 1, -1, br);
        if (fut.valid()) fut.get();
        threads_left->fetch_add(1);
        l = (int32_t)n2.size();
        adopt(lc, me);
        r = (int32_t)n2.size();
        adopt(rc, me);
    }

    int32_t build(std::vector<Ref>& A, size_t first, size_t count, int depth, int32_t parent, long budget)
    {
        const int32_t me = (int32_t)n2.size();
        n2.emplace_back();
        Ref* const refs = A.data() + first;
        Box nb, cb;
        for (size_t i = 0; i < count; i++)
        {
            nb.add(refs[i].box);
            const float c[3] = { centre(refs[i], 0), centre(refs[i], 1), centre(refs[i], 2) };
            cb.add(c);
        }
        n2[me].box    = nb;
        n2[me].parent = parent;
        n2[me].count  = (int32_t)count;
        if ((int)count <= bvh2_leaf)
        {
            n2[me].first = (int32_t)leaves.size();
            leaves.insert(leaves.end(), refs, refs + count);
            return me;
        }
        size_t mid = 0;   // object / median split: A[first .. first + mid) | A[first + mid .. first + count)
        if (depth >= sah_depth) mid = median_split(refs, count, cb);
        else
        {
            // ---- object split: binned SAH over the reference centres, all three axes
            double best = DBL_MAX;
            int    bax = -1, bsplit = 0;
            Box    obj_l, obj_r;
            for (int ax = 0; ax < 3; ax++)
            {
                const float ext = cb.hi[ax] - cb.lo[ax];
                if (!(ext > 0.0f)) continue;
                Box         bbox[kBins];
                int         bcnt[kBins] = { 0 };
                const float k           = (float)kBins / ext;
                for (size_t i = 0; i < count; i++)
                {
                    const int b = bin_of(centre(refs[i], ax), cb.lo[ax], k);
                    bbox[b].add(refs[i].box);
                    bcnt[b]++;
                }
                Box    racc[kBins];
                int    rcnt[kBins];
                Box    acc;
                int    c = 0;
                for (int b = kBins - 1; b >= 1; b--)
                {
                    acc.add(bbox[b]);
                    c += bcnt[b];
                    racc[b] = acc;
                    rcnt[b] = c;
                }
                Box lacc;
                int lc = 0;
                for (int b = 0; b < kBins - 1; b++)
                {
                    lacc.add(bbox[b]);
                    lc += bcnt[b];
                    if (lc == 0 || rcnt[b + 1] == 0) continue;
                    const double cost = lacc.half_area() * lc + racc[b + 1].half_area() * rcnt[b + 1];
                    if (cost < best) { best = cost; bax = ax; bsplit = b; obj_l = lacc; obj_r = racc[b + 1]; }
                }
            }
            // ---- spatial split: only where the object split leaves its children overlapping
            if (spatial && budget > 0 && bax >= 0)
            {
                Box ov = obj_l;
                ov.clip_to(obj_r);
                if (ov.valid() && ov.half_area() > alpha * root_area)
                {
                    std::vector<Ref> L, R;
                    long dup = 0;
                    if (spatial_split(refs, count, nb, best, L, R, budget, dup))
                    {
                        int32_t l, r;
                        build_children(me, L, 0, L.size(), R, 0, R.size(), depth, budget - dup, l, r);
                        n2[me].a = l;
                        n2[me].b = r;
                        return me;
                    }
                }
            }
            if (bax < 0) mid = median_split(refs, count, cb);
            else
            {
                const float k = (float)kBins / (cb.hi[bax] - cb.lo[bax]), lo = cb.lo[bax];
                mid = (size_t)(std::partition(refs, refs + count, [&](const Ref& r) { return bin_of(centre(r, bax), lo, k) <= bsplit; }) - refs);
                if (mid == 0 || mid == count) mid = median_split(refs, count, cb);
            }
        }
        int32_t l, r;
        build_children(me, A, first, mid, A, first + mid, count - mid, depth, budget, l, r);
        n2[me].a = l;
        n2[me].b = r;
        return me;
    }

    // halves the references by the centre along the widest axis of the centre bounds (ties: original index, then box)
    size_t median_split(Ref* refs, size_t count, const Box& cb)
    {
        int ax = 0;
        for (int k = 1; k < 3; k++)
            if (cb.hi[k] - cb.lo[k] > cb.hi[ax] - cb.lo[ax]) ax = k;
        const size_t mid = count / 2;
        std::nth_element(refs, refs + mid, refs + count, [&](const Ref& a, const Ref& b) {
            const float ca = centre(a, ax), cb_ = centre(b, ax);
            if (ca != cb_) return ca < cb_;
            if (a.prim != b.prim) return a.prim < b.prim;
            return a.box.lo[ax] < b.box.lo[ax];
        });
        return mid;
    }

    // Chopped binning over the node's box on every axis; performs the split (filling L, R) and returns true when the best plane
    // beats the object split's cost `object_cost`.
    bool spatial_split(const Ref* refs_, size_t count, const Box& nb, double object_cost, std::vector<Ref>& L, std::vector<Ref>& R, long budget, long& dup_out)
    {
        struct Range { const Ref *b, *e; const Ref* begin() const { return b; } const Ref* end() const { return e; } } refs { refs_, refs_ + count };
        double best = object_cost;
        int    bax = -1, bsplit = 0;
        for (int ax = 0; ax < 3; ax++)
        {
            const float ext = nb.hi[ax] - nb.lo[ax];
            if (!(ext > 0.0f)) continue;
            const float  k = (float)kSpatialBins / ext;
            const double w = (double)ext / kSpatialBins;
            Box bbox[kSpatialBins];
            int enter[kSpatialBins] = { 0 }, leave[kSpatialBins] = { 0 };
            for (const Ref& r : refs)
            {
                const int b0 = bin_of(r.box.lo[ax], nb.lo[ax], k, kSpatialBins), b1 = bin_of(r.box.hi[ax], nb.lo[ax], k, kSpatialBins);
                enter[b0]++; leave[b1]++;
                if (b0 == b1) { bbox[b0].add(r.box); continue; }
                Poly rest, piece, next;
                tri_poly(pos, r.prim, rest);
                for (int j = b0; j <= b1 && rest.n >= 3; j++)
                {
                    if (j < b1)
                    {
                        const double plane = (double)nb.lo[ax] + w * (j + 1);
                        clip_poly(rest, ax, plane, true, piece);
                        clip_poly(rest, ax, plane, false, next);
                    }
                    else { piece = rest; next.n = 0; }
                    if (piece.n >= 3)
                    {
                        const Box pb = poly_box(piece, r.box);
                        if (pb.valid()) bbox[j].add(pb);
                    }
                    rest = next;
                }
            }
            Box racc[kSpatialBins];
            int rcnt[kSpatialBins];
            Box acc;
            int c = 0;
            for (int b = kSpatialBins - 1; b >= 1; b--)
            {
                acc.add(bbox[b]);
                c += leave[b];
                racc[b] = acc;
                rcnt[b] = c;
            }
            Box lacc;
            int lc = 0;
            for (int b = 0; b < kSpatialBins - 1; b++)
            {
                lacc.add(bbox[b]);
                lc += enter[b];
                if (lc == 0 || rcnt[b + 1] == 0) continue;
                const double cost = lacc.half_area() * lc + racc[b + 1].half_area() * rcnt[b + 1];
                if (cost < best) { best = cost; bax = ax; bsplit = b; }
            }
        }
        if (bax < 0) return false;
        const int    ax    = bax;
        const double w     = ((double)nb.hi[ax] - nb.lo[ax]) / kSpatialBins;
        const double plane = (double)nb.lo[ax] + w * (bsplit + 1);
        struct Straddler { Ref whole, l, r; };
        std::vector<Straddler> st;
        Box lb, rb;
        for (const Ref& r : refs)
        {
            if ((double)r.box.hi[ax] <= plane) { L.push_back(r); lb.add(r.box); }
            else if ((double)r.box.lo[ax] >= plane) { R.push_back(r); rb.add(r.box); }
            else
            {
                Poly tri, pl, pr;
                tri_poly(pos, r.prim, tri);
                clip_poly(tri, ax, plane, true, pl);
                clip_poly(tri, ax, plane, false, pr);
                Straddler s;
                s.whole = r;
                s.l = Ref { poly_box(pl, r.box), r.prim };
                s.r = Ref { poly_box(pr, r.box), r.prim };
                const bool lv = pl.n >= 3 && s.l.box.valid(), rv = pr.n >= 3 && s.r.box.valid();
                if (lv && rv) st.push_back(s);
                else if (lv) { L.push_back(r); lb.add(r.box); }
                else { R.push_back(r); rb.add(r.box); }
            }
        }
        // reference unsplitting: a straddler goes to one side whole when that is cheaper than referencing it from both
        long nl = (long)L.size(), nr = (long)R.size(), dup = 0;
        dup_out = 0;
        for (const Straddler& s : st)
        {
            const Box    lub = merged(lb, s.whole.box), rub = merged(rb, s.whole.box), ldb = merged(lb, s.l.box), rdb = merged(rb, s.r.box);
            const double c_split = ldb.half_area() * (nl + 1) + rdb.half_area() * (nr + 1);
            const double c_left  = lub.half_area() * (nl + 1) + rb.half_area() * nr;
            const double c_right = lb.half_area() * nl + rub.half_area() * (nr + 1);
            if (c_split < c_left && c_split < c_right && dup < budget)
            {
                L.push_back(s.l); R.push_back(s.r); lb = ldb; rb = rdb; nl++; nr++; dup++;
            }
            else if (c_left <= c_right) { L.push_back(s.whole); lb = lub; nl++; n_unsplit++; }
            else { R.push_back(s.whole); rb = rub; nr++; n_unsplit++; }
        }
        if (L.empty() || R.empty())
        {
            L.clear(); R.clear();
            return false;
        }
        dup_out = dup;
        n_spatial++;
        return true;
    }

    // ---- insertion-based optimisation (Bittner et al. 2013) ----------------------------------------------------------------
    // A node v is cut out together with its parent p (v's sibling takes p's place), the tree above is refitted, and v is put back
    // beside the node x that minimises  area(x ∪ v) + Σ over x's ancestors of the growth of their area  (branch and bound from the
    // root, cheapest induced cost first); p is re-used as the common parent of x and v.  The old position is among the candidates,
    // so the SAH cost never rises.  Per pass the `fraction` of the nodes with the largest area are processed, largest first.
    int32_t root = 0;
    void refit_up(int32_t i)
    {
        while (i >= 0)
        {
            const Box b = merged(n2[n2[i].a].box, n2[n2[i].b].box);
            if (b.same(n2[i].box)) break;
            n2[i].box = b;
            i = n2[i].parent;
        }
    }
    int32_t find_best(const Box& vb)
    {
        struct Item { double induced; int32_t node; bool operator<(const Item& o) const { return induced > o.induced; } };
        static thread_local std::vector<Item> heap;
        heap.clear();
        const double va = vb.half_area();
        double  best_cost = DBL_MAX;
        int32_t best = root;
        heap.push_back({ 0.0, root });
        while (!heap.empty())
        {
            std::pop_heap(heap.begin(), heap.end());
            const Item it = heap.back();
            heap.pop_back();
            if (it.induced + va >= best_cost) break;
            const Bin2&  x      = n2[it.node];
            const double direct = merged(x.box, vb).half_area(), total = it.induced + direct;
            if (total < best_cost) { best_cost = total; best = it.node; }
            const double child_induced = total - x.box.half_area();
            if (x.a >= 0 && child_induced + va < best_cost)
            {
                heap.push_back({ child_induced, x.a }); std::push_heap(heap.begin(), heap.end());
                heap.push_back({ child_induced, x.b }); std::push_heap(heap.begin(), heap.end());
            }
        }
        return best;
    }
    bool reinsert(int32_t v)
    {
        const int32_t p = n2[v].parent;
        if (p < 0 || n2[p].parent < 0) return false;   // the root and its children stay
        const int32_t g = n2[p].parent, s = n2[p].a == v ? n2[p].b : n2[p].a;
        (n2[g].a == p ? n2[g].a : n2[g].b) = s;
        n2[s].parent = g;
        refit_up(g);
        const int32_t x = find_best(n2[v].box), px = n2[x].parent;
        if (px >= 0) (n2[px].a == x ? n2[px].a : n2[px].b) = p;
        else root = p;
        n2[p].parent = px;
        n2[p].a = x; n2[p].b = v;
        n2[x].parent = p; n2[v].parent = p;
        n2[p].box = merged(n2[x].box, n2[v].box);
        refit_up(px);
        return x != s;
    }
    int depth_of_tree() const
    {
        std::vector<std::pair<int32_t, int>> st;
        st.push_back({ root, 1 });
        int d = 0;
        while (!st.empty())
        {
            const auto [n, dn] = st.back();
            st.pop_back();
            if (dn > d) d = dn;
            if (n2[n].a >= 0) { st.push_back({ n2[n].a, dn + 1 }); st.push_back({ n2[n].b, dn + 1 }); }
        }
        return d;
    }
    double sah() const
    {
        double c = 0;
        for (const Bin2& n : n2) c += n.box.half_area();
        return c / (root_area > 0 ? root_area : 1.0);
    }
    // Nodes larger than this share of the root's area are left where the (spatial-split) build put them: re-inserting the top of the
    // tree lowers the SAH cost further but lengthens the slowest rays of the shadow pass, whose launch is bound by its tail (measured
    // at 1080p, standard tier, trace in us: no limit 98.5 / 370 / 265 / 195 for shadows / AO / DDGI / reflections, 0.05: 89.7 / 372 /
    // 260 / 196, no reinsertion at all 91 / 369 / 270 / 197; hard tier 163 / 666 / 432 / 299 against 160 / 671 / 425 / 309).
    double max_area_fraction = 0.05;   // developer switch HR_BVH_REINSERT_MAX_AREA
    long optimise(int passes, double fraction)
    {
        long moved = 0;
        std::vector<std::pair<double, int32_t>> cand;
        for (int pass = 0; pass < passes; pass++)
        {
            cand.clear();
            for (int32_t i = 0; i < (int32_t)n2.size(); i++)
                if (n2[i].parent >= 0 && n2[n2[i].parent].parent >= 0 && n2[i].box.half_area() <= max_area_fraction * root_area) cand.push_back({ n2[i].box.half_area(), i });
            const size_t take = (size_t)((double)cand.size() * fraction);
            if (take == 0) break;
            std::partial_sort(cand.begin(), cand.begin() + take, cand.end(), [](const auto& a, const auto& b) { return a.first > b.first || (a.first == b.first && a.second < b.second); });
            long m = 0;
            for (size_t i = 0; i < take; i++) m += reinsert(cand[i].second) ? 1 : 0;
            moved += m;
            if (m == 0) break;
        }
        return moved;
    }

    // After the optimisation the subtrees are no longer contiguous in `leaves`: re-emit the references in depth-first order, set
    // first / count of every node, pad the boxes and return the nodes children-before-parents.
    std::vector<int32_t> finalise(float pad)
    {
        std::vector<int32_t> post;
        post.reserve(n2.size());
        std::vector<Ref> out;
        out.reserve(leaves.size());
        std::vector<std::pair<int32_t, bool>> st;
        st.push_back({ root, false });
        while (!st.empty())
        {
            const auto [n, seen] = st.back();
            st.pop_back();
            Bin2& c = n2[n];
            if (c.a < 0)
            {
                const int32_t first = (int32_t)out.size();
                Box b;
                for (int32_t t = 0; t < c.count; t++) { out.push_back(leaves[c.first + t]); b.add(leaves[c.first + t].box); }
                c.first = first;
                for (int a = 0; a < 3; a++) { b.lo[a] -= pad; b.hi[a] += pad; }
                c.box = b;
                post.push_back(n);
            }
            else if (!seen)
            {
                st.push_back({ n, true });
                st.push_back({ c.b, false });
                st.push_back({ c.a, false });
            }
            else
            {
                c.first = n2[c.a].first;
                c.count = n2[c.a].count + n2[c.b].count;
                c.box   = merged(n2[c.a].box, n2[c.b].box);
                post.push_back(n);
            }
        }
        leaves.swap(out);
        return post;
    }
};

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
    out.child_boxes.clear();
    out.tris.clear();
    out.max_depth = 0;
    Builder B;
    B.pos = positions;
    Box all;
    for (int i = 0; i < n_tris; i++)
    {
        const float* p = positions + (size_t)i * 9;
        bool finite = true;
        for (int k = 0; k < 9; k++) finite = finite && std::isfinite(p[k]);
        if (!finite) continue;   // see below: such a triangle is never hit and gets no reference
        all.add(p); all.add(p + 3); all.add(p + 6);
    }
    if (n_tris == 0 || !all.valid())
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
    if (n_tris == 0)
    {
        Node8 n;
        std::memset(&n, 0, sizeof(n));
        n.ex = n.ey = n.ez = 1;
        out.nodes.push_back(n);
        return;
    }
    double split_fraction = 0.0;
    if (const char* e = getenv("HR_BVH_SPLIT")) split_fraction = atof(e);
    const float limit = split_fraction > 0.0 ? (float)(diag * split_fraction) : FLT_MAX;
    std::vector<Ref> refs;
    refs.reserve((size_t)n_tris + n_tris / 4);
    for (int i = 0; i < n_tris; i++)
    {
        const float* p = positions + (size_t)i * 9;
        // A triangle with a NaN or infinite coordinate can never be hit (the watertight test's edge functions come out NaN / of mixed
        // sign and its interval test fails on NaN), so it gets no reference at all: non-finite boxes would poison the SAH areas of every
        // node above them (and made the reinsertion search quadratic: 15 s for 3000 triangles, two of them non-finite).
        bool finite = true;
        for (int k = 0; k < 9; k++) finite = finite && std::isfinite(p[k]);
        if (!finite) continue;
        Box tb;
        tb.add(p); tb.add(p + 3); tb.add(p + 6);
        Poly poly;
        tri_poly(positions, i, poly);
        split_refs(poly, tb, i, limit, 0, refs);
    }
    if (refs.empty())
    {
        Node8 n;
        std::memset(&n, 0, sizeof(n));
        n.ex = n.ey = n.ez = 1;
        out.nodes.push_back(n);
        return;
    }
    // developer switches (A/B of the build steps; tests/test_bvh_host.py, tools/bvh_eval.cpp)
    B.bvh2_leaf = getenv("HR_BVH_GREEDY") ? kMaxLeaf : 1;
    if (const char* e = getenv("HR_BVH_SAH_DEPTH")) { const int v = atoi(e); if (v >= 0 && v < Builder::kSahDepth) B.sah_depth = v; }
    if (const char* e = getenv("HR_BVH_SBVH")) B.spatial = atoi(e) != 0;
    if (const char* e = getenv("HR_BVH_ALPHA")) B.alpha = atof(e);
    double budget_fraction = 0.3;
    if (const char* e = getenv("HR_BVH_BUDGET")) budget_fraction = atof(e);
    int    passes = 2;
    double fraction = 0.1;
    if (const char* e = getenv("HR_BVH_REINSERT")) passes = atoi(e);
    if (const char* e = getenv("HR_BVH_REINSERT_FRACTION")) fraction = atof(e);
    if (const char* e = getenv("HR_BVH_REINSERT_MAX_AREA")) B.max_area_fraction = atof(e);
    const long budget = (long)(budget_fraction * (double)refs.size());
    int n_threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* e = getenv("HR_BVH_THREADS")) n_threads = std::max(1, atoi(e));
    std::atomic<int> threads_left(n_threads - 1);
    B.threads_left = n_threads > 1 ? &threads_left : nullptr;
    B.root_area = all.half_area();
    B.n2.reserve(refs.size() * 2 + refs.size() / 2);
    B.leaves.reserve(refs.size() + refs.size() / 3);
    const size_t n_input_refs = refs.size();
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
    B.root = B.build(refs, 0, refs.size(), 0, -1, budget);
    const double t_built = since();
    std::vector<Ref>().swap(refs);
    const double sah_built = getenv("HR_BVH_STATS") ? B.sah() : 0.0;
    long moved = 0;
    if (passes > 0 && B.bvh2_leaf == 1 && B.n2.size() > 8)
    {
        // the optimisation may deepen the tree; the traversal stack bounds the depth (see kSahDepth): keep the built tree if it does
        const int depth_limit = kMaxTraversalDepth - 2;
        const std::vector<Bin2> backup = B.n2;
        const int32_t           root_backup = B.root;
        moved = B.optimise(passes, fraction);
        if (B.depth_of_tree() > depth_limit) { B.n2 = backup; B.root = root_backup; moved = 0; }
    }
    const double t_opt = since();
    if (getenv("HR_BVH_STATS"))
        fprintf(stderr, "bvh2: build %.2f s, optimise %.2f s\n", t_built, t_opt - t_built);
    if (getenv("HR_BVH_STATS"))
        fprintf(stderr, "bvh2: %d triangles, %zu input refs, %zu refs (%ld spatial splits, %ld unsplit), %zu nodes, SAH %.3f -> %.3f (%ld reinsertions), depth %d\n", n_tris,
                n_input_refs, B.leaves.size(), B.n_spatial, B.n_unsplit, B.n2.size(), sah_built, B.sah(), moved, B.depth_of_tree());
    const std::vector<int32_t> post = B.finalise(out.pad);
    const int32_t root2 = B.root;
    const int     n_refs = (int)B.leaves.size();

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
        double C_node = 1.0, C_prim = 0.35;
        if (const char* e = getenv("HR_BVH_CPRIM")) C_prim = atof(e);   // developer switch (tools/bvh_eval.cpp sweeps it)
        cost.assign(nn * 8, 0.0); best_k.assign(nn * 8, 0); use_split.assign(nn * 8, 0); as_leaf.assign(nn, 0);
        for (const int32_t r_ : post)   // children before parents
        {
            const size_t r = (size_t)r_;
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
        int sort_axis = 0;
        {
            // Internal children are ordered along the node's longest axis (largest scale exponent, first one on ties; the centre
            // spread of the children was tried as the criterion: no better), stored in the spare bits of slot 0's meta byte (bvh.h),
            // so that a ray can walk them in order of distance: ascending slots for a positive direction component, descending
            // for a negative one — closest-hit queries near to far, any-hit queries far to near (traverse.h).
            int ax = 0;
            if (eb[1] > eb[0]) ax = 1;
            if (eb[2] > eb[ax]) ax = 2;
            sort_axis = ax;
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
                n.meta[i] = (uint8_t)(0x10 | (i == 0 ? sort_axis : 0));
                n_internal++;
            }
            else
            {
                // a leaf may hold two references to one triangle (pieces that a spatial split separated and the collapse
                // joined again): the triangle is stored once
                int32_t prims[kMaxLeaf];
                int     np = 0;
                for (int t = 0; t < c.count; t++)
                {
                    const int32_t prim = B.leaves[c.first + t].prim;
                    if (std::find(prims, prims + np, prim) == prims + np) prims[np++] = prim;
                }
                n.meta[i] = (uint8_t)((np << 5) | tri_off);
                for (int t = 0; t < np; t++)
                {
                    const int32_t prim = prims[t];
                    const float* p    = positions + (size_t)prim * 9;
                    TriGPU       tg;
                    std::memset(&tg, 0, sizeof(tg));
                    std::memcpy(tg.v0, p, 12);
                    std::memcpy(tg.v1, p + 3, 12);
                    std::memcpy(tg.v2, p + 6, 12);
                    tg.prim = (uint32_t)prim;
                    out.tris.push_back(tg);
                }
                tri_off += (uint32_t)np;
            }
        }
        n.counts = (uint8_t)(n_internal | (nk << 4));
        // reserve the internal children contiguously, then enqueue them in slot order
        size_t base = out.nodes.size();
        out.nodes.resize(base + n_internal);
        out.nodes[pd.n8] = n;
        if (out.want_child_boxes)
        {
            if (out.child_boxes.size() < out.nodes.size() * 48) out.child_boxes.resize(out.nodes.size() * 48, 0.0f);
            for (int i = 0; i < nk; i++)
                for (int a = 0; a < 3; a++)
                {
                    out.child_boxes[((size_t)pd.n8 * 8 + i) * 6 + a]     = B.n2[kids[i]].box.lo[a];
                    out.child_boxes[((size_t)pd.n8 * 8 + i) * 6 + 3 + a] = B.n2[kids[i]].box.hi[a];
                }
        }
        int slot = 0;
        for (int i = 0; i < nk; i++)
            if (i < n_int_kids) q.push({ kids[i], (int32_t)(base + slot++), pd.depth + 1 });
    }
    out.n_refs = (int)out.tris.size();
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


// Host-side self-check of the property every query relies on: a ray that meets triangle T at point p must find T in a leaf it
// reaches through boxes that all contain p.  With spatial splits a triangle is referenced from several leaves, each covering the
// part of it inside that leaf's (clipped) box — the pieces together must cover the triangle.  For `samples` points of every
// triangle (corners, edge midpoints, centroid, then hashed barycentric points) the tree is descended through the dequantised
// child boxes that contain the point; returns how many (triangle, point) pairs reach no leaf holding the triangle.
int64_t check_bvh8_coverage(const float* positions, int n_tris, const BuiltBVH& b, int samples)
{
    if (n_tris == 0) return 0;
    int64_t bad = 0;
    std::vector<uint32_t> stack;
    for (int t = 0; t < n_tris; t++)
    {
        const float* p = positions + (size_t)t * 9;
        for (int s = 0; s < samples; s++)
        {
            double w[3];
            if (s < 3) { w[0] = s == 0; w[1] = s == 1; w[2] = s == 2; }
            else if (s < 6) { w[0] = s == 5 ? 0.5 : (s == 3 ? 0.5 : 0.0); w[1] = s == 3 ? 0.5 : (s == 4 ? 0.5 : 0.0); w[2] = 1.0 - w[0] - w[1]; }
            else if (s == 6) { w[0] = w[1] = 1.0 / 3.0; w[2] = 1.0 - w[0] - w[1]; }
            else
            {
                uint32_t h = (uint32_t)t * 2654435761u + (uint32_t)s * 40503u;
                h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16;
                double u = (h & 0xffff) / 65536.0, v = (h >> 16) / 65536.0;
                if (u + v > 1.0) { u = 1.0 - u; v = 1.0 - v; }
                w[0] = u; w[1] = v; w[2] = 1.0 - u - v;
            }
            double q[3];
            for (int k = 0; k < 3; k++) q[k] = w[0] * p[k] + w[1] * p[3 + k] + w[2] * p[6 + k];
            bool found = false;
            stack.clear();
            stack.push_back(0u);
            while (!stack.empty() && !found)
            {
                const Node8& n = b.nodes[stack.back()];
                stack.pop_back();
                const double sc[3] = { std::ldexp(1.0, (int)n.ex - 127), std::ldexp(1.0, (int)n.ey - 127), std::ldexp(1.0, (int)n.ez - 127) };
                const double o[3]  = { n.ox, n.oy, n.oz };
                const int    nk = n.counts >> 4, nin = n.counts & 15;
                for (int i = 0; i < nk && !found; i++)
                {
                    bool in = true;
                    for (int a = 0; a < 3; a++) in = in && q[a] >= o[a] + n.qlo[a][i] * sc[a] && q[a] <= o[a] + n.qhi[a][i] * sc[a];
                    if (!in) continue;
                    if (i < nin) stack.push_back(n.child_base + (uint32_t)i);
                    else
                        for (uint32_t k = 0; k < (uint32_t)(n.meta[i] >> 5); k++)
                            if (b.tris[n.tri_base + (n.meta[i] & 31u) + k].prim == (uint32_t)t) found = true;
                }
            }
            if (!found) bad++;
        }
    }
    return bad;
}

} // namespace hr
