// Instanced scenes with a per-frame acceleration-structure update — the MI355X replacement for dw::RayTracedScene's instance list
// (scene_descriptor_set.glsl:30-34 Instance, :102-131 fetch_hit_info / fetch_triangle through instance.mesh_idx, :150-160
// transform_vertex) and for main.cpp:74 build_tlas(cmd_buf), which the reference runs every frame.
//
// Layout (hr_api.h hr_scene_create_instanced): ONE world-space 8-wide BVH.  The topology of a mesh is built once, in object space, by the
// host builder (bvh_build.cpp); every instance gets a private copy of it (nodes + triangle references: HBM is 288 GB, a traversal
// that never leaves world space is worth the copies) under a top level over the instance roots:
//
//   nodes:  [ top level, root = 0 | the instance roots, grouped by eight under their top-level node | instance 0's other nodes | ... ]
//   tris:   [ instance 0's references | instance 1's references | ... ]       prim = global triangle index (instance order, then mesh order)
//
// hr_scene_update_instances (all on the caller's stream, no host synchronisation):
//   k_instances_transform   one thread per triangle: world vertices = model_matrix * (p, 1) (the hit records, the G-buffer synthesiser and
//                           the references read them), world vertex normals for the G-buffer synthesiser, per-instance bounds by atomic min / max
//   k_instances_gather      one thread per triangle reference: the 36 vertex bytes of its TriGPU
//   k_instances_refit[_top] one launch per WIDE tree level, deepest first, then one single-workgroup launch for all the narrow levels near the root
//                           (a barrier between levels): a thread per node recomputes the boxes of its eight children (internal
//                           children: the box their own thread stored one launch earlier; leaves: the bounds of their triangles CUT to the
//                           leaf's cell — the builder splits references spatially (SBVH), so a leaf of a mesh with long triangles bounds
//                           only the pieces inside its object-space cell: the cell goes through the instance's matrix (centre + |M| extent,
//                           in double, rounded outwards) and is intersected with the triangles' world bounds; without the cells the bench
//                           building traced 1.8x slower after a refit than as built), the node's origin / scale exponents, and
//                           requantises — the encoding rules of bvh_build.cpp, in double like there.
//                           Launch boundaries are the only ordering used: the per-XCD L2s are not coherent with each other, and a
//                           bottom-up walk with arrival counters inside one launch would need an agent-scope release per node.
// Only instances whose matrix CHANGED are touched: the kernels drop the triangles, references and nodes of the others on a 4-byte flag, so
// an update costs what moved (a static building under a few hundred movers: the movers), plus the top level.
// The topology (which triangles share a leaf, which nodes share a parent) never changes, so a query's ANSWER is the one a fresh build over
// the same world-space vertices would give — any-hit is a function of the geometry, closest hit is the smallest t with ties to the
// smallest triangle index — while the boxes stay as tight as the moved geometry allows: a rigidly moving instance keeps its own subtree.
#include "hr_internal.h"
#include "device_math.h"
#include <atomic>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>

using namespace hr;

namespace {

struct RefitArgs
{
    Node8*             nodes;
    const TriGPU*      tris;
    float*             node_box;     // [n_nodes][8]: lo xyz, pad, hi xyz, pad
    const uint32_t*    list;         // the nodes of this level
    const float*       cells;        // [n_nodes][8][6]: object-space cell of every LEAF slot (the builder's box of it), lo xyz hi xyz
    const int32_t*     node_inst;    // instance a node belongs to, -1: top level
    const InstanceRec* inst;
    const uint32_t*    dirty;        // per instance: matrix changed in this update
    int                count;
    float              pad;
};

__device__ uint8_t exponent_for_dev(float extent)
{
    // smallest e with extent <= 255 * 2^(e - 127) (bvh_build.cpp exponent_for; the answer is unique, so the starting guess is free)
    if (!(extent > 0.0f)) return 1;
    int e = (int)((__float_as_uint(extent) >> 23) & 0xffu) - 7;
    if (e < 1) e = 1;
    if (e > 254) e = 254;
    while (e > 1 && ldexp(255.0, e - 1 - 127) >= (double)extent) e--;
    while (e < 254 && ldexp(255.0, e - 127) < (double)extent) e++;
    return (uint8_t)e;
}
__device__ float round_down(double v) { float f = (float)v; return (double)f > v ? nextafterf(f, -INFINITY) : f; }
__device__ float round_up(double v) { float f = (float)v; return (double)f < v ? nextafterf(f, INFINITY) : f; }

__device__ void refit_node(const RefitArgs& a, const uint32_t ni)
{
    const int      in = a.node_inst[ni];
    if (in >= 0 && !a.dirty[in]) return;   // the instance did not move: its subtree stands
    Node8 n = a.nodes[ni];
    const int n_internal = n.counts & 15, n_children = n.counts >> 4;
    float clo[8][3], chi[8][3];
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int c = 0; c < n_children; c++)
    {
        if (c < n_internal)
        {
            const float* b = a.node_box + (size_t)(n.child_base + c) * 8;
            for (int k = 0; k < 3; k++) { clo[c][k] = b[k]; chi[c][k] = b[4 + k]; }
        }
        else
        {
            const uint32_t m = n.meta[c], cnt = m >> 5, off = m & 31u;
            float l[3] = { INFINITY, INFINITY, INFINITY }, h[3] = { -INFINITY, -INFINITY, -INFINITY };
            for (uint32_t t = 0; t < cnt; t++)
            {
                const TriGPU& tr = a.tris[n.tri_base + off + t];
                for (int k = 0; k < 3; k++)
                {
                    l[k] = fminf(l[k], fminf(tr.v0[k], fminf(tr.v1[k], tr.v2[k])));
                    h[k] = fmaxf(h[k], fmaxf(tr.v0[k], fmaxf(tr.v1[k], tr.v2[k])));
                }
            }
            if (in >= 0)
            {
                // the leaf's object-space cell through the instance's matrix: centre c' = M c, half extent e' = |mat3(M)| e, rounded outwards
                const float*       cell = a.cells + ((size_t)ni * 8 + c) * 6;
                const float*       M    = a.inst[in].m;
                const double cx = 0.5 * ((double)cell[0] + cell[3]), cy = 0.5 * ((double)cell[1] + cell[4]), cz = 0.5 * ((double)cell[2] + cell[5]);
                const double ex = 0.5 * ((double)cell[3] - cell[0]), ey = 0.5 * ((double)cell[4] - cell[1]), ez = 0.5 * ((double)cell[5] - cell[2]);
                for (int k = 0; k < 3; k++)
                {
                    const double wc = ((double)M[k] * cx + (double)M[4 + k] * cy) + ((double)M[8 + k] * cz + (double)M[12 + k]);
                    const double we = (fabs((double)M[k]) * ex + fabs((double)M[4 + k]) * ey) + fabs((double)M[8 + k]) * ez;
                    const double sl = 1e-12 * (fabs(wc) + we);   // the double arithmetic's own rounding, generously
                    l[k] = fmaxf(l[k], round_down(wc - we - sl));
                    h[k] = fminf(h[k], round_up(wc + we + sl));
                }
            }
            for (int k = 0; k < 3; k++)
            {
                if (h[k] < l[k]) h[k] = l[k];   // (cell and triangles disjoint up to rounding: cannot happen for a builder cell, harmless if it did)
                clo[c][k] = l[k] - a.pad; chi[c][k] = h[k] + a.pad;   // bvh_build.cpp finalise(pad)
            }
        }
        for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], clo[c][k]); hi[k] = fmaxf(hi[k], chi[c][k]); }
    }
    if (n_children == 0) { for (int k = 0; k < 3; k++) { lo[k] = 0.0f; hi[k] = 0.0f; } }
    float* nb = a.node_box + (size_t)ni * 8;
    nb[0] = lo[0]; nb[1] = lo[1]; nb[2] = lo[2]; nb[3] = 0.0f; nb[4] = hi[0]; nb[5] = hi[1]; nb[6] = hi[2]; nb[7] = 0.0f;
    n.ox = lo[0]; n.oy = lo[1]; n.oz = lo[2];
    n.ex = exponent_for_dev(hi[0] - lo[0]); n.ey = exponent_for_dev(hi[1] - lo[1]); n.ez = exponent_for_dev(hi[2] - lo[2]);
    const uint8_t eb[3] = { n.ex, n.ey, n.ez };
    for (int c = 0; c < 8; c++)
        for (int k = 0; k < 3; k++)
        {
            uint8_t ql = 0, qh = 0;
            if (c < n_children)
            {
                // child box = origin + q * 2^(e - 127), lo floored / hi ceiled => conservative (bvh.h)
                const double s = ldexp(1.0, (int)eb[k] - 127), o = (double)lo[k];
                double l = floor(((double)clo[c][k] - o) / s), h = ceil(((double)chi[c][k] - o) / s);
                if (!(l > 0.0)) l = 0.0;
                if (l > 255.0) l = 255.0;
                if (!(h < 255.0)) h = 255.0;
                if (h < l) h = l;
                ql = (uint8_t)l; qh = (uint8_t)h;
            }
            n.qlo[k][c] = ql; n.qhi[k][c] = qh;
        }
    a.nodes[ni] = n;
}

__global__ __launch_bounds__(64) void k_instances_refit(RefitArgs a)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < a.count) refit_node(a, a.list[i]);
}

// The levels near the root hold a handful of nodes each: ONE workgroup walks them all, deepest first, a barrier between levels (the nodes a level
// reads were written by this workgroup: __syncthreads orders them) — a dozen launches of a few microseconds each become one.
struct RefitTopArgs { RefitArgs r; const uint32_t* lists; int offs[kMaxTraversalDepth + 2]; int d_top; };
__global__ __launch_bounds__(256) void k_instances_refit_top(RefitTopArgs t)
{
    for (int d = t.d_top; d >= 0; d--)
    {
        for (int i = t.offs[d] + (int)threadIdx.x; i < t.offs[d + 1]; i += 256) refit_node(t.r, t.lists[i]);
        __threadfence_block();
        __syncthreads();
    }
}

// ordered-integer image of a float: unsigned comparison == float comparison
__device__ uint32_t ordered_bits(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
float ordered_float(uint32_t k)
{
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

struct TransformArgs
{
    const InstanceRec* inst;
    const uint32_t*    tri_instance;
    const uint32_t*    dirty;           // per instance
    const float*       mesh_positions;
    const float*       mesh_normals;    // or null
    float*             positions;       // world, by global triangle
    float*             normals;         // world mat3(model) * n (NOT normalised: the synthesiser normalises the interpolated sum), or null
    uint32_t*          bounds_bits;     // [n_instances][6] ordered-integer min xyz, max xyz of the instance's world vertices
    int                n_tris;
};

__global__ __launch_bounds__(256) void k_instances_reset_bounds(uint32_t* bounds_bits, const uint32_t* dirty, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !dirty[i]) return;
    for (int k = 0; k < 3; k++) { bounds_bits[(size_t)i * 6 + k] = 0xffffffffu; bounds_bits[(size_t)i * 6 + 3 + k] = 0u; }
}

__global__ __launch_bounds__(256) void k_instances_transform(TransformArgs a)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    int   ii = -1;
    if (t < a.n_tris)
    {
        const uint32_t in = a.tri_instance[t];
        if (a.dirty[in])
        {
            ii = (int)in;
            const InstanceRec& r = a.inst[in];
            const size_t q = (size_t)r.mesh_tri_base + ((uint32_t)t - r.first_tri);
            const float* p = a.mesh_positions + q * 9;
            float*       w = a.positions + (size_t)t * 9;
            for (int v = 0; v < 3; v++)
            {
                // model_matrix * vec4(p, 1): ((m0 x + m1 y) + m2 z) + m3 w per row (scene_descriptor_set.glsl:155; device_math.h mul_m4)
                const f4 x = mul_m4(r.m, p[v * 3], p[v * 3 + 1], p[v * 3 + 2], 1.0f);
                w[v * 3] = x.x; w[v * 3 + 1] = x.y; w[v * 3 + 2] = x.z;
                const float c[3] = { x.x, x.y, x.z };
                for (int k = 0; k < 3; k++) { if (c[k] < lo[k]) lo[k] = c[k]; if (c[k] > hi[k]) hi[k] = c[k]; }
            }
            if (a.mesh_normals)
            {
                const float* n = a.mesh_normals + q * 9;
                float*       o = a.normals + (size_t)t * 9;
                for (int v = 0; v < 3; v++)
                {
                    const float x = n[v * 3], y = n[v * 3 + 1], z = n[v * 3 + 2];
                    o[v * 3]     = (r.m[0] * x + r.m[4] * y) + r.m[8] * z;
                    o[v * 3 + 1] = (r.m[1] * x + r.m[5] * y) + r.m[9] * z;
                    o[v * 3 + 2] = (r.m[2] * x + r.m[6] * y) + r.m[10] * z;
                }
            }
        }
    }
    // per-instance bounds: one atomic pair per axis per WAVE when the wave's live lanes share the instance (big meshes), per lane otherwise
    const int  first = __builtin_amdgcn_readfirstlane(ii);
    const bool same  = __all(ii == first);
    if (same)
    {
        if (first < 0) return;
        for (int k = 0; k < 3; k++)
        {
            float l = lo[k], h = hi[k];
            for (int o = 32; o > 0; o >>= 1) { l = fminf(l, __shfl_xor(l, o)); h = fmaxf(h, __shfl_xor(h, o)); }
            if ((threadIdx.x & 63) == 0 && l <= h) { atomicMin(a.bounds_bits + (size_t)first * 6 + k, ordered_bits(l)); atomicMax(a.bounds_bits + (size_t)first * 6 + 3 + k, ordered_bits(h)); }
        }
    }
    else if (ii >= 0)
        for (int k = 0; k < 3; k++)
            if (lo[k] <= hi[k]) { atomicMin(a.bounds_bits + (size_t)ii * 6 + k, ordered_bits(lo[k])); atomicMax(a.bounds_bits + (size_t)ii * 6 + 3 + k, ordered_bits(hi[k])); }
}

__global__ __launch_bounds__(256) void k_instances_gather(TriGPU* tris, const float* positions, const uint32_t* tri_instance, const uint32_t* dirty, int n_refs)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_refs) return;
    TriGPU t = tris[r];
    if (!dirty[tri_instance[t.prim]]) return;
    const float* p = positions + (size_t)t.prim * 9;
    t.v0[0] = p[0]; t.v0[1] = p[1]; t.v0[2] = p[2];
    t.v1[0] = p[3]; t.v1[1] = p[4]; t.v1[2] = p[5];
    t.v2[0] = p[6]; t.v2[1] = p[7]; t.v2[2] = p[8];
    tris[r] = t;
}

bool finite16(const float* m)
{
    for (int i = 0; i < 16; i++)
        if (!std::isfinite(m[i])) return false;
    return true;
}

// ---- the top level ------------------------------------------------------------------------------------------------------------------
// Host side.  Inputs are the instances' conservative world boxes (mesh box corners through the matrix, in double); the result is a binary
// SAH tree (full sweep along the three axes of the box centres) collapsed to 8-wide nodes by opening the child with the largest area first,
// laid out breadth first in the first top_cap node slots: the children of a node — other top-level nodes and instance ROOTS alike, both are
// "internal children" to the traversal — sit contiguously, sorted along the node's longest axis (the traversal's near-to-far hint).
struct TopLevel
{
    std::vector<Node8>   nodes;       // top_cap slots; instance-root slots are filled in by place_roots()
    std::vector<int32_t> inst;        // per slot: -1 = top-level node, else the instance whose root sits there
    std::vector<int32_t> depth;       // per slot
    std::vector<int32_t> root_slot;   // per instance
    int                  used = 0, max_depth = 0;
};
struct BinNode { float lo[3], hi[3]; int left, right, inst; };

inline double half_area(const float* lo, const float* hi)
{
    const double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
    return x * y + y * z + z * x;
}

int build_binary(std::vector<BinNode>& t, std::vector<int>& items, int begin, int end, const float* boxes, int depth)
{
    BinNode n;
    for (int a = 0; a < 3; a++) { n.lo[a] = INFINITY; n.hi[a] = -INFINITY; }
    for (int k = begin; k < end; k++)
        for (int a = 0; a < 3; a++) { n.lo[a] = std::min(n.lo[a], boxes[(size_t)items[(size_t)k] * 6 + a]); n.hi[a] = std::max(n.hi[a], boxes[(size_t)items[(size_t)k] * 6 + 3 + a]); }
    n.left = n.right = -1; n.inst = -1;
    const int id = (int)t.size();
    t.push_back(n);
    const int cnt = end - begin;
    if (cnt == 1) { t[(size_t)id].inst = items[(size_t)begin]; return id; }
    int    best_axis = -1, best_split = begin + cnt / 2;
    double best_cost = 1e300;
    std::vector<int>    sorted(items.begin() + begin, items.begin() + end), best_sorted;
    std::vector<double> right_area((size_t)cnt);
    if (depth < 40)   // beyond that: median splits (the traversal stack bounds the depth)
        for (int a = 0; a < 3; a++)
        {
            std::stable_sort(sorted.begin(), sorted.end(), [&](int x, int y) {
                return (double)boxes[(size_t)x * 6 + a] + boxes[(size_t)x * 6 + 3 + a] < (double)boxes[(size_t)y * 6 + a] + boxes[(size_t)y * 6 + 3 + a]; });
            float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
            for (int k = cnt - 1; k > 0; k--)
            {
                for (int b = 0; b < 3; b++) { lo[b] = std::min(lo[b], boxes[(size_t)sorted[(size_t)k] * 6 + b]); hi[b] = std::max(hi[b], boxes[(size_t)sorted[(size_t)k] * 6 + 3 + b]); }
                right_area[(size_t)k] = half_area(lo, hi);
            }
            for (int b = 0; b < 3; b++) { lo[b] = INFINITY; hi[b] = -INFINITY; }
            for (int k = 1; k < cnt; k++)
            {
                for (int b = 0; b < 3; b++) { lo[b] = std::min(lo[b], boxes[(size_t)sorted[(size_t)k - 1] * 6 + b]); hi[b] = std::max(hi[b], boxes[(size_t)sorted[(size_t)k - 1] * 6 + 3 + b]); }
                const double c = half_area(lo, hi) * k + right_area[(size_t)k] * (cnt - k);
                if (c < best_cost) { best_cost = c; best_axis = a; best_split = begin + k; best_sorted = sorted; }
            }
        }
    if (best_axis >= 0) std::copy(best_sorted.begin(), best_sorted.end(), items.begin() + begin);
    const int l = build_binary(t, items, begin, best_split, boxes, depth + 1);
    const int r = build_binary(t, items, best_split, end, boxes, depth + 1);
    t[(size_t)id].left = l; t[(size_t)id].right = r;
    return id;
}

// per instance: conservative world box (host); also the scene's conservative bounds (grid_lo / grid_hi)
void instance_boxes(hr_scene* s)
{
    const int I = s->n_instances;
    s->inst_box.assign((size_t)I * 6, 0.0f);
    double L[3] = { 1e300, 1e300, 1e300 }, H[3] = { -1e300, -1e300, -1e300 };
    for (int i = 0; i < I; i++)
    {
        const float* m  = s->inst_host[(size_t)i].m;
        const float* mb = &s->mesh_bounds[(size_t)s->inst_mesh[(size_t)i] * 6];
        double l[3] = { 1e300, 1e300, 1e300 }, h[3] = { -1e300, -1e300, -1e300 };
        if (mb[0] <= mb[3])
            for (int c = 0; c < 8; c++)
            {
                const double x = mb[(c & 1) ? 3 : 0], y = mb[(c & 2) ? 4 : 1], z = mb[(c & 4) ? 5 : 2];
                for (int k = 0; k < 3; k++)
                {
                    const double v = (double)m[k] * x + (double)m[4 + k] * y + (double)m[8 + k] * z + (double)m[12 + k];
                    const double e = 1e-6 * (std::fabs((double)m[k] * x) + std::fabs((double)m[4 + k] * y) + std::fabs((double)m[8 + k] * z) + std::fabs((double)m[12 + k]));
                    l[k] = std::min(l[k], v - e); h[k] = std::max(h[k], v + e);
                }
            }
        else
            for (int k = 0; k < 3; k++) { l[k] = h[k] = (double)m[12 + k]; }   // empty mesh: a point at the instance's origin
        for (int k = 0; k < 3; k++)
        {
            float lo = (float)l[k], hi = (float)h[k];
            if ((double)lo > l[k]) lo = std::nextafter(lo, -INFINITY);
            if ((double)hi < h[k]) hi = std::nextafter(hi, INFINITY);
            s->inst_box[(size_t)i * 6 + k] = lo; s->inst_box[(size_t)i * 6 + 3 + k] = hi;
            if (mb[0] <= mb[3]) { L[k] = std::min(L[k], (double)lo); H[k] = std::max(H[k], (double)hi); }
        }
    }
    for (int k = 0; k < 3; k++)
    {
        if (!(L[k] <= H[k])) { L[k] = 0.0; H[k] = 0.0; }
        s->grid_lo[k] = (float)L[k]; s->grid_hi[k] = (float)H[k];
    }
}

void build_top_level(const hr_scene* s, TopLevel& tl)
{
    const int I = s->n_instances;
    tl.nodes.assign((size_t)s->top_cap, Node8());
    std::memset(tl.nodes.data(), 0, tl.nodes.size() * sizeof(Node8));
    tl.inst.assign((size_t)s->top_cap, -1);
    tl.depth.assign((size_t)s->top_cap, 0);
    tl.root_slot.assign((size_t)I, 0);
    tl.used = 1; tl.max_depth = 0;
    if (I == 1) { tl.inst[0] = 0; return; }
    std::vector<BinNode> bin;
    bin.reserve((size_t)I * 2);
    std::vector<int> items((size_t)I);
    for (int i = 0; i < I; i++) items[(size_t)i] = i;
    const int root = build_binary(bin, items, 0, I, s->inst_box.data(), 0);
    struct Q { int bin, slot, depth; };
    std::vector<Q> queue { { root, 0, 0 } };
    for (size_t qi = 0; qi < queue.size(); qi++)
    {
        const Q q = queue[qi];
        // open the child of largest area until eight children stand (instance leaves cannot be opened)
        int kids[8], nk = 2;
        kids[0] = bin[(size_t)q.bin].left; kids[1] = bin[(size_t)q.bin].right;
        while (nk < 8)
        {
            int    best = -1;
            double ba = -1.0;
            for (int c = 0; c < nk; c++)
                if (bin[(size_t)kids[c]].inst < 0)
                {
                    const double ar = half_area(bin[(size_t)kids[c]].lo, bin[(size_t)kids[c]].hi);
                    if (ar > ba) { ba = ar; best = c; }
                }
            if (best < 0) break;
            const int k = kids[best];
            kids[best] = bin[(size_t)k].left; kids[nk++] = bin[(size_t)k].right;
        }
        // sorted along the longest axis of the node's box: the traversal walks them near to far / far to near by the ray's sign (bvh.h)
        const BinNode& me = bin[(size_t)q.bin];
        int ax = 0;
        if (me.hi[1] - me.lo[1] > me.hi[ax] - me.lo[ax]) ax = 1;
        if (me.hi[2] - me.lo[2] > me.hi[ax] - me.lo[ax]) ax = 2;
        std::stable_sort(kids, kids + nk, [&](int x, int y) { return (double)bin[(size_t)x].lo[ax] + bin[(size_t)x].hi[ax] < (double)bin[(size_t)y].lo[ax] + bin[(size_t)y].hi[ax]; });
        Node8& n = tl.nodes[(size_t)q.slot];
        n.ex = n.ey = n.ez = 1;
        n.counts = (uint8_t)(nk | (nk << 4));
        n.child_base = (uint32_t)tl.used;
        n.tri_base = 0;
        for (int c = 0; c < nk; c++) n.meta[c] = (uint8_t)(0x10 | (c == 0 ? ax : 0));
        tl.inst[(size_t)q.slot] = -1; tl.depth[(size_t)q.slot] = q.depth;
        for (int c = 0; c < nk; c++)
        {
            const int slot = tl.used++;
            tl.depth[(size_t)slot] = q.depth + 1;
            tl.max_depth = std::max(tl.max_depth, q.depth + 1);
            if (bin[(size_t)kids[c]].inst >= 0) { tl.inst[(size_t)slot] = bin[(size_t)kids[c]].inst; tl.root_slot[(size_t)bin[(size_t)kids[c]].inst] = slot; }
            else queue.push_back({ kids[c], slot, q.depth + 1 });
        }
    }
}

// half-area sum of the top-level nodes with the instances' CURRENT boxes (host refit of the top region): what a re-build is judged by
double top_level_area(const hr_scene* s, const TopLevel* fresh = nullptr)
{
    const std::vector<Node8>&   nodes = fresh ? fresh->nodes : s->top_nodes_host;
    const std::vector<int32_t>& inst  = fresh ? fresh->inst : s->node_inst_host;
    const int used = fresh ? fresh->used : s->top_used;
    if (s->n_instances <= 1) return 0.0;
    std::vector<float> box((size_t)used * 6);
    double sum = 0.0;
    for (int slot = used - 1; slot >= 0; slot--)   // children sit behind their parent
    {
        float* b = &box[(size_t)slot * 6];
        if (inst[(size_t)slot] >= 0) { std::memcpy(b, &s->inst_box[(size_t)inst[(size_t)slot] * 6], 24); continue; }
        const Node8& n = nodes[(size_t)slot];
        for (int a = 0; a < 3; a++) { b[a] = INFINITY; b[3 + a] = -INFINITY; }
        for (int c = 0; c < (n.counts >> 4); c++)
            for (int a = 0; a < 3; a++) { b[a] = std::min(b[a], box[((size_t)n.child_base + c) * 6 + a]); b[3 + a] = std::max(b[3 + a], box[((size_t)n.child_base + c) * 6 + 3 + a]); }
        sum += half_area(b, b + 3);
    }
    return sum;
}

// writes the top level into the scene's host mirrors: node slots (top-level nodes + instance roots), owner / cell arrays, level lists
void adopt_top_level(hr_scene* s, const TopLevel& tl)
{
    const int I = s->n_instances;
    const size_t n_nodes = s->node_inst_host.size();
    s->top_nodes_host = tl.nodes;
    s->top_used = tl.used;
    s->inst_root_slot = tl.root_slot;
    s->top_cells_host.assign((size_t)s->top_cap * 48, 0.0f);
    for (int slot = 0; slot < s->top_cap; slot++) s->node_inst_host[(size_t)slot] = -1;
    for (int i = 0; i < I; i++)
    {
        const int slot = tl.root_slot[(size_t)i];
        s->top_nodes_host[(size_t)slot] = s->inst_root_node[(size_t)i];
        s->node_inst_host[(size_t)slot] = i;
        std::memcpy(&s->top_cells_host[(size_t)slot * 48], &s->inst_root_cells[(size_t)i * 48], 48 * sizeof(float));
    }
    // unused slots of the region: childless nodes nobody points at (the refit skips them: depth -1)
    int max_depth = 0;
    std::vector<int32_t> depth(n_nodes, -1);
    for (int slot = 0; slot < tl.used; slot++) depth[(size_t)slot] = tl.depth[(size_t)slot];
    for (size_t j = (size_t)s->top_cap; j < n_nodes; j++)
    {
        const int i = s->node_inst_host[j];
        depth[j] = tl.depth[(size_t)tl.root_slot[(size_t)i]] + s->node_rel_depth[j];
    }
    for (size_t j = 0; j < n_nodes; j++) max_depth = std::max(max_depth, depth[j]);
    s->info.max_depth = max_depth;
    s->level_offsets.assign((size_t)max_depth + 2, 0);
    for (size_t j = 0; j < n_nodes; j++) if (depth[j] >= 0) s->level_offsets[(size_t)depth[j] + 1]++;
    for (size_t dd = 0; dd + 1 < s->level_offsets.size(); dd++) s->level_offsets[dd + 1] += s->level_offsets[dd];
    s->level_nodes_host.assign(n_nodes, 0u);
    std::vector<int32_t> cur(s->level_offsets.begin(), s->level_offsets.end() - 1);
    for (size_t j = 0; j < n_nodes; j++) if (depth[j] >= 0) s->level_nodes_host[(size_t)cur[(size_t)depth[j]]++] = (uint32_t)j;
    s->top_area_at_build = top_level_area(s);
}

// the top region + the level lists to the device, ordered on `st`
hr_status upload_top_level(hr_scene* s, hipStream_t st)
{
    HR_HIP(hipMemcpyAsync(s->nodes.p, s->top_nodes_host.data(), (size_t)s->top_cap * sizeof(Node8), hipMemcpyHostToDevice, st));
    HR_HIP(hipMemcpyAsync(s->node_inst.p, s->node_inst_host.data(), (size_t)s->top_cap * 4, hipMemcpyHostToDevice, st));
    HR_HIP(hipMemcpyAsync(s->leaf_cells.p, s->top_cells_host.data(), (size_t)s->top_cap * 48 * 4, hipMemcpyHostToDevice, st));
    HR_HIP(hipMemcpyAsync(s->level_nodes.p, s->level_nodes_host.data(), s->level_nodes_host.size() * 4, hipMemcpyHostToDevice, st));
    return HR_OK;
}

hr_status update_impl(hr_scene* s, const float* matrices, hipStream_t st, bool all_dirty)
{
    const int m = s->n_instances;
    for (int i = 0; i < m; i++)
    {
        if (!finite16(matrices + (size_t)i * 16)) { set_last_error("hr_scene_update_instances: model_matrices[" + std::to_string(i) + "] is not finite"); return HR_ERR_INVALID_ARG; }
    }
    // which instances moved: their subtrees are refitted, the others stand
    s->inst_dirty.resize((size_t)m);
    bool any = false;
    for (int i = 0; i < m; i++)
    {
        const bool d = all_dirty || std::memcmp(s->inst_host[(size_t)i].m, matrices + (size_t)i * 16, 64) != 0;
        s->inst_dirty[(size_t)i] = d ? 1u : 0u;
        any = any || d;
        std::memcpy(s->inst_host[(size_t)i].m, matrices + (size_t)i * 16, 64);
    }
    if (!any) return HR_OK;
    HR_HIP(hipSetDevice(s->ctx->device));
    // the records are small (80 B per instance); the copies are ordered on `st` and read the scene's own host arrays, which live until the next update
    HR_HIP(hipMemcpyAsync(s->inst_records.p, s->inst_host.data(), (size_t)m * sizeof(InstanceRec), hipMemcpyHostToDevice, st));
    HR_HIP(hipMemcpyAsync(s->inst_dirty_dev.p, s->inst_dirty.data(), (size_t)m * 4, hipMemcpyHostToDevice, st));
    instance_boxes(s);
    {
        const double dx = (double)s->grid_hi[0] - s->grid_lo[0], dy = (double)s->grid_hi[1] - s->grid_lo[1], dz = (double)s->grid_hi[2] - s->grid_lo[2];
        float pad = (float)(3e-5 * std::sqrt(dx * dx + dy * dy + dz * dz));   // bvh_build.cpp: well above the fp32 error of the triangle test
        if (!(pad > 0.0f)) pad = 1e-6f;
        // subtrees that stand keep the pad they were refitted with; the refitted ones never get a smaller one (250x the fp32 epsilon of the
        // scene's diagonal: a scene that doubles in size still leaves the standing boxes a margin of two orders of magnitude)
        if (all_dirty || pad > s->info.box_pad) s->info.box_pad = pad;
    }
    // The top level is re-built when the instances have moved far enough for its boxes to overlap: the half-area sum of its nodes over the
    // instances' current boxes against the sum when it was built (host arithmetic over `instances` boxes; the reference re-builds its TLAS
    // every frame, main.cpp:74).  A re-build re-places the instance roots, so every level is refitted once.
    if (!all_dirty && m > 1 && s->auto_rebuild && top_level_area(s) > s->rebuild_ratio * s->top_area_at_build)
    {
        TopLevel tl;
        build_top_level(s, tl);
        if (top_level_area(s, &tl) < 0.9 * top_level_area(s))   // only if the fresh one IS better (instances that merely spread out gain nothing)
        {
            adopt_top_level(s, tl);
            const hr_status us = upload_top_level(s, st);
            if (us != HR_OK) return us;
            s->top_rebuilds++;
            all_dirty = true;
            for (int i = 0; i < m; i++) s->inst_dirty[(size_t)i] = 1u;
            HR_HIP(hipMemcpyAsync(s->inst_dirty_dev.p, s->inst_dirty.data(), (size_t)m * 4, hipMemcpyHostToDevice, st));
        }
        else s->top_area_at_build = top_level_area(s);   // the spread is the new normal
    }
    const uint32_t* dirty = (const uint32_t*)s->inst_dirty_dev.p;
    hipLaunchKernelGGL(k_instances_reset_bounds, dim3(cdiv(m, 256)), dim3(256), 0, st, (uint32_t*)s->bounds_bits.p, dirty, m);
    const int n_tris = s->info.n_tris, n_refs = (int)(s->tris.bytes / sizeof(TriGPU));
    if (n_tris > 0)
    {
        TransformArgs t;
        t.inst = (const InstanceRec*)s->inst_records.p; t.tri_instance = (const uint32_t*)s->tri_instance.p; t.dirty = dirty;
        t.mesh_positions = (const float*)s->mesh_positions.p; t.mesh_normals = s->has_normals ? (const float*)s->mesh_normals.p : nullptr;
        t.positions = (float*)s->positions.p; t.normals = s->has_normals ? (float*)s->tri_normals.p : nullptr;
        t.bounds_bits = (uint32_t*)s->bounds_bits.p; t.n_tris = n_tris;
        hipLaunchKernelGGL(k_instances_transform, dim3(cdiv(n_tris, 256)), dim3(256), 0, st, t);
        hipLaunchKernelGGL(k_instances_gather, dim3(cdiv(n_refs, 256)), dim3(256), 0, st, (TriGPU*)s->tris.p, (const float*)s->positions.p, (const uint32_t*)s->tri_instance.p, dirty, n_refs);
    }
    RefitArgs r;
    r.nodes = (Node8*)s->nodes.p; r.tris = (const TriGPU*)s->tris.p; r.node_box = (float*)s->node_box.p; r.pad = s->info.box_pad;
    r.cells = (const float*)s->leaf_cells.p; r.node_inst = (const int32_t*)s->node_inst.p; r.inst = (const InstanceRec*)s->inst_records.p; r.dirty = dirty;
    // levels 0 .. d_top (each no larger than a few hundred nodes) in one launch, the wide levels below them one launch each
    const int n_levels = (int)s->level_offsets.size() - 1;
    int d_top = -1;
    while (d_top + 1 < n_levels && d_top + 1 <= kMaxTraversalDepth && s->level_offsets[(size_t)d_top + 2] - s->level_offsets[(size_t)d_top + 1] <= 512) d_top++;
    for (int d = n_levels - 1; d > d_top; d--)
    {
        r.list = (const uint32_t*)s->level_nodes.p + s->level_offsets[(size_t)d];
        r.count = s->level_offsets[(size_t)d + 1] - s->level_offsets[(size_t)d];
        if (r.count > 0) hipLaunchKernelGGL(k_instances_refit, dim3(cdiv(r.count, 64)), dim3(64), 0, st, r);
    }
    if (d_top >= 0)
    {
        RefitTopArgs t;
        t.r = r; t.lists = (const uint32_t*)s->level_nodes.p; t.d_top = d_top;
        for (int d = 0; d <= d_top + 1; d++) t.offs[d] = s->level_offsets[(size_t)d];
        hipLaunchKernelGGL(k_instances_refit_top, dim3(1), dim3(256), 0, st, t);
    }
    HR_HIP(hipGetLastError());
    s->geometry_epoch++;
    s->bounds_stale = true;
    return HR_OK;
}

hr_status create_impl(hr_ctx* ctx, const hr_instanced_scene_desc* d, hr_scene** out)
{
    HR_CHECK_ARG(ctx && d && out && d->n_meshes > 0 && d->meshes && d->n_instances > 0 && d->instances);
    HR_CHECK_ARG(d->n_materials >= 0 && (d->materials || d->n_materials == 0));
    const int M = d->n_meshes, I = d->n_instances;
    bool all_normals = true, any_normals = false, all_mat = true, any_mat = false, all_uv = true, all_tan = true;
    for (int k = 0; k < M; k++)
    {
        const hr_mesh_desc& me = d->meshes[k];
        HR_CHECK_ARG(me.n_tris >= 0 && (me.positions || me.n_tris == 0));
        all_normals = all_normals && me.normals; any_normals = any_normals || me.normals;
        all_mat = all_mat && me.tri_material; any_mat = any_mat || me.tri_material;
        all_uv = all_uv && me.uvs; all_tan = all_tan && me.tangents;
        if (me.tri_material)
        {
            if (!d->materials) { set_last_error("hr_scene_create_instanced: tri_material given without materials"); return HR_ERR_INVALID_ARG; }
            for (int i = 0; i < me.n_tris; i++)
                if (me.tri_material[i] >= (uint32_t)d->n_materials) { set_last_error("hr_scene_create_instanced: a tri_material entry >= n_materials"); return HR_ERR_INVALID_ARG; }
        }
    }
    if (any_normals && !all_normals) { set_last_error("hr_scene_create_instanced: vertex normals on some meshes only"); return HR_ERR_INVALID_ARG; }
    if (any_mat && !all_mat) { set_last_error("hr_scene_create_instanced: tri_material on some meshes only"); return HR_ERR_INVALID_ARG; }
    for (int i = 0; i < I; i++)
    {
        if (d->instances[i].mesh_idx >= (uint32_t)M) { set_last_error("hr_scene_create_instanced: instances[" + std::to_string(i) + "].mesh_idx >= n_meshes"); return HR_ERR_INVALID_ARG; }
        if (!finite16(d->instances[i].model_matrix)) { set_last_error("hr_scene_create_instanced: instances[" + std::to_string(i) + "].model_matrix is not finite"); return HR_ERR_INVALID_ARG; }
    }
    HR_HIP(hipSetDevice(ctx->device));

    // ---- per-mesh topologies (object space) ------------------------------------------------------------------------------------------
    std::vector<BuiltBVH> blas((size_t)M);
    std::vector<std::vector<int>> blas_depth((size_t)M);
    std::vector<uint32_t> mesh_tri_base((size_t)M + 1, 0u);
    std::unique_ptr<hr_scene> guard(new hr_scene());
    hr_scene* s = guard.get();
    s->ctx = ctx;
    s->mesh_bounds.assign((size_t)M * 6, 0.0f);
    for (int k = 0; k < M; k++)
    {
        blas[(size_t)k].want_child_boxes = true;
        build_bvh8(d->meshes[k].positions, d->meshes[k].n_tris, blas[(size_t)k]);
        blas[(size_t)k].child_boxes.resize(blas[(size_t)k].nodes.size() * 48, 0.0f);
        const BuiltBVH& b = blas[(size_t)k];
        mesh_tri_base[(size_t)k + 1] = mesh_tri_base[(size_t)k] + (uint32_t)d->meshes[k].n_tris;
        for (int a = 0; a < 3; a++) { s->mesh_bounds[(size_t)k * 6 + a] = b.lo[a]; s->mesh_bounds[(size_t)k * 6 + 3 + a] = b.hi[a]; }
        if (d->meshes[k].n_tris == 0) { s->mesh_bounds[(size_t)k * 6] = 1.0f; s->mesh_bounds[(size_t)k * 6 + 3] = 0.0f; }   // empty: lo > hi
        // depth of every node (children follow their parent in the builder's breadth-first order)
        std::vector<int>& dep = blas_depth[(size_t)k];
        dep.assign(b.nodes.size(), 0);
        for (size_t j = 0; j < b.nodes.size(); j++)
            for (int c = 0; c < (b.nodes[j].counts & 15); c++) dep[(size_t)b.nodes[j].child_base + c] = dep[j] + 1;
    }

    // ---- layout: [ top region: top-level nodes + instance roots, top_cap slots | instance 0's other nodes | instance 1's ... ] ------------------
    s->n_instances = I;
    s->inst_mesh.resize((size_t)I);
    s->inst_host.resize((size_t)I);
    uint64_t total_tris = 0, total_refs = 0, total_sub_nodes = 0;
    for (int i = 0; i < I; i++)
    {
        const uint32_t k = d->instances[i].mesh_idx;
        s->inst_mesh[(size_t)i] = k;
        InstanceRec& r = s->inst_host[(size_t)i];
        std::memcpy(r.m, d->instances[i].model_matrix, 64);
        r.first_tri = (uint32_t)total_tris; r.mesh_tri_base = mesh_tri_base[k]; r.mesh_id = d->instances[i].mesh_id; r.n_tris = (uint32_t)d->meshes[k].n_tris;
        total_tris += (uint64_t)d->meshes[k].n_tris; total_refs += blas[k].tris.size(); total_sub_nodes += blas[k].nodes.size() - 1;
    }
    if (total_tris >= (1ull << 31) || total_refs >= (1ull << 26)) { set_last_error("hr_scene_create_instanced: more than 2^26 triangle references"); return HR_ERR_UNSUPPORTED; }
    s->top_cap = I > 1 ? 2 * I : 1;   // at most I - 1 top-level nodes over I roots
    const uint64_t n_nodes64 = (uint64_t)s->top_cap + total_sub_nodes;
    if (n_nodes64 >= (1ull << 23)) { set_last_error("hr_scene_create_instanced: more than 2^23 BVH nodes"); return HR_ERR_UNSUPPORTED; }
    const size_t n_nodes = (size_t)n_nodes64;
    std::vector<Node8>  nodes(n_nodes);
    std::vector<float>  cells(n_nodes * 48, 0.0f);     // object-space cell of every leaf slot
    std::vector<TriGPU> tris((size_t)total_refs);
    std::memset(nodes.data(), 0, n_nodes * sizeof(Node8));
    s->node_inst_host.assign(n_nodes, -1);
    s->node_rel_depth.assign(n_nodes, 0);
    s->inst_root_node.resize((size_t)I);
    s->inst_root_cells.assign((size_t)I * 48, 0.0f);
    size_t node_at = (size_t)s->top_cap, ref_at = 0;
    int    max_rel = 0;
    std::vector<uint32_t> tri_instance((size_t)total_tris);
    for (int i = 0; i < I; i++)
    {
        const uint32_t  k = s->inst_mesh[(size_t)i];
        const BuiltBVH& b = blas[k];
        const size_t    base = node_at;   // mesh node j >= 1 -> base + j - 1; the root (j = 0) goes wherever the top level puts it
        for (size_t j = 0; j < b.nodes.size(); j++)
        {
            Node8 n = b.nodes[j];
            if (n.counts & 15) n.child_base = (uint32_t)(base + n.child_base - 1);
            n.tri_base += (uint32_t)ref_at;
            if (j == 0)
            {
                s->inst_root_node[(size_t)i] = n;
                std::memcpy(&s->inst_root_cells[(size_t)i * 48], &b.child_boxes[0], 48 * sizeof(float));
                continue;
            }
            const size_t at = base + j - 1;
            nodes[at] = n;
            s->node_inst_host[at] = i;
            s->node_rel_depth[at] = blas_depth[k][j];
            std::memcpy(&cells[at * 48], &b.child_boxes[j * 48], 48 * sizeof(float));
            max_rel = std::max(max_rel, blas_depth[k][j]);
        }
        const InstanceRec& r = s->inst_host[(size_t)i];
        for (size_t t = 0; t < b.tris.size(); t++)
        {
            TriGPU tg = b.tris[t];
            tg.prim += r.first_tri;
            tris[ref_at + t] = tg;
        }
        for (uint32_t t = 0; t < r.n_tris; t++) tri_instance[(size_t)r.first_tri + t] = (uint32_t)i;
        node_at += b.nodes.size() - 1; ref_at += b.tris.size();
    }
    // ---- the top level over the instances' initial boxes ---------------------------------------------------------------------------------
    instance_boxes(s);
    {
        TopLevel tl;
        build_top_level(s, tl);
        if (tl.max_depth + max_rel + 1 >= kMaxTraversalDepth) { set_last_error("hr_scene_create_instanced: BVH depth exceeds the traversal stack"); return HR_ERR_UNSUPPORTED; }
        adopt_top_level(s, tl);
    }
    std::memcpy(nodes.data(), s->top_nodes_host.data(), (size_t)s->top_cap * sizeof(Node8));
    std::memcpy(cells.data(), s->top_cells_host.data(), (size_t)s->top_cap * 48 * sizeof(float));
    const std::vector<int32_t>&  node_inst = s->node_inst_host;
    const std::vector<uint32_t>& level_nodes = s->level_nodes_host;
    const int max_depth = s->info.max_depth;

    // ---- attributes --------------------------------------------------------------------------------------------------------------------
    const size_t MT = mesh_tri_base[(size_t)M], N = (size_t)total_tris;
    std::vector<float> mpos(MT * 9), mnor(all_normals ? MT * 9 : 0), muv(all_uv ? MT * 6 : 0), mtan(all_tan ? MT * 9 : 0);
    std::vector<uint32_t> mmat(all_mat ? MT : 0), gmat(all_mat ? N : 0), gid(N);
    for (int k = 0; k < M; k++)
    {
        const hr_mesh_desc& me = d->meshes[k];
        const size_t o = mesh_tri_base[(size_t)k], n = (size_t)me.n_tris;
        if (n == 0) continue;
        std::memcpy(&mpos[o * 9], me.positions, n * 36);
        if (all_normals) std::memcpy(&mnor[o * 9], me.normals, n * 36);
        if (all_uv) std::memcpy(&muv[o * 6], me.uvs, n * 24);
        if (all_tan) std::memcpy(&mtan[o * 9], me.tangents, n * 36);
        if (all_mat) std::memcpy(&mmat[o], me.tri_material, n * 4);
    }
    for (int i = 0; i < I; i++)
    {
        const InstanceRec& r = s->inst_host[(size_t)i];
        for (uint32_t t = 0; t < r.n_tris; t++)
        {
            gid[(size_t)r.first_tri + t] = r.mesh_id;
            if (all_mat) gmat[(size_t)r.first_tri + t] = mmat[(size_t)r.mesh_tri_base + t];
        }
    }
    hr_status st;
#define UP(buf, src, nbytes)                                                                     \
    if ((st = s->buf.alloc(nbytes)) != HR_OK) return st;                                         \
    if ((nbytes) > 0) { hipError_t e_ = hipMemcpy(s->buf.p, src, nbytes, hipMemcpyHostToDevice); \
        if (e_ != hipSuccess) { set_last_error(std::string("hipMemcpy H2D failed: ") + hipGetErrorString(e_)); return HR_ERR_HIP; } }
    UP(nodes, nodes.data(), n_nodes * sizeof(Node8))
    UP(tris, tris.data(), tris.size() * sizeof(TriGPU))
    UP(level_nodes, level_nodes.data(), n_nodes * 4)
    UP(leaf_cells, cells.data(), cells.size() * 4)
    UP(node_inst, node_inst.data(), n_nodes * 4)
    UP(tri_instance, tri_instance.data(), N * 4)
    UP(mesh_positions, mpos.data(), MT * 36)
    UP(materials, d->materials, d->materials ? (size_t)d->n_materials * 32 : 0)
    if (all_normals) { UP(mesh_normals, mnor.data(), MT * 36) s->has_normals = true; if ((st = s->tri_normals.alloc(N * 36)) != HR_OK) return st; }
    if (all_mat) { UP(mesh_material, mmat.data(), MT * 4) UP(tri_material, gmat.data(), N * 4) s->has_material = true; }
    UP(tri_mesh_id, gid.data(), N * 4)
    s->has_mesh_id = true;
    if ((st = s->positions.alloc(N * 36)) != HR_OK) return st;
    if ((st = s->node_box.alloc(n_nodes * 32)) != HR_OK) return st;
    if ((st = s->bounds_bits.alloc((size_t)I * 24)) != HR_OK) return st;
    if ((st = s->inst_dirty_dev.alloc((size_t)I * 4)) != HR_OK) return st;
    if ((st = s->inst_records.alloc((size_t)I * sizeof(InstanceRec))) != HR_OK) return st;
    if (d->material_textures && d->materials && d->n_textures > 0 && d->textures)
    {
        std::vector<uint32_t> table;
        std::vector<uint8_t>  texels;
        for (int i = 0; i < d->n_textures; i++)
        {
            const hr_texture& t = d->textures[i];
            if (!t.rgba8 || t.width <= 0 || t.height <= 0) { set_last_error("hr_scene_create_instanced: empty texture"); return HR_ERR_INVALID_ARG; }
            table.insert(table.end(), { (uint32_t)(texels.size() / 4), (uint32_t)t.width, (uint32_t)t.height, 0u });
            texels.insert(texels.end(), t.rgba8, t.rgba8 + (size_t)t.width * t.height * 4);
        }
        for (int i = 0; i < d->n_materials * 4; i++)
            if (d->material_textures[(i / 4) * 6 + (i % 4)] >= d->n_textures) { set_last_error("hr_scene_create_instanced: material texture index out of range"); return HR_ERR_INVALID_ARG; }
        UP(mat_tex, d->material_textures, (size_t)d->n_materials * 24)
        UP(tex_table, table.data(), table.size() * 4)
        UP(tex_data, texels.data(), texels.size())
        if (all_uv) { UP(mesh_uvs, muv.data(), MT * 24) s->has_uvs = true; }
        if (all_tan) { UP(mesh_tangents, mtan.data(), MT * 36) s->has_tangents = true; }
        s->has_textures = true;
    }
#undef UP
    s->n_materials = d->materials ? d->n_materials : 0;
    { static std::atomic<uint64_t> next_uid { 1ull << 40 }; s->uid = next_uid.fetch_add(1); }   // disjoint from hr_scene_create's counter
    if (const char* e = getenv("HR_TOP_LEVEL_REBUILD")) s->auto_rebuild = atoi(e) != 0;
    s->info.n_tris     = (int32_t)N;
    s->info.n_nodes    = (int32_t)n_nodes;
    (void)max_depth;   // set by adopt_top_level
    s->info.node_bytes = n_nodes * sizeof(Node8);
    s->info.tri_bytes  = tris.size() * sizeof(TriGPU);
    // first update: the instances' own matrices, then wait (creation is synchronous like hr_scene_create)
    std::vector<float> mats((size_t)I * 16);
    for (int i = 0; i < I; i++) std::memcpy(&mats[(size_t)i * 16], d->instances[i].model_matrix, 64);
    if ((st = update_impl(s, mats.data(), nullptr, true)) != HR_OK) return st;
    HR_HIP(hipStreamSynchronize(nullptr));
    s->geometry_epoch = 0;
    hr_scene_info tmp;
    if ((st = hr_scene_get_info(s, &tmp)) != HR_OK) return st;
    *out = guard.release();
    return HR_OK;
}

} // namespace

// hr_scene_get_info of an instanced scene: the exact bounds of the last update, read back on demand (synchronises the device)
hr_status hr::instanced_scene_refresh_bounds(const hr_scene* scene)
{
    if (!scene->bounds_stale) return HR_OK;
    hr_scene* s = const_cast<hr_scene*>(scene);
    HR_HIP(hipSetDevice(s->ctx->device));
    HR_HIP(hipDeviceSynchronize());
    std::vector<uint32_t> bits((size_t)s->n_instances * 6);
    HR_HIP(hipMemcpy(bits.data(), s->bounds_bits.p, bits.size() * 4, hipMemcpyDeviceToHost));
    uint32_t lo[3] = { 0xffffffffu, 0xffffffffu, 0xffffffffu }, hi[3] = { 0u, 0u, 0u };
    for (int i = 0; i < s->n_instances; i++)
        for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], bits[(size_t)i * 6 + k]); hi[k] = std::max(hi[k], bits[(size_t)i * 6 + 3 + k]); }
    for (int k = 0; k < 3; k++)
    {
        const bool any = lo[k] <= hi[k] && lo[k] != 0xffffffffu;
        s->info.bounds_lo[k] = any ? ordered_float(lo[k]) : 0.0f;
        s->info.bounds_hi[k] = any ? ordered_float(hi[k]) : 0.0f;
    }
    s->bounds_stale = false;
    return HR_OK;
}

extern "C" {

hr_status hr_scene_create_instanced(hr_ctx* ctx, const hr_instanced_scene_desc* desc, hr_scene** out)
{
    try
    {
        return create_impl(ctx, desc, out);
    }
    catch (const std::bad_alloc&)
    {
        set_last_error("hr_scene_create_instanced: host allocation failed");
        return HR_ERR_OUT_OF_MEMORY;
    }
    catch (const std::exception& e)
    {
        set_last_error(std::string("hr_scene_create_instanced: ") + e.what());
        return HR_ERR_UNSUPPORTED;
    }
}

hr_status hr_scene_update_instances(hr_scene* scene, const float* model_matrices, void* stream)
{
    HR_CHECK_ARG(scene && model_matrices);
    if (scene->n_instances <= 0) { set_last_error("hr_scene_update_instances: not an instanced scene (hr_scene_create_instanced)"); return HR_ERR_INVALID_ARG; }
    return update_impl(scene, model_matrices, (hipStream_t)stream, false);
}

int32_t hr_scene_instance_count(const hr_scene* scene) { return scene ? scene->n_instances : 0; }

hr_status hr_scene_rebuild_top_level(hr_scene* scene, void* stream)
{
    HR_CHECK_ARG(scene);
    if (scene->n_instances <= 0) { set_last_error("hr_scene_rebuild_top_level: not an instanced scene"); return HR_ERR_INVALID_ARG; }
    if (scene->n_instances == 1) return HR_OK;
    hipStream_t st = (hipStream_t)stream;
    HR_HIP(hipSetDevice(scene->ctx->device));
    TopLevel tl;
    build_top_level(scene, tl);
    adopt_top_level(scene, tl);
    const hr_status us = upload_top_level(scene, st);
    if (us != HR_OK) return us;
    scene->top_rebuilds++;
    std::vector<float> mats((size_t)scene->n_instances * 16);
    for (int i = 0; i < scene->n_instances; i++) std::memcpy(&mats[(size_t)i * 16], scene->inst_host[(size_t)i].m, 64);
    return update_impl(scene, mats.data(), st, true);   // every level once: the instance roots moved to other slots
}
int32_t hr_scene_top_level_rebuilds(const hr_scene* scene) { return scene ? scene->top_rebuilds : 0; }

} // extern "C"
