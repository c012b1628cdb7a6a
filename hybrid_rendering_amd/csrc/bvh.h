// Compressed 8-wide BVH: the software replacement for VK_KHR_acceleration_structure
// (reference call sites: main.cpp:74 build_tlas, common.cpp:355-521 initialize_for_ray_tracing).
//
// Node = 80 bytes (five 16-byte loads per lane):
//   [ 0] origin.xyz (f32), exponent bytes ex,ey,ez (biased like an fp32 exponent),
//        counts = n_internal | n_children << 4.  Slots 0..n_internal-1 are the internal children (slot i is node
//        child_base + i), slots n_internal..n_children-1 the leaves, the rest empty.
//   [16] child_base (index of first internal child), tri_base (index of first leaf triangle),
//        meta[8]: 0 = empty slot; internal: 0x10 (slot 0: 0x10 | the axis 0..2 the internal children are sorted along);
//                 leaf: (count << 5) | offset  (triangles tri_base+offset .. +count-1), count 1..4
//   [32] qlo.x[8] qlo.y[8]   [48] qlo.z[8] qhi.x[8]   [64] qhi.y[8] qhi.z[8]   (uint8 grid coords)
//   child box = origin + q * 2^(e-127), lo floored / hi ceiled => conservative.
// Triangle = 48 bytes: v0.xyz, prim | v1.xyz, 0 | v2.xyz, 0   (original vertex positions: the
// watertight test needs them unmodified for bit-reproducible hit decisions).
#pragma once
#include <stdint.h>
#include <vector>

namespace hr {

struct alignas(16) Node8
{
    float    ox, oy, oz;
    uint8_t  ex, ey, ez, counts;
    uint32_t child_base;
    uint32_t tri_base;
    uint8_t  meta[8];
    uint8_t  qlo[3][8];
    uint8_t  qhi[3][8];
};
static_assert(sizeof(Node8) == 80, "Node8 must be 80 bytes");

struct alignas(16) TriGPU
{
    float    v0[3];
    uint32_t prim;
    float    v1[3];
    uint32_t pad1;
    float    v2[3];
    uint32_t pad2;
};
static_assert(sizeof(TriGPU) == 48, "TriGPU must be 48 bytes");

// One instance of an instanced scene (instances.hip; scene_descriptor_set.glsl:30-34 Instance { mat4 model_matrix; uint mesh_idx; }): the
// hit shading maps a global triangle index to the mesh's object-space attributes and applies `m` (shading.h surface_at).
struct alignas(16) InstanceRec
{
    float    m[16];          // column-major object -> world
    uint32_t first_tri;      // global index of the instance's first triangle
    uint32_t mesh_tri_base;  // index of the mesh's first triangle in the concatenated per-mesh attribute arrays
    uint32_t mesh_id;        // GB3.z of the instance's pixels
    uint32_t n_tris;
};
static_assert(sizeof(InstanceRec) == 80, "InstanceRec must be 80 bytes");

// deepest 8-wide tree the traversal's per-lane stack can walk (one entry per level, traverse.h LaneStack)
constexpr int kMaxTraversalDepth = 64;

struct BuiltBVH
{
    std::vector<Node8>  nodes;
    std::vector<TriGPU> tris;
    float               lo[3], hi[3];
    float               pad;
    int                 max_depth;
    int                 n_refs = 0;   // triangle references after splitting (== tris.size())
    // want_child_boxes (set before build_bvh8): child_boxes[(node * 8 + slot) * 6 ..] = lo xyz, hi xyz of the slot's box as the builder had it
    // (padded, not yet quantised).  A LEAF's box may be smaller than the bounds of its triangles: the builder splits references spatially
    // (SBVH), and a leaf then bounds only the pieces inside its cell.  The instanced scenes' refit needs these cells (instances.hip).
    bool                want_child_boxes = false;
    std::vector<float>  child_boxes;
};

// positions: [n][3][3].  Deterministic: the result does not depend on the number of builder threads (HR_BVH_THREADS, default min(hardware, 16)).
void build_bvh8(const float* positions, int n_tris, BuiltBVH& out);

// Host-side self-check (bvh_build.cpp): number of (triangle, sample point) pairs that reach no leaf holding the triangle — 0 for a
// correct tree.
int64_t check_bvh8_coverage(const float* positions, int n_tris, const BuiltBVH& b, int samples_per_triangle);

} // namespace hr
