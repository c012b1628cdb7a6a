// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).  Pinned against the reference's own shaders: tests/test_ref_shaders.py.
//
// Ray/scene intersection.  In the reference this arithmetic is NOT in the tree: it is the
// Vulkan driver behind VK_KHR_acceleration_structure / GL_EXT_ray_query
// (call sites: ray_query.glsl:13-27,42-56; reflections_ray_trace.rgen:150,165;
// gi_ray_trace.rgen:96).  The Vulkan spec fixes only the semantics reproduced here:
//   * all geometry opaque (ray_query.glsl:37 gl_RayFlagsOpaqueEXT), no face culling;
//   * a triangle hit is a candidate iff  t_min < t < t_max;
//   * "terminate on first hit" queries return whether ANY candidate exists.
// The triangle test is pinned to the watertight test of Woop, Benthin, Wald (JCGT 2013) in
// fp32 with individually rounded ops, so the any-hit answer is a pure function of
// (ray, triangle set) and does not depend on the acceleration structure as long as box
// culling is conservative.  The BVH2 below (binned SAH) pads every box by 3e-5 * scene
// diagonal, far above the fp32 error of the triangle test; a brute-force
// path exists to validate that claim (tests/test_oracle_bvh.py).
#pragma once
#include "orc_math.h"
#include <vector>

namespace orc {

struct Tri { vec3 v0, v1, v2; };

struct Hit
{
    float    t;
    float    u, v;   // barycentrics of v1, v2
    int32_t  prim;   // triangle index, -1 = miss
};

struct RayPre
{
    vec3  o, d;
    int   kx, ky, kz;
    float Sx, Sy, Sz;
};

static inline float comp(vec3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }

static inline RayPre ray_prepare(vec3 o, vec3 d)
{
    RayPre r;
    r.o = o;
    r.d = d;
    float ax = std::fabs(d.x), ay = std::fabs(d.y), az = std::fabs(d.z);
    int   kz = 0;
    if (ay > ax) kz = 1;
    if (az > (kz == 0 ? ax : ay)) kz = 2;
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    if (comp(d, kz) < 0.0f) { int t = kx; kx = ky; ky = t; }
    r.kx = kx; r.ky = ky; r.kz = kz;
    float dz = comp(d, kz);
    r.Sx = comp(d, kx) / dz;
    r.Sy = comp(d, ky) / dz;
    r.Sz = 1.0f / dz;
    return r;
}

// Returns true if the triangle is hit with t_min < t < t_max.  T/det etc. returned for closest hit.
static inline bool ray_tri(const RayPre& r, const Tri& tr, float t_min, float t_max, float* t_out, float* u_out, float* v_out)
{
    vec3 A = tr.v0 - r.o, B = tr.v1 - r.o, C = tr.v2 - r.o;
    float Akz = comp(A, r.kz), Bkz = comp(B, r.kz), Ckz = comp(C, r.kz);
    float Ax = comp(A, r.kx) - r.Sx * Akz, Ay = comp(A, r.ky) - r.Sy * Akz;
    float Bx = comp(B, r.kx) - r.Sx * Bkz, By = comp(B, r.ky) - r.Sy * Bkz;
    float Cx = comp(C, r.kx) - r.Sx * Ckz, Cy = comp(C, r.ky) - r.Sy * Ckz;
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    float det = (U + V) + W;
    if (det == 0.0f) return false;
    float Az = r.Sz * Akz, Bz = r.Sz * Bkz, Cz = r.Sz * Ckz;
    float T  = (U * Az + V * Bz) + W * Cz;
    float sg = det < 0.0f ? -1.0f : 1.0f;
    float Ts = T * sg;
    float ad = det * sg;
    if (!(Ts > t_min * ad && Ts < t_max * ad)) return false;
    if (t_out)
    {
        float inv = 1.0f / det;
        *t_out    = T * inv;
        *u_out    = V * inv; // weight of v1
        *v_out    = W * inv; // weight of v2
    }
    return true;
}

struct BVH2Node
{
    float   lo[3], hi[3];
    int32_t left;  // internal: index of left child (right = left + 1); leaf: first triangle slot
    int32_t count; // 0 = internal, >0 = leaf triangle count
};

struct Scene
{
    std::vector<Tri>      tris;     // original order
    std::vector<int32_t>  order;    // BVH leaf order -> original triangle index
    std::vector<BVH2Node> nodes;
    float                 lo[3], hi[3];
    float                 pad;
    // per-triangle shading data (reflections / DDGI); optional
    std::vector<uint32_t> tri_material; // material index per triangle
    std::vector<float>    tri_normals;  // [n][3][3] vertex normals (optional; else geometric)
    std::vector<float>    materials;    // [m][8]: albedo rgb, metallic, roughness, emissive rgb
    std::vector<uint32_t> tri_mesh_id;
    // textured materials (scene_descriptor_set.glsl:20-27, :168-220); all optional
    std::vector<float>    tri_uvs;       // [n][3][2]
    std::vector<float>    tri_tangents;  // [n][3][3]
    std::vector<int32_t>  mat_tex;       // [m][6]: albedo, normal, roughness, metallic texture (-1 = none), roughness channel, metallic channel
    struct Texture { std::vector<uint8_t> rgba; int w, h; };
    std::vector<Texture>  textures;      // RGBA8 UNORM

    // instanced scenes (scene_descriptor_set.glsl:30-34 Instance, :150-160 transform_vertex): `tris` / `tri_normals` are then the WORLD-space
    // vertices / mat3(model) * n that orc_instances_flatten produced (intersection, G-buffer synthesis), and the hit shading (orc_shading.h
    // surface_at) interpolates the OBJECT-space per-mesh attributes below before it applies the instance's matrix
    struct Instance { float m[16]; uint32_t first_tri, mesh_tri_base, mesh_id, n_tris; };
    std::vector<Instance> instances;
    std::vector<uint32_t> tri_instance;                                       // [n] global triangle -> instance
    std::vector<float>    mesh_positions, mesh_normals, mesh_uvs, mesh_tangents;
    std::vector<uint32_t> mesh_material;

    void build(const float* verts, int n_tris);
    bool any_hit(vec3 o, vec3 d, float t_min, float t_max) const;
    bool any_hit_brute(vec3 o, vec3 d, float t_min, float t_max) const;
    Hit  closest_hit(vec3 o, vec3 d, float t_min, float t_max) const;
    Hit  closest_hit_brute(vec3 o, vec3 d, float t_min, float t_max) const;
};

// instrumentation for the CPU replay: node visits / triangle tests of the calling THREAD (a counter shared by the OpenMP team put
// every traversal step of every thread on one cache line: the "parallel" replay of round 1 ran no faster than one core)
struct TraversalStats { uint64_t nodes = 0, tris = 0; };
TraversalStats& traversal_stats();

} // namespace orc
