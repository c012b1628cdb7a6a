// C-ABI glue: context, scene (BVH build + upload), raw ray queries, G-buffer synthesis.
#include "hr_internal.h"
#include <algorithm>
#include <atomic>
#include <memory>
#include <new>
#include "traverse.h"
#include <cstring>

using namespace hr;

namespace hr {
static thread_local std::string g_last_error;
void set_last_error(const std::string& s) { g_last_error = s; }
} // namespace hr

// ------------------------------------------------------------------------------------------------
// profiler ranges (hr_internal.h): roctx through dlopen — the library is only needed when somebody asks for markers
#include <dlfcn.h>
#include <cstdlib>
#include <mutex>
namespace hr {
namespace {
std::atomic<int> g_marker_mode { -1 };   // -1: not decided (HR_MARKERS), 0 off, 1 roctx, 2 in-process log
int  (*g_roctx_push)(const char*) = nullptr;
int  (*g_roctx_pop)() = nullptr;
std::mutex               g_marker_mutex;
std::vector<std::string> g_marker_log;   // mode 2: "+name" / "-" in call order, capped
int marker_mode()
{
    int m = g_marker_mode.load(std::memory_order_relaxed);
    if (m >= 0) return m;
    const char* e = getenv("HR_MARKERS");
    m = e ? atoi(e) : 0;
    if (m < 0 || m > 2) m = 0;
    g_marker_mode.store(m);
    return m;
}
bool roctx_ready()
{
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* lib : { "librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4" })
            if (void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL))
            {
                g_roctx_push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
                g_roctx_pop  = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (g_roctx_push && g_roctx_pop) return;
            }
        g_roctx_push = nullptr; g_roctx_pop = nullptr;
    });
    return g_roctx_push && g_roctx_pop;
}
} // namespace
bool samples_on() { return marker_mode() != 0; }
void sample_push(const char* name)
{
    const int m = marker_mode();
    if (m == 1) { if (roctx_ready()) (void)g_roctx_push(name); }
    else if (m == 2) { std::lock_guard<std::mutex> l(g_marker_mutex); if (g_marker_log.size() < 4096) g_marker_log.push_back(std::string("+") + name); }
}
void sample_pop()
{
    const int m = marker_mode();
    if (m == 1) { if (roctx_ready()) (void)g_roctx_pop(); }
    else if (m == 2) { std::lock_guard<std::mutex> l(g_marker_mutex); if (g_marker_log.size() < 4096) g_marker_log.push_back("-"); }
}
const char* sample_name_of_stage(const char* stage)
{
    static const struct { const char* stage; const char* label; } table[] = {
        { "ray_trace", "Ray Trace" }, { "temporal_accumulation", "Temporal Accumulation" }, { "upsample", "Upsample" },
        { "atrous_0", "Iteration 0" }, { "atrous_1", "Iteration 1" }, { "atrous_2", "Iteration 2" }, { "atrous_3", "Iteration 3" }, { "atrous_4", "Iteration 4" },
        { "atrous_01", "Iteration 0 + Iteration 1" },   // the tolerance mode runs the first two iterations in one launch
        { "blur_x", "Vertical" }, { "blur_y", "Horizontal" },   // the reference's labels: its "Vertical" pass blurs along (1, 0) (ray_traced_ao.cpp:1042,1066)
        { "blur_xy", "Vertical + Horizontal" },
        { "probe_update", "Irradiance + Depth + Border Update" },   // one launch for the three (ddgi.hip k_ddgi_probe_update)
        { "sample_probe_grid", "Sample Probe Grid" },
        { "path_trace", "Ground Truth Path Trace" }, { "taa", "TAA" },
    };
    for (const auto& t : table)
        if (std::strcmp(t.stage, stage) == 0) return t.label;
    return stage;
}
} // namespace hr

extern "C" hr_status hr_set_markers(int32_t mode)
{
    HR_CHECK_ARG(mode >= 0 && mode <= 2);
    hr::g_marker_mode.store(mode);
    std::lock_guard<std::mutex> l(hr::g_marker_mutex);
    hr::g_marker_log.clear();
    return HR_OK;
}
extern "C" int32_t hr_markers_log(char* out, int32_t capacity)
{
    std::lock_guard<std::mutex> l(hr::g_marker_mutex);
    std::string s;
    for (const std::string& e : hr::g_marker_log) { s += e; s += '\n'; }
    if (out && capacity > 0)
    {
        const size_t n = std::min(s.size(), (size_t)capacity - 1);
        std::memcpy(out, s.data(), n);
        out[n] = 0;
    }
    return (int32_t)s.size();
}

// ------------------------------------------------------------------------------------------------
// raw ray queries
__global__ __launch_bounds__(256) void k_any_hit_batch(const Node8* nodes, const TriGPU* tris, long long n, const float* rays, uint8_t* out, unsigned long long* stats)
{
    __shared__ uint32_t s_stack[4][HR_STACK_ENTRIES * 64];
    __shared__ CoopWave s_coop[4];
    const int       lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long i    = (long long)blockIdx.x * 256 + threadIdx.x;
    uint32_t        nn = 0, nt = 0;
    // the counting query walks per lane (trace_any); the plain one is the wave-cooperative walk the AO pass uses (trace_coop)
    float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
    if (i < n) { a = ((const float4*)rays)[i * 2]; b = ((const float4*)rays)[i * 2 + 1]; }
    if (stats)
    {
        if (i < n) out[i] = trace_any<true>(nodes, tris, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), b.w, a.w, s_stack[wave], lane, nn, nt) ? 1 : 0;
    }
    else
    {
        const bool occ = trace_coop<true>(i < n, nodes, tris, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), b.w, a.w, s_stack[wave], s_coop[wave], lane).prim == 0;
        if (i < n) out[i] = occ ? 1 : 0;
    }
    if (stats)
    {
        for (int o = 32; o > 0; o >>= 1) { nn += __shfl_down(nn, o); nt += __shfl_down(nt, o); }
        if (lane == 0)
        {
            atomicAdd(stats + 0, (unsigned long long)nn);
            atomicAdd(stats + 1, (unsigned long long)nt);
        }
    }
}

__global__ __launch_bounds__(256) void k_closest_hit_batch(const Node8* nodes, const TriGPU* tris, long long n, const float* rays, float* out_tuv, int32_t* out_prim)
{
    __shared__ uint32_t s_stack[4][HR_STACK_ENTRIES * 64];
    __shared__ CoopWave s_coop[4];
    const int       lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long i    = (long long)blockIdx.x * 256 + threadIdx.x;
    float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
    if (i < n) { a = ((const float4*)rays)[i * 2]; b = ((const float4*)rays)[i * 2 + 1]; }
    // the walk the DDGI and reflection passes use (trace_coop); lanes past the end of the batch only serve as triangle-test lanes
    const HitRec h = trace_coop<false>(i < n, nodes, tris, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), b.w, a.w, s_stack[wave], s_coop[wave], lane);
    if (i >= n) return;
    out_tuv[i * 3 + 0] = h.t;
    out_tuv[i * 3 + 1] = h.u;
    out_tuv[i * 3 + 2] = h.v;
    out_prim[i]        = h.prim;
}

// ------------------------------------------------------------------------------------------------
// G-buffer synthesis by primary rays.  Output conventions of g_buffer.frag:86-112 (see hr_api.h).
struct GBufArgs
{
    float           vpi[16], vp[16], pvp[16];
    float           cam[3];
    const Node8*    nodes;
    const TriGPU*   tris;
    const float*    positions;     // unused (vertices come from TriGPU)
    const float*    normals;       // [n][3][3] or null, indexed by original prim
    const uint32_t* tri_material;  // or null
    const uint32_t* tri_mesh_id;   // or null
    const float*    materials;     // [m][8] or null
    const float*    verts;         // [n][3][3] original positions by prim
    int             w, h;
    uint32_t*       gb1;
    uint2*          gb2;
    uint2*          gb3;
    float*          depth;
};

HR_DEV f3 gb_pixel_dir(const GBufArgs& a, float px, float py)
{
    f3 far_p = world_pos_from_depth(__fdiv_rn(px, (float)a.w), __fdiv_rn(py, (float)a.h), 1.0f, a.vpi);
    return normalize3(sub3(far_p, mk3(a.cam[0], a.cam[1], a.cam[2])));
}

HR_DEV f3 gb_normal_at(const GBufArgs& a, int prim, float b0, float b1, float b2)
{
    if (a.normals)
    {
        const float* n = a.normals + (size_t)prim * 9;
        return mk3(n[0] * b0 + n[3] * b1 + n[6] * b2, n[1] * b0 + n[4] * b1 + n[7] * b2, n[2] * b0 + n[5] * b1 + n[8] * b2);
    }
    const float* p = a.verts + (size_t)prim * 9;
    f3 v0 = mk3(p[0], p[1], p[2]), v1 = mk3(p[3], p[4], p[5]), v2 = mk3(p[6], p[7], p[8]);
    return normalize3(cross3(sub3(v1, v0), sub3(v2, v0)));
}

HR_DEV bool gb_plane_bary(const GBufArgs& a, int prim, f3 o, f3 d, float& b0, float& b1, float& b2)
{
    const float* p = a.verts + (size_t)prim * 9;
    f3 v0 = mk3(p[0], p[1], p[2]), v1 = mk3(p[3], p[4], p[5]), v2 = mk3(p[6], p[7], p[8]);
    f3 e1 = sub3(v1, v0), e2 = sub3(v2, v0);
    f3 n  = cross3(e1, e2);
    float dn = dot3(n, d);
    if (dn == 0.0f) return false;
    float tt = __fdiv_rn(dot3(n, sub3(v0, o)), dn);
    f3    pp = sub3(add3(o, scale3(d, tt)), v0);
    float d11 = dot3(e1, e1), d12 = dot3(e1, e2), d22 = dot3(e2, e2), p1 = dot3(pp, e1), p2 = dot3(pp, e2);
    float den = d11 * d22 - d12 * d12;
    if (den == 0.0f) return false;
    b1 = __fdiv_rn(d22 * p1 - d12 * p2, den);
    b2 = __fdiv_rn(d11 * p2 - d12 * p1, den);
    b0 = 1.0f - b1 - b2;
    return true;
}

__global__ __launch_bounds__(256) void k_gbuffer_raycast(GBufArgs a)
{
    __shared__ uint32_t s_stack[4][HR_STACK_ENTRIES * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // one wave = one 8x8 tile for ray coherence
    const int tiles_x = (a.w + 7) / 8, tiles_y = (a.h + 7) / 8;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= tiles_x * tiles_y) return;
    const int x = (tile % tiles_x) * 8 + (lane & 7), y = (tile / tiles_x) * 8 + (lane >> 3);
    if (x >= a.w || y >= a.h) return;
    const size_t i   = (size_t)y * a.w + x;
    const f3     cam = mk3(a.cam[0], a.cam[1], a.cam[2]);
    const f3     d   = gb_pixel_dir(a, (float)x + 0.5f, (float)y + 0.5f);
    HitRec       hit = trace_closest(a.nodes, a.tris, cam, d, 0.0f, 1.0e30f, s_stack[wave], lane);
    if (hit.prim < 0)
    {
        a.gb1[i]   = 0u;
        a.gb2[i]   = make_uint2(0u, 0u);
        a.gb3[i]   = make_uint2(0u, pack_h2(0.0f, -1.0f));
        a.depth[i] = 1.0f;
        return;
    }
    const f3 P     = add3(cam, scale3(d, hit.t));
    const f4 clip  = mul_m4(a.vp, P.x, P.y, P.z, 1.0f);
    const f4 pclip = mul_m4(a.pvp, P.x, P.y, P.z, 1.0f);
    const float b0 = 1.0f - hit.u - hit.v;
    const f3 nI    = gb_normal_at(a, hit.prim, b0, hit.u, hit.v);
    f3       n     = normalize3(nI);
    if (dot3(n, d) > 0.0f) n = neg3(n);
    float curvature = 0.0f;
    if (a.normals)
    {
        float c0, c1, c2;
        f3    dxv = mk3(0, 0, 0), dyv = mk3(0, 0, 0);
        if (gb_plane_bary(a, hit.prim, cam, gb_pixel_dir(a, (float)x + 1.5f, (float)y + 0.5f), c0, c1, c2)) dxv = sub3(gb_normal_at(a, hit.prim, c0, c1, c2), nI);
        if (gb_plane_bary(a, hit.prim, cam, gb_pixel_dir(a, (float)x + 0.5f, (float)y + 1.5f), c0, c1, c2)) dyv = sub3(gb_normal_at(a, hit.prim, c0, c1, c2), nI);
        curvature = hr_sqrt(max2(dot3(dxv, dxv), dot3(dyv, dyv)));
    }
    float ox, oy;
    oct_encode(n, ox, oy);
    const float cx = __fdiv_rn(clip.x, clip.w) * 0.5f + 0.5f, cy = __fdiv_rn(clip.y, clip.w) * 0.5f + 0.5f;
    const float px = __fdiv_rn(pclip.x, pclip.w) * 0.5f + 0.5f, py = __fdiv_rn(pclip.y, pclip.w) * 0.5f + 0.5f;
    const uint32_t mat = a.tri_material ? a.tri_material[hit.prim] : 0u;
    float albedo[3] = { 0.8f, 0.8f, 0.8f }, metallic = 0.0f, roughness = 0.5f;
    if (a.materials)
    {
        const float* m = a.materials + (size_t)mat * 8;
        albedo[0] = m[0]; albedo[1] = m[1]; albedo[2] = m[2]; metallic = m[3]; roughness = m[4];
    }
    uint32_t g1 = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) g1 |= (uint32_t)(clamp1(albedo[c], 0.0f, 1.0f) * 255.0f + 0.5f) << (8 * c);
    g1 |= (uint32_t)(clamp1(metallic, 0.0f, 1.0f) * 255.0f + 0.5f) << 24;
    a.gb1[i] = g1;
    a.gb2[i] = make_uint2(pack_h2(ox, oy), pack_h2(px - cx, py - cy));
    const float mesh_id = a.tri_mesh_id ? (float)a.tri_mesh_id[hit.prim] : 0.0f;
    a.gb3[i] = make_uint2(pack_h2(max2(roughness, 0.1f), curvature), pack_h2(mesh_id, clip.z));
    const float dd = __fdiv_rn(clip.z, clip.w);
    a.depth[i] = dd >= 1.0f ? 0.99999994f : dd;
}

__global__ void k_selftest_math(int which, long long n, const float* in, float* out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = in[i * 3], y = in[i * 3 + 1], z = in[i * 3 + 2];
    float       r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
    switch (which)
    {
        case 0: det_sincos(x, r0, r1); break;
        case 1: r0 = det_exp(x); break;
        case 2: r0 = det_log(x); break;
        case 3: r0 = det_pow_auto(x, y); break;
        case 4: r0 = (float)f2h(x); r1 = h2f(f2h(x)); break;
        case 5: { f3 v = oct_decode(x, y); r0 = v.x; r1 = v.y; r2 = v.z; break; }
        case 6: oct_encode(mk3(x, y, z), r0, r1); break;
        default: break;
    }
    out[i * 3] = r0; out[i * 3 + 1] = r1; out[i * 3 + 2] = r2;
}

// ------------------------------------------------------------------------------------------------
extern "C" {

hr_status hr_selftest_math(int32_t which, int64_t n, const float* in, float* out, void* stream)
{
    HR_CHECK_ARG(n >= 0 && (n == 0 || (in && out)));
    if (n == 0) return HR_OK;
    hipLaunchKernelGGL(k_selftest_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (int)which, (long long)n, in, out);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

const char* hr_status_string(hr_status s)
{
    switch (s)
    {
        case HR_OK: return "HR_OK";
        case HR_ERR_INVALID_ARG: return "HR_ERR_INVALID_ARG";
        case HR_ERR_HIP: return "HR_ERR_HIP";
        case HR_ERR_NO_DEVICE: return "HR_ERR_NO_DEVICE";
        case HR_ERR_OUT_OF_MEMORY: return "HR_ERR_OUT_OF_MEMORY";
        case HR_ERR_UNSUPPORTED: return "HR_ERR_UNSUPPORTED";
        case HR_ERR_TIMEOUT: return "HR_ERR_TIMEOUT";
        case HR_ERR_COMM: return "HR_ERR_COMM";
        default: return "HR_ERR_UNKNOWN";
    }
}
const char* hr_last_error(void) { return g_last_error.c_str(); }
#ifdef HR_DEV_PATHS
const char* hr_version(void) { return "hybrid_rendering_amd 0.4 (gfx950) +dev"; }   // built with the A/B paths that lost (HR_CFLAGS=-DHR_DEV_PATHS)
#else
const char* hr_version(void) { return "hybrid_rendering_amd 0.4 (gfx950)"; }
#endif
int32_t hr_api_revision(void) { return HR_API_REVISION; }

hr_status hr_ctx_create(int device_ordinal, hr_ctx** out)
{
    HR_CHECK_ARG(out);
    int        n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
    {
        set_last_error(std::string("no HIP device: ") + hipGetErrorString(e));
        return HR_ERR_NO_DEVICE;
    }
    HR_CHECK_ARG(device_ordinal >= 0 && device_ordinal < n);
    HR_HIP(hipSetDevice(device_ordinal));
    hr_ctx* c = new (std::nothrow) hr_ctx();
    if (!c) return HR_ERR_OUT_OF_MEMORY;
    c->device = device_ordinal;
    e = hipGetDeviceProperties(&c->props, device_ordinal);
    if (e != hipSuccess)
    {
        set_last_error(std::string("hipGetDeviceProperties failed: ") + hipGetErrorString(e));
        delete c;
        return HR_ERR_HIP;
    }
    *out = c;
    return HR_OK;
}

hr_status hr_ctx_destroy(hr_ctx* ctx)
{
    delete ctx;
    return HR_OK;
}

int32_t hr_ctx_device(const hr_ctx* ctx) { return ctx ? ctx->device : -1; }

static hr_status scene_create_impl(hr_ctx* ctx, const hr_scene_desc* d, hr_scene** out);

// Host-only: build the 8-wide BVH of a triangle soup and report its shape (no device, no upload).  What hr_scene_create
// would build for the same positions — lets an integrator (and the CPU test-suite) check depth / size limits up front.
hr_status hr_bvh_build_info(const float* positions, int32_t n_tris, hr_scene_info* info)
{
    HR_CHECK_ARG(info && n_tris >= 0 && (positions || n_tris == 0));
    try
    {
        BuiltBVH b;
        build_bvh8(positions, n_tris, b);
        std::memset(info, 0, sizeof(*info));
        info->n_tris     = n_tris;
        info->n_nodes    = (int32_t)b.nodes.size();
        info->max_depth  = b.max_depth;
        info->node_bytes = b.nodes.size() * sizeof(Node8);
        info->tri_bytes  = b.tris.size() * sizeof(TriGPU);
        info->box_pad    = b.pad;
        for (int a = 0; a < 3; a++) { info->bounds_lo[a] = b.lo[a]; info->bounds_hi[a] = b.hi[a]; }
        return (b.nodes.size() >= (1u << 23) || b.max_depth >= kMaxTraversalDepth) ? HR_ERR_UNSUPPORTED : HR_OK;
    }
    catch (const std::bad_alloc&)
    {
        set_last_error("hr_bvh_build_info: host allocation failed");
        return HR_ERR_OUT_OF_MEMORY;
    }
    catch (const std::exception& e)   // nothing else is expected; no exception may cross the C ABI
    {
        set_last_error(std::string("hr_bvh_build_info: ") + e.what());
        return HR_ERR_UNSUPPORTED;
    }
}

// Host-only: builds the same BVH and checks that every triangle is found from every point of its surface (bvh.h
// check_bvh8_coverage) — the invariant the spatial splits of the builder have to keep.
hr_status hr_bvh_selfcheck(const float* positions, int32_t n_tris, int32_t samples_per_triangle, int64_t* uncovered)
{
    HR_CHECK_ARG(uncovered && n_tris >= 0 && samples_per_triangle > 0 && (positions || n_tris == 0));
    try
    {
        BuiltBVH b;
        build_bvh8(positions, n_tris, b);
        *uncovered = check_bvh8_coverage(positions, n_tris, b, samples_per_triangle);
        return HR_OK;
    }
    catch (const std::bad_alloc&)
    {
        set_last_error("hr_bvh_selfcheck: host allocation failed");
        return HR_ERR_OUT_OF_MEMORY;
    }
    catch (const std::exception& e)
    {
        set_last_error(std::string("hr_bvh_selfcheck: ") + e.what());
        return HR_ERR_UNSUPPORTED;
    }
}

// No exception crosses the C ABI: the builder's and the staging vectors' allocation failures become HR_ERR_OUT_OF_MEMORY.
hr_status hr_scene_create(hr_ctx* ctx, const hr_scene_desc* d, hr_scene** out)
{
    try
    {
        return scene_create_impl(ctx, d, out);
    }
    catch (const std::bad_alloc&)
    {
        set_last_error("hr_scene_create: host allocation failed");
        return HR_ERR_OUT_OF_MEMORY;
    }
    catch (const std::exception& e)
    {
        set_last_error(std::string("hr_scene_create: ") + e.what());
        return HR_ERR_UNSUPPORTED;
    }
}

static hr_status scene_create_impl(hr_ctx* ctx, const hr_scene_desc* d, hr_scene** out)
{
    HR_CHECK_ARG(ctx && d && out && d->n_tris >= 0 && (d->positions || d->n_tris == 0));
    HR_CHECK_ARG(d->n_materials >= 0 && (d->materials || d->n_materials == 0));
    // every triangle's material index is dereferenced by the hit shading (shading.h surface_at: materials[m * 8], mat_tex[m * 6])
    if (d->tri_material)
    {
        if (!d->materials) { set_last_error("hr_scene_create: tri_material given without materials"); return HR_ERR_INVALID_ARG; }
        for (int i = 0; i < d->n_tris; i++)
            if (d->tri_material[i] >= (uint32_t)d->n_materials)
            {
                set_last_error("hr_scene_create: tri_material[" + std::to_string(i) + "] = " + std::to_string(d->tri_material[i]) + " >= n_materials");
                return HR_ERR_INVALID_ARG;
            }
    }
    HR_HIP(hipSetDevice(ctx->device));
    BuiltBVH b;
    build_bvh8(d->positions, d->n_tris, b);
    if (b.nodes.size() >= (1u << 23))   // traversal stack entries hold child_base in 23 bits
    {
        set_last_error("hr_scene_create: more than 2^23 BVH nodes");
        return HR_ERR_UNSUPPORTED;
    }
    if (b.tris.size() >= kCoopMaxTriangles)   // cooperative triangle jobs hold the triangle reference in 26 bits (traverse.h CoopWave)
    {
        set_last_error("hr_scene_create: more than 2^26 triangle references");
        return HR_ERR_UNSUPPORTED;
    }
    if (b.max_depth >= kMaxTraversalDepth)   // one stack entry per level (traverse.h); the builder's depth cap keeps real input below it
    {
        set_last_error("hr_scene_create: BVH depth " + std::to_string(b.max_depth) + " exceeds the traversal stack (" + std::to_string(kMaxTraversalDepth) + ")");
        return HR_ERR_UNSUPPORTED;
    }
    std::unique_ptr<hr_scene> guard(new hr_scene());
    hr_scene* s = guard.get();
    s->ctx      = ctx;
    hr_status st;
#define UP(buf, src, nbytes)                                                                     \
    if ((st = s->buf.alloc(nbytes)) != HR_OK) return st;                                         \
    if ((nbytes) > 0) { hipError_t e_ = hipMemcpy(s->buf.p, src, nbytes, hipMemcpyHostToDevice); \
        if (e_ != hipSuccess) { set_last_error(std::string("hipMemcpy H2D failed: ") + hipGetErrorString(e_)); return HR_ERR_HIP; } }
    UP(nodes, b.nodes.data(), b.nodes.size() * sizeof(Node8))
    UP(tris, b.tris.data(), b.tris.size() * sizeof(TriGPU))
    const size_t n = (size_t)d->n_tris;
    // original-order vertex positions are kept for shading-time interpolation
    UP(materials, d->materials, d->materials ? (size_t)d->n_materials * 32 : 0)
    if (d->normals) { UP(tri_normals, d->normals, n * 36) s->has_normals = true; }
    if (d->tri_material) { UP(tri_material, d->tri_material, n * 4) s->has_material = true; }
    if (d->tri_mesh_id) { UP(tri_mesh_id, d->tri_mesh_id, n * 4) s->has_mesh_id = true; }
    UP(positions, d->positions, n * 36)
    if (d->material_textures && d->materials && d->n_textures > 0 && d->textures)
    {
        // one buffer of texels + a table {texel offset, width, height, 0} per texture
        std::vector<uint32_t> table;
        std::vector<uint8_t>  texels;
        for (int i = 0; i < d->n_textures; i++)
        {
            const hr_texture& t = d->textures[i];
            if (!t.rgba8 || t.width <= 0 || t.height <= 0) { set_last_error("hr_scene_create: empty texture"); return HR_ERR_INVALID_ARG; }
            table.insert(table.end(), { (uint32_t)(texels.size() / 4), (uint32_t)t.width, (uint32_t)t.height, 0u });
            texels.insert(texels.end(), t.rgba8, t.rgba8 + (size_t)t.width * t.height * 4);
        }
        for (int i = 0; i < d->n_materials * 4; i++)
        {
            const int32_t ti = d->material_textures[(i / 4) * 6 + (i % 4)];
            if (ti >= d->n_textures) { set_last_error("hr_scene_create: material texture index out of range"); return HR_ERR_INVALID_ARG; }
        }
        UP(mat_tex, d->material_textures, (size_t)d->n_materials * 24)
        UP(tex_table, table.data(), table.size() * 4)
        UP(tex_data, texels.data(), texels.size())
        if (d->uvs) { UP(tri_uvs, d->uvs, n * 24) s->has_uvs = true; }
        if (d->tangents) { UP(tri_tangents, d->tangents, n * 36) s->has_tangents = true; }
        s->has_textures = true;
    }
#undef UP
    s->n_materials      = d->materials ? d->n_materials : 0;
    { static std::atomic<uint64_t> next_uid { 1 }; s->uid = next_uid.fetch_add(1); }
    s->info.n_tris      = d->n_tris;
    s->info.n_nodes     = (int32_t)b.nodes.size();
    s->info.max_depth   = b.max_depth;
    s->info.node_bytes  = b.nodes.size() * sizeof(Node8);
    s->info.tri_bytes   = b.tris.size() * sizeof(TriGPU);
    s->info.box_pad     = b.pad;
    for (int a = 0; a < 3; a++) { s->info.bounds_lo[a] = s->grid_lo[a] = b.lo[a]; s->info.bounds_hi[a] = s->grid_hi[a] = b.hi[a]; }
    *out = guard.release();
    return HR_OK;
}

// nearest-filtered mip level of the four G-buffer images (g_buffer.cpp:240-243: vkCmdBlitImage, VK_FILTER_NEAREST):
// destination texel (x, y) = source texel (x << level, y << level)
__global__ __launch_bounds__(256) void k_gbuffer_mip(hr_gbuffer_level src, hr_gbuffer_level dst, int level)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= dst.width || y >= dst.height) return;
    const size_t so = (size_t)(y << level) * src.width + (x << level), o = (size_t)y * dst.width + x;
    if (src.gb1 && dst.gb1) ((uint32_t*)dst.gb1)[o] = ((const uint32_t*)src.gb1)[so];
    ((uint2*)dst.gb2)[o] = ((const uint2*)src.gb2)[so];
    ((uint2*)dst.gb3)[o] = ((const uint2*)src.gb3)[so];
    ((float*)dst.depth)[o] = src.depth[so];
}

extern "C" hr_status hr_gbuffer_mip_nearest(const hr_gbuffer_level* src, const hr_gbuffer_level* dst, int32_t level, void* stream)
{
    HR_CHECK_ARG(src && dst && level >= 1 && level <= 8 && src->gb2 && src->gb3 && src->depth && dst->gb2 && dst->gb3 && dst->depth);
    HR_CHECK_ARG(dst->width == (src->width >> level) && dst->height == (src->height >> level) && dst->width > 0 && dst->height > 0);
    hipLaunchKernelGGL(k_gbuffer_mip, dim3(cdiv(dst->width, 32), cdiv(dst->height, 8)), dim3(256), 0, (hipStream_t)stream, *src, *dst, (int)level);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

hr_status hr_scene_get_info(const hr_scene* scene, hr_scene_info* info)
{
    HR_CHECK_ARG(scene && info);
    if (scene->n_instances > 0)
    {
        const hr_status s = instanced_scene_refresh_bounds(scene);   // the exact bounds of the last hr_scene_update_instances, read back on demand
        if (s != HR_OK) return s;
    }
    *info = scene->info;
    return HR_OK;
}

uint64_t hr_scene_id(const hr_scene* scene) { return scene ? scene->uid : 0; }

hr_status hr_scene_read_bvh(const hr_scene* scene, void* nodes_out, void* tris_out)
{
    HR_CHECK_ARG(scene);
    HR_HIP(hipSetDevice(scene->ctx->device));
    HR_HIP(hipDeviceSynchronize());
    if (nodes_out) HR_HIP(hipMemcpy(nodes_out, scene->nodes.p, (size_t)scene->info.node_bytes, hipMemcpyDeviceToHost));
    if (tris_out) HR_HIP(hipMemcpy(tris_out, scene->tris.p, (size_t)scene->info.tri_bytes, hipMemcpyDeviceToHost));
    return HR_OK;
}

hr_status hr_scene_destroy(hr_scene* scene)
{
    if (scene)
    {
        (void)hipSetDevice(scene->ctx->device);
        (void)hipDeviceSynchronize();
        delete scene;
    }
    return HR_OK;
}

hr_status hr_trace_any_hit(const hr_scene* scene, int64_t n, const float* rays, uint8_t* out, uint64_t* stats, void* stream)
{
    HR_CHECK_ARG(scene && n >= 0 && (n == 0 || (rays && out)));
    if (n == 0) return HR_OK;
    hipLaunchKernelGGL(k_any_hit_batch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const Node8*)scene->nodes.p,
                       (const TriGPU*)scene->tris.p, (long long)n, rays, out, (unsigned long long*)stats);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

hr_status hr_trace_closest_hit(const hr_scene* scene, int64_t n, const float* rays, float* out_tuv, int32_t* out_prim, void* stream)
{
    HR_CHECK_ARG(scene && n >= 0 && (n == 0 || (rays && out_tuv && out_prim)));
    if (n == 0) return HR_OK;
    hipLaunchKernelGGL(k_closest_hit_batch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const Node8*)scene->nodes.p,
                       (const TriGPU*)scene->tris.p, (long long)n, rays, out_tuv, out_prim);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

hr_status hr_gbuffer_raycast(const hr_scene* scene, const hr_ubo* ubo, int32_t w, int32_t h, void* gb1, void* gb2, void* gb3, float* depth, void* stream)
{
    HR_CHECK_ARG(scene && ubo && w > 0 && h > 0 && gb1 && gb2 && gb3 && depth);
    GBufArgs a;
    for (int i = 0; i < 16; i++) { a.vpi[i] = ubo->view_proj_inverse[i]; a.vp[i] = ubo->view_proj[i]; a.pvp[i] = ubo->prev_view_proj[i]; }
    for (int i = 0; i < 3; i++) a.cam[i] = ubo->cam_pos[i];
    a.nodes = (const Node8*)scene->nodes.p; a.tris = (const TriGPU*)scene->tris.p;
    a.positions = nullptr;
    a.normals = scene->has_normals ? (const float*)scene->tri_normals.p : nullptr;
    a.tri_material = scene->has_material ? (const uint32_t*)scene->tri_material.p : nullptr;
    a.tri_mesh_id = scene->has_mesh_id ? (const uint32_t*)scene->tri_mesh_id.p : nullptr;
    a.materials = scene->n_materials ? (const float*)scene->materials.p : nullptr;
    a.verts = (const float*)scene->positions.p;
    a.w = w; a.h = h;
    a.gb1 = (uint32_t*)gb1; a.gb2 = (uint2*)gb2; a.gb3 = (uint2*)gb3; a.depth = depth;
    const int tiles = ((w + 7) / 8) * ((h + 7) / 8);
    hipLaunchKernelGGL(k_gbuffer_raycast, dim3((tiles + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

} // extern "C"
