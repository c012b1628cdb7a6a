// Internal host-side plumbing shared by the pass implementations: error mapping, device buffers,
// per-stage hipEvent profiling (the DW_SCOPED_SAMPLE analogue, e.g. ray_traced_shadows.cpp:102,974).
#pragma once
#include "../../include/hr_api.h"
#include "bvh.h"
#include <hip/hip_runtime.h>
#include <string>
#include <vector>

namespace hr {

void set_last_error(const std::string& s);

#define HR_HIP(expr)                                                                                              \
    do                                                                                                            \
    {                                                                                                             \
        hipError_t _e = (expr);                                                                                   \
        if (_e != hipSuccess)                                                                                     \
        {                                                                                                         \
            ::hr::set_last_error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")"); \
            return HR_ERR_HIP;                                                                                    \
        }                                                                                                         \
    } while (0)

#define HR_CHECK_ARG(cond)                                                                 \
    do                                                                                     \
    {                                                                                      \
        if (!(cond))                                                                       \
        {                                                                                  \
            ::hr::set_last_error(std::string("invalid argument: ") + #cond);               \
            return HR_ERR_INVALID_ARG;                                                     \
        }                                                                                  \
    } while (0)

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

struct DevBuf
{
    void*  p     = nullptr;
    size_t bytes = 0;
    hr_status alloc(size_t n)
    {
        free();
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess)
        {
            set_last_error(std::string("hipMalloc(") + std::to_string(n) + ") failed: " + hipGetErrorString(e));
            p = nullptr;
            return e == hipErrorOutOfMemory ? HR_ERR_OUT_OF_MEMORY : HR_ERR_HIP;
        }
        bytes = n;
        return HR_OK;
    }
    void free()
    {
        if (p) (void)hipFree(p);
        p     = nullptr;
        bytes = 0;
    }
    ~DevBuf() { free(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

// Stage profiler: one event pair per named stage, recorded on the pass stream.
struct StageProfiler
{
    bool                     enabled = false;
    std::vector<std::string> names;
    std::vector<hipEvent_t>  ev0, ev1;
    std::vector<uint64_t>    bytes;
    std::vector<char>        used;
    int                      find_or_add(const char* name)
    {
        for (size_t i = 0; i < names.size(); i++)
            if (names[i] == name) return (int)i;
        if (names.size() >= HR_MAX_STAGES) return -1;
        names.push_back(name);
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        ev0.push_back(a);
        ev1.push_back(b);
        bytes.push_back(0);
        used.push_back(0);
        return (int)names.size() - 1;
    }
    void begin_frame()
    {
        for (auto& u : used) u = 0;
    }
    int begin(const char* name, hipStream_t s, uint64_t algorithmic_bytes)
    {
        if (!enabled) return -1;
        int i = find_or_add(name);
        if (i < 0) return -1;
        bytes[i] = algorithmic_bytes;
        used[i]  = 1;
        (void)hipEventRecord(ev0[i], s);
        return i;
    }
    void end(int i, hipStream_t s)
    {
        if (i >= 0) (void)hipEventRecord(ev1[i], s);
    }
    void collect(hr_stage_times* out)
    {
        out->n_stages = 0;
        for (size_t i = 0; i < names.size(); i++)
        {
            if (!used[i]) continue;
            float ms = 0.0f;
            (void)hipEventSynchronize(ev1[i]);
            (void)hipEventElapsedTime(&ms, ev0[i], ev1[i]);
            int k        = out->n_stages++;
            out->name[k] = names[i].c_str();
            out->ms[k]   = ms;
            out->bytes[k] = bytes[i];
        }
    }
    ~StageProfiler()
    {
        for (auto e : ev0) (void)hipEventDestroy(e);
        for (auto e : ev1) (void)hipEventDestroy(e);
    }
};

} // namespace hr

struct hr_ctx
{
    int            device = 0;
    hipDeviceProp_t props;
};

struct hr_scene
{
    hr_ctx*       ctx = nullptr;
    hr::DevBuf    nodes, tris;
    hr::DevBuf    tri_normals, tri_material, tri_mesh_id, materials, positions; // shading data (by original triangle index)
    hr_scene_info info;
    int           n_materials = 0;
    bool          has_normals = false, has_material = false, has_mesh_id = false;
};
