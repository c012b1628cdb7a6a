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

// Profiler ranges under the REFERENCE's sample names (DW_SCOPED_SAMPLE: ray_traced_shadows.cpp:102,974,1043,1096,1147,1221, ray_traced_ao.cpp:100,
// 865,985,1034,1042,1091, ray_traced_reflections.cpp:109,999,1089,1145,1188,1262, ddgi.cpp:91,769,831,864,906,945, ...), so that a
// `rocprofv3 --marker-trace` of an integrated frame reads like the reference's profiler tree.  Off unless hr_set_markers() / HR_MARKERS asks:
// 1 = roctx (librocprofiler-sdk-roctx.so, dlopen'ed on first use), 2 = an in-process log (tests).  api.hip.
void        sample_push(const char* name);
void        sample_pop();
const char* sample_name_of_stage(const char* stage);   // StageProfiler stage -> the reference's label
bool        samples_on();
struct ScopedSample
{
    bool on;
    explicit ScopedSample(const char* name) : on(samples_on()) { if (on) sample_push(name); }
    ~ScopedSample() { if (on) sample_pop(); }
    ScopedSample(const ScopedSample&) = delete;
    ScopedSample& operator=(const ScopedSample&) = delete;
};
#define HR_SCOPED_SAMPLE(name) ::hr::ScopedSample hr_scoped_sample_(name)

// is per-stage event profiling switched on for this pass? (frame.hip: timing events cannot be captured into a hipGraph)
bool profiling_enabled(const hr_shadows* p);
bool profiling_enabled(const hr_ao* p);
bool profiling_enabled(const hr_ddgi* p);
bool profiling_enabled(const hr_reflections* p);
// instances.hip: brings info.bounds_* of an instanced scene up to date with its last update (synchronises the device when they lag)
hr_status instanced_scene_refresh_bounds(const hr_scene* scene);

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
        // Every pass image starts as zeros: rows a pass never writes (a band's history apron before the first exchange, the other
        // half of a ping-pong pair) then read the same on every run and in every object, instead of whatever the allocator recycled.
        e = hipMemset(p, 0, n);
        if (e != hipSuccess)
        {
            set_last_error(std::string("hipMemset(") + std::to_string(n) + ") failed: " + hipGetErrorString(e));
            (void)hipFree(p);
            p = nullptr;
            return HR_ERR_HIP;
        }
        // hipMemset on device memory returns before the fill has run (it is queued on the NULL stream); a caller that renders straight away on
        // a hipStreamNonBlocking stream is not ordered after it.  Creation is not a hot path: wait here, so that "created" means "zeroed"
        // (hr_api.h: every hr_*_create returns with its images initialised)
        e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess)
        {
            set_last_error(std::string("hipStreamSynchronize after hipMemset failed: ") + hipGetErrorString(e));
            (void)hipFree(p);
            p = nullptr;
            return HR_ERR_HIP;
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
// Per-stage HIP-event timing on the launch stream.  Every stage keeps a RING of event pairs, one pair per profiled frame,
// so that a caller can leave profiling on over a whole timed region and read the per-kernel AVERAGE afterwards
// (bench.py: roofline.achieved) without synchronising inside it.  A new frame starts when begin_frame() is called or when a
// stage is begun for the second time in the current frame (stage-level callers).  collect() averages every frame recorded
// since the previous collect() and starts over.
struct StageProfiler
{
    static constexpr int kRing = 512;
    struct Stage
    {
        std::string             name;
        std::vector<hipEvent_t> ev0, ev1;   // created lazily, up to kRing pairs
        std::vector<long long>  stamp;      // epoch * 2^32 + frame of the recording in each slot
        uint64_t                bytes = 0;
        int                     last_frame = -1;   // last frame index this stage was recorded in
    };
    bool               enabled = false;
    std::vector<Stage> stages;
    int                frame = 0;        // frames started since the last collect()
    long long          epoch = 0;        // collect() calls so far
    bool               frame_open = false;

    int find_or_add(const char* name)
    {
        for (size_t i = 0; i < stages.size(); i++)
            if (stages[i].name == name) return (int)i;
        if (stages.size() >= HR_MAX_STAGES) return -1;
        stages.emplace_back();
        stages.back().name = name;
        return (int)stages.size() - 1;
    }
    void begin_frame()
    {
        if (frame_open) frame++;
        frame_open = false;
    }
    int  marker_depth = 0;           // ranges begin() opened and end() still has to close
    int begin(const char* name, hipStream_t s, uint64_t algorithmic_bytes)
    {
        if (samples_on()) { sample_push(sample_name_of_stage(name)); marker_depth++; }
        if (!enabled) return -1;
        int i = find_or_add(name);
        if (i < 0) return -1;
        Stage& st = stages[i];
        if (frame_open && st.last_frame == frame) frame++;   // the stage repeats: a new frame has begun
        frame_open = true;
        const int slot = frame % kRing;
        while ((int)st.ev0.size() <= slot)
        {
            hipEvent_t a, b;
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            st.ev0.push_back(a);
            st.ev1.push_back(b);
            st.stamp.push_back(-1);
        }
        st.stamp[slot] = (epoch << 32) + frame;
        st.bytes      = algorithmic_bytes;
        st.last_frame = frame;
        (void)hipEventRecord(st.ev0[slot], s);
        return i;
    }
    void end(int i, hipStream_t s)
    {
        if (marker_depth > 0) { sample_pop(); marker_depth--; }
        if (i >= 0) (void)hipEventRecord(stages[i].ev1[stages[i].last_frame % kRing], s);
    }
    void collect(hr_stage_times* out)
    {
        out->n_stages = 0;
        const int n_frames = frame + (frame_open ? 1 : 0);
        const int first    = n_frames > kRing ? n_frames - kRing : 0;   // older slots were overwritten
        for (Stage& st : stages)
        {
            double sum = 0.0;
            int    cnt = 0;
            for (int f = first; f < n_frames; f++)
            {
                const int slot = f % kRing;
                if (slot >= (int)st.ev0.size()) continue;
                if (st.stamp[slot] != (epoch << 32) + f) continue;   // this stage did not run in frame f
                float ms = 0.0f;
                if (hipEventSynchronize(st.ev1[slot]) != hipSuccess) continue;
                if (hipEventElapsedTime(&ms, st.ev0[slot], st.ev1[slot]) != hipSuccess) continue;
                sum += ms;
                cnt++;
            }
            if (st.last_frame < 0 || cnt == 0) continue;
            const int k     = out->n_stages++;
            out->name[k]    = st.name.c_str();
            out->ms[k]      = (float)(sum / cnt);
            out->bytes[k]   = st.bytes;
            st.last_frame   = -1;
        }
        frame = 0;
        frame_open = false;
        epoch++;
    }
    ~StageProfiler()
    {
        for (Stage& st : stages)
        {
            for (auto e : st.ev0) (void)hipEventDestroy(e);
            for (auto e : st.ev1) (void)hipEventDestroy(e);
        }
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
    hr::DevBuf    tri_uvs, tri_tangents, mat_tex, tex_table, tex_data;          // textured materials (optional)
    bool          has_uvs = false, has_tangents = false, has_textures = false;
    hr_scene_info info;
    uint64_t      uid = 0;   // unique per hr_scene_create (a destroyed scene's device addresses may be handed out again): key of per-scene caches in the passes
    int           n_materials = 0;
    bool          has_normals = false, has_material = false, has_mesh_id = false;
    // ---- instanced scenes (instances.hip); n_instances == 0: a flattened scene from hr_scene_create.  `nodes` / `tris` / `positions` /
    // `tri_normals` are then REWRITTEN by hr_scene_update_instances (world space); the mesh_* arrays hold the object-space attributes the
    // hit shading interpolates (meshes concatenated), tri_instance / inst_records map a global triangle to them
    int           n_instances = 0;
    uint64_t      geometry_epoch = 0;   // bumped by every update: with `uid` the key of geometry-dependent caches of the passes (AO entry table)
    hr::DevBuf    inst_records, tri_instance, mesh_positions, mesh_normals, mesh_uvs, mesh_tangents, mesh_material;
    hr::DevBuf    level_nodes, node_box, bounds_bits, leaf_cells, node_inst, inst_dirty_dev;
    // the top level over the instance roots lives in the first `top_cap` node slots and can be RE-BUILT (host SAH over the instances' boxes, then
    // the slots, the level lists and everything above the subtrees are uploaded again): what a fixed top level loses after long motion
    int                   top_cap = 1;            // node slots reserved for top-level nodes + instance roots (2 x instances)
    std::vector<hr::Node8> inst_root_node;        // per instance: its subtree's root (topology fields final), wherever the top level puts it
    std::vector<float>    inst_root_cells;        // per instance: the 48 leaf-cell floats of that root
    std::vector<int32_t>  node_inst_host, node_rel_depth;   // per node: owning instance (-1 top level) / depth below the instance root
    std::vector<float>    inst_box;               // per instance: conservative world box (lo xyz, hi xyz) of the last update, host side
    std::vector<int32_t>  top_parent, top_first_child;      // host mirror of the top level (slot -> parent slot / first child slot) for its quality check
    std::vector<hr::Node8> top_nodes_host;        // upload staging, kept alive for the asynchronous copies
    std::vector<uint32_t> level_nodes_host;
    std::vector<int32_t>  inst_root_slot;
    std::vector<float>    top_cells_host;
    int           top_used = 1;
    bool          auto_rebuild = true;            // HR_TOP_LEVEL_REBUILD=0 switches the automatic re-build off (developer A/B)
    double        rebuild_ratio = 1.5;
    double        top_area_at_build = 0.0;        // sum of the top-level nodes' half areas when it was built: the re-build trigger compares against it
    int           top_rebuilds = 0;
    std::vector<uint32_t> inst_dirty;             // per instance: its matrix changed in the update being enqueued
    std::vector<int32_t>  level_offsets;          // level_nodes[level_offsets[d] .. level_offsets[d + 1]): the nodes of depth d
    std::vector<uint32_t> inst_mesh;              // per instance: mesh index
    std::vector<hr::InstanceRec> inst_host;       // host copy of inst_records (matrices of the last update)
    std::vector<float>    mesh_bounds;            // per mesh: object-space lo xyz, hi xyz
    float         grid_lo[3] = { 0, 0, 0 }, grid_hi[3] = { 0, 0, 0 };   // bounds a pass may read WITHOUT synchronising: exact for a flattened scene, the
                                                                         // host's conservative bounds (transformed mesh boxes) for an instanced one
    mutable bool  bounds_stale = false;           // info.bounds_* lag the last update until hr_scene_get_info reads them back
};
