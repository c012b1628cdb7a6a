// C++ shims of the row-tiled (multi-GPU) frame over include/hr_comm.h + include/hr/passes.hpp — the native counterpart of
// hybrid_rendering_amd/tiling.py.  One hr::Tiled* object per GPU and pass: it owns the pass of ONE row band and the
// per-frame neighbour exchange of the history rows the NEXT frame's reprojection reads.  render() keeps the reference's arity.
//
//   hr::Comm comm(ctx, world, rank, id);                         // RCCL (or hr::Comm::loopback for several ranks in one process)
//   hr::TiledShadows shadows(ctx, comm, &common, &g_buffer, bounds);
//   shadows.render(cmd_buf);                                     // trace -> [wait last exchange] -> temporal -> a-trous -> post exchange
//
// Redundant compute instead of per-iteration halo traffic: every band traces / filters `halo` = 24 rows beyond its own (8 mask rows of
// the 17x17 statistics + 1 + 2 + 4 + 8 a-trous rows), and ONE grouped neighbour exchange per frame refreshes the `history_halo` rows of
// the images the next frame reprojects from.  The alternative SURVEY §8e names — exchanging 2^i + 1 rows of every a-trous iteration —
// needs five dependent exchanges per frame for a 4 us saving of redundant work per band at 4K / 8 GPUs.
#pragma once
#include "../hr_comm.h"
#include "passes.hpp"
#include <stdexcept>
#include <string>
#include <vector>

namespace hr {

constexpr int kHalo = 24, kHistoryHalo = 40;

class Comm
{
public:
    Comm(Context& ctx, int world, int rank, const uint8_t id[HR_COMM_ID_BYTES]) { check(hr_comm_create_rccl(ctx.handle(), world, rank, id, &m_comm), "hr_comm_create_rccl"); }
    static Comm loopback(Context& ctx, int world, int rank, const char* name)
    {
        Comm c;
        check(hr_comm_create_loopback(ctx.handle(), world, rank, name, &c.m_comm), "hr_comm_create_loopback");
        return c;
    }
    Comm(Comm&& o) noexcept : m_comm(o.m_comm) { o.m_comm = nullptr; }
    Comm(const Comm&) = delete;
    Comm& operator=(const Comm&) = delete;
    ~Comm() { hr_comm_destroy(m_comm); }
    int      rank() const { return hr_comm_rank(m_comm); }
    int      world() const { return hr_comm_world(m_comm); }
    void     wait(Stream s) { check(hr_comm_wait(m_comm, s), "hr_comm_wait"); }                                   // everything posted so far
    void     wait(hr_comm_ticket t, Stream s) { check(hr_comm_wait_ticket(m_comm, t, s), "hr_comm_wait_ticket"); }   // up to ticket t only
    hr_comm* handle() const { return m_comm; }
    static void unique_id(uint8_t id[HR_COMM_ID_BYTES]) { check(hr_comm_get_unique_id(id), "hr_comm_get_unique_id"); }
private:
    Comm() = default;
    hr_comm* m_comm = nullptr;
};

// uniform band boundaries on the 8-row tile grid (tiling.band_rows); cost-balanced boundaries come from the application
inline std::vector<int32_t> uniform_bounds(int height, int world, int align = 8)
{
    std::vector<int32_t> b(world + 1);
    const int tiles = (height + align - 1) / align;
    for (int r = 0; r <= world; r++) b[r] = (int32_t)((long long)tiles * r / world) * align;
    b[world] = height;
    return b;
}
inline hr_band band_of(const std::vector<int32_t>& bounds, int rank, int halo, int history_halo)
{
    // the neighbour exchange refreshes `history_halo` rows per boundary from the DIRECT neighbour only: a shorter band would leave rows
    // of the second neighbour stale (hr_comm_exchange_rows refuses it too; tiling._TiledPass has the same guard)
    for (size_t r = 0; r + 1 < bounds.size(); r++)
        if (bounds.size() > 2 && bounds[r + 1] - bounds[r] < history_halo)
            throw std::invalid_argument("hr::Tiled*: band " + std::to_string(r) + " of the row-tiled frame is shorter than the " + std::to_string(history_halo) + "-row history apron: use fewer / taller bands");
    hr_band b;
    b.band_y0 = bounds[rank]; b.band_y1 = bounds[rank + 1]; b.halo = halo; b.history_halo = history_halo;
    return b;
}

class TiledShadows
{
public:
    TiledShadows(Context& ctx, Comm& comm, CommonResources* common, GBuffer* g_buffer, std::vector<int32_t> bounds, RayTraceScale scale = RAY_TRACE_SCALE_FULL_RES) :
        m_comm(comm), m_bounds(std::move(bounds)), m_band(band_of(m_bounds, comm.rank(), kHalo, kHistoryHalo)),
        m_pass(ctx, common, g_buffer, scale, comm.world() > 1 ? &m_band : nullptr), m_common(common), m_g_buffer(g_buffer)
    {
    }
    // RayTracedShadows::render stage by stage (ray_traced_shadows.cpp:100-116): the trace reads no history, so it is enqueued BEFORE
    // the compute stream is made to wait for last frame's halo rows; the exchange of this frame's rows is posted at the end and
    // overlaps whatever the application enqueues next
    void render(Stream cmd_buf)
    {
        const Frame f = make_frame(*m_common, *m_g_buffer, (int)m_pass.scale());
        hr_shadows* p = m_pass.handle();
        check(hr_shadows_ray_trace(p, f.scene->handle(), &f.inputs, &m_pass.params, cmd_buf), "TiledShadows::ray_trace");
        m_comm.wait(m_ticket, cmd_buf);   // the rows THIS pass exchanged last frame — not what another pass posted a moment ago
        if (m_pass.params.denoise)
        {
            check(hr_shadows_denoise(p, &f.inputs, &m_pass.params, cmd_buf), "TiledShadows::denoise");   // temporal + a-trous chain (+ upsample), as render()
            check(hr_shadows_exchange_history(p, m_comm.handle(), m_bounds.data(), f.inputs.ping_pong, kHistoryHalo, cmd_buf, &m_ticket), "TiledShadows::exchange");
        }
    }
    // a history tap fell on an image row this GPU does not hold (motion beyond the history apron): it read as disoccluded
    bool history_apron_exceeded() { int32_t v = 0; check(hr_shadows_history_apron_exceeded(m_pass.handle(), &v), "history_apron_exceeded"); return v != 0; }
    RayTracedShadows& pass() { return m_pass; }
    int band_y0() const { return m_bounds[m_comm.rank()]; }
    int band_y1() const { return m_bounds[m_comm.rank() + 1]; }
private:
    Comm&                m_comm;
    std::vector<int32_t> m_bounds;
    hr_band              m_band;
    RayTracedShadows     m_pass;
    CommonResources*     m_common;
    GBuffer*             m_g_buffer;
    hr_comm_ticket       m_ticket = 0;
};

class TiledAO
{
public:
    TiledAO(Context& ctx, Comm& comm, CommonResources* common, GBuffer* g_buffer, std::vector<int32_t> bounds, RayTraceScale scale = RAY_TRACE_SCALE_HALF_RES) :
        m_comm(comm), m_bounds(std::move(bounds)), m_band(band_of(m_bounds, comm.rank(), kHalo, kHalo)),
        m_pass(ctx, common, g_buffer, scale, comm.world() > 1 ? &m_band : nullptr), m_common(common)
    {
    }
    void render(Stream cmd_buf)
    {
        m_comm.wait(m_ticket, cmd_buf);
        m_pass.render(cmd_buf);
        if (m_pass.params.denoise) check(hr_ao_exchange_history(m_pass.handle(), m_comm.handle(), m_bounds.data(), m_common->ping_pong ? 1 : 0, kHalo, cmd_buf, &m_ticket), "TiledAO::exchange");
    }
    bool history_apron_exceeded() { int32_t v = 0; check(hr_ao_history_apron_exceeded(m_pass.handle(), &v), "history_apron_exceeded"); return v != 0; }
    RayTracedAO& pass() { return m_pass; }
private:
    Comm&                m_comm;
    std::vector<int32_t> m_bounds;
    hr_band              m_band;
    RayTracedAO          m_pass;
    CommonResources*     m_common;
    hr_comm_ticket       m_ticket = 0;
};

// DDGI: probes partitioned by grid z-slab for the ray trace and both probe updates, atlas rows all-gathered, every rank samples the
// full atlases for its own row band (tiling.ShardedDDGI)
class ShardedDDGI
{
public:
    ShardedDDGI(Context& ctx, Comm& comm, CommonResources* common, GBuffer* g_buffer, const hr_ddgi_uniforms& grid, std::vector<int32_t> bounds) :
        m_comm(comm), m_pass(ctx, common, g_buffer, grid), m_common(common), m_g_buffer(g_buffer)
    {
        const int cz = grid.probe_counts[2], w = comm.world(), r = comm.rank();
        if (w > 1) check(hr_ddgi_set_shard(m_pass.handle(), (int)((long long)cz * r / w), (int)((long long)cz * (r + 1) / w), bounds[r], bounds[r + 1]), "hr_ddgi_set_shard");
    }
    void render(Stream cmd_buf)
    {
        const Frame f = make_frame(*m_common, *m_g_buffer, 0);
        hr_ddgi* p = m_pass.handle();
        check(hr_ddgi_ray_trace(p, f.scene->handle(), &f.inputs, f.environment, &m_pass.params, cmd_buf), "ShardedDDGI::ray_trace");
        check(hr_ddgi_probe_update(p, cmd_buf), "ShardedDDGI::probe_update");
        hr_comm_ticket t = 0;
        check(hr_ddgi_allgather_atlases(p, m_comm.handle(), cmd_buf, &t), "ShardedDDGI::allgather");
        m_comm.wait(t, cmd_buf);   // loopback: the copies may sit on another rank's stream; RCCL: already ordered on cmd_buf
        check(hr_ddgi_sample_probe_grid(p, &f.inputs, &m_pass.params, cmd_buf), "ShardedDDGI::sample_probe_grid");
        check(hr_ddgi_end_frame(p), "ShardedDDGI::end_frame");
    }
    DDGI& pass() { return m_pass; }
private:
    Comm&            m_comm;
    DDGI             m_pass;
    CommonResources* m_common;
    GBuffer*         m_g_buffer;
};

class TiledReflections
{
public:
    TiledReflections(Context& ctx, Comm& comm, CommonResources* common, GBuffer* g_buffer, std::vector<int32_t> bounds, RayTraceScale scale = RAY_TRACE_SCALE_HALF_RES) :
        m_comm(comm), m_bounds(std::move(bounds)), m_band(band_of(m_bounds, comm.rank(), kHalo, kHalo)),
        m_pass(ctx, common, g_buffer, scale, comm.world() > 1 ? &m_band : nullptr), m_common(common)
    {
    }
    void render(Stream cmd_buf, DDGI* ddgi)
    {
        m_comm.wait(m_ticket, cmd_buf);
        m_pass.render(cmd_buf, ddgi);
        if (m_pass.params.denoise)
            check(hr_reflections_exchange_history(m_pass.handle(), m_comm.handle(), m_bounds.data(), m_common->ping_pong ? 1 : 0, kHalo, cmd_buf, &m_ticket), "TiledReflections::exchange");
    }
    bool history_apron_exceeded() { int32_t v = 0; check(hr_reflections_history_apron_exceeded(m_pass.handle(), &v), "history_apron_exceeded"); return v != 0; }
    RayTracedReflections& pass() { return m_pass; }
private:
    Comm&                m_comm;
    std::vector<int32_t> m_bounds;     // in rows of the PASS image (full height >> scale)
    hr_band              m_band;
    RayTracedReflections m_pass;
    CommonResources*     m_common;
    hr_comm_ticket       m_ticket = 0;
};

// The row-tiled frame with its independent chains forked over streams (the N > 1 counterpart of hr::HybridFrame): shadows and AO on
// two of the frame object's internal streams, DDGI -> reflections on cmd_buf.  Every pass still posts its own neighbour exchange from
// inside render() — on its own stream, all through ONE communicator: the collectives reach the communication stream in host-call order,
// which is the same on every rank, and each pass waits for its own ticket only (hr_comm.h).
class TiledHybridFrame
{
public:
    TiledHybridFrame(Context& ctx, TiledShadows* shadows, TiledAO* ao, ShardedDDGI* ddgi, TiledReflections* reflections) :
        m_shadows(shadows), m_ao(ao), m_ddgi(ddgi), m_reflections(reflections)
    {
        check(hr_hybrid_frame_create(ctx.handle(), shadows ? shadows->pass().handle() : nullptr, ao ? ao->pass().handle() : nullptr,
                                     ddgi ? ddgi->pass().handle() : nullptr, reflections ? reflections->pass().handle() : nullptr, &m_frame), "hr_hybrid_frame_create");
    }
    ~TiledHybridFrame() { hr_hybrid_frame_destroy(m_frame); }
    TiledHybridFrame(const TiledHybridFrame&) = delete;
    TiledHybridFrame& operator=(const TiledHybridFrame&) = delete;
    // main.cpp:80-83 in one call; forked = false: the four render() calls on cmd_buf
    void render(Stream cmd_buf, bool forked = true)
    {
        void* side[3] = { cmd_buf, cmd_buf, cmd_buf };
        if (forked) check(hr_hybrid_frame_fork(m_frame, cmd_buf, side), "hr_hybrid_frame_fork");
        if (m_ddgi) m_ddgi->render(cmd_buf);                         // the longest chain first
        if (m_shadows) m_shadows->render(side[0]);
        if (m_ao) m_ao->render(side[1]);
        if (m_reflections) m_reflections->render(cmd_buf, m_ddgi ? &m_ddgi->pass() : nullptr);
        if (forked) check(hr_hybrid_frame_join(m_frame, cmd_buf), "hr_hybrid_frame_join");
    }
private:
    TiledShadows*     m_shadows;
    TiledAO*          m_ao;
    ShardedDDGI*      m_ddgi;
    TiledReflections* m_reflections;
    hr_hybrid_frame*  m_frame = nullptr;
};

} // namespace hr
