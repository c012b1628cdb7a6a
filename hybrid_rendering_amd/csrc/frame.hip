// hr_hybrid_frame — the four ray-traced passes of one frame (main.cpp:80-83: shadows, AO, DDGI, reflections) enqueued as the
// dependency graph they really form instead of as one serial command buffer:
//
//        +--> shadows: trace -> temporal -> a-trous ---------------------------------+
//   in --+--> AO:      trace -> temporal -> blur ------------------------------------+--> out
//        +--> DDGI:    probe trace -> probe updates -+--> probe-grid sample ---------+
//                                                    +--> reflections: trace -> ... -+
//
// The reference records the passes into one Vulkan command buffer with per-resource barriers only, so the GPU overlaps them as far as
// the barriers allow; a serial HIP stream forbids that.  The chains share no image (shadows, AO and DDGI read the G-buffer only;
// reflections read the DDGI ATLASES, which are final after the probe updates; the per-pixel probe-grid sample feeds only the
// composite), so every output is bit-identical to the serial order (tests/test_gpu_configs.py::test_hybrid_frame_*).  What the overlap buys: the
// latency-bound denoise kernels of one chain fill the SIMD slots the VALU-bound trace kernels of another leave idle.
//
// Three ways to enqueue the same launches:
//   HR_FRAME_SERIAL   everything on the caller's stream, in the reference's call order
//   HR_FRAME_STREAMS  fork / join with events over three internal streams + the caller's stream
//   HR_FRAME_GRAPH    the forked frame captured into a hipGraph (cross-stream capture) and launched as ONE graph; the instantiated
//                     graph is kept and UPDATED from each frame's capture (hipGraphExecUpdate: same topology, new kernel arguments —
//                     the per-frame UBO, frame counter and ping-pong parity travel by value in the kernel arguments)
// Host code only.
#include "hr_internal.h"

using namespace hr;

struct hr_hybrid_frame
{
    hr_ctx*         ctx = nullptr;
    hr_shadows*     shadows = nullptr;
    hr_ao*          ao = nullptr;
    hr_ddgi*        ddgi = nullptr;
    hr_reflections* reflections = nullptr;
    hipStream_t     side[3] = { nullptr, nullptr, nullptr };   // shadows, AO, probe-grid sample
    hipStream_t     capture = nullptr;                         // origin stream of the graph capture (the caller's may be the legacy stream)
    hipEvent_t      ev_in = nullptr, ev_atlas = nullptr, ev_out[3] = { nullptr, nullptr, nullptr }, ev_done = nullptr;
    hipGraphExec_t  exec = nullptr;
    hipStream_t     last_launch = nullptr;                     // stream of the last hipGraphLaunch (synchronised before the exec is destroyed)
    int             instantiations = 0, updates = 0;
};

namespace {

// A frame that was captured but never ran (capture / instantiate / launch failed) has still advanced the passes' host state — ping-pong
// parities, first-frame flags, the DDGI atlas swap: restart their histories so that the next frame does not read images nobody wrote.
void reset_after_failed_frame(hr_hybrid_frame* f)
{
    if (f->shadows) (void)hr_shadows_reset_history(f->shadows);
    if (f->ao) (void)hr_ao_reset_history(f->ao);
    if (f->reflections) (void)hr_reflections_reset_history(f->reflections);
    if (f->ddgi) (void)hr_ddgi_restart_accumulation(f->ddgi);
}

// the forked frame on `main` + the three side streams (used directly by HR_FRAME_STREAMS and under capture by HR_FRAME_GRAPH)
hr_status enqueue_forked(hr_hybrid_frame* f, const hr_scene* scene, const hr_hybrid_frame_desc* d, hipStream_t main)
{
    hr_status s;
    HR_HIP(hipEventRecord(f->ev_in, main));
    // the longest chain first: DDGI probe trace + updates, then the reflections that read the atlases
    if (f->ddgi)
    {
        if ((s = hr_ddgi_ray_trace(f->ddgi, scene, d->ddgi_inputs, d->environment, d->ddgi_params, main)) != HR_OK) return s;
        if ((s = hr_ddgi_probe_update(f->ddgi, main)) != HR_OK) return s;
    }
    if (f->shadows)
    {
        HR_HIP(hipStreamWaitEvent(f->side[0], f->ev_in, 0));
        if ((s = hr_shadows_render(f->shadows, scene, d->shadows_inputs, d->shadows_params, f->side[0])) != HR_OK) return s;
        HR_HIP(hipEventRecord(f->ev_out[0], f->side[0]));
    }
    if (f->ao)
    {
        HR_HIP(hipStreamWaitEvent(f->side[1], f->ev_in, 0));
        if ((s = hr_ao_render(f->ao, scene, d->ao_inputs, d->ao_params, f->side[1])) != HR_OK) return s;
        HR_HIP(hipEventRecord(f->ev_out[1], f->side[1]));
    }
    if (f->ddgi)
    {
        // the per-pixel sample only feeds the composite: off the DDGI -> reflections chain
        HR_HIP(hipEventRecord(f->ev_atlas, main));
        HR_HIP(hipStreamWaitEvent(f->side[2], f->ev_atlas, 0));
        if ((s = hr_ddgi_sample_probe_grid(f->ddgi, d->ddgi_inputs, d->ddgi_params, f->side[2])) != HR_OK) return s;
        HR_HIP(hipEventRecord(f->ev_out[2], f->side[2]));
        if ((s = hr_ddgi_end_frame(f->ddgi)) != HR_OK) return s;
    }
    if (f->reflections)
    {
        if ((s = hr_reflections_render(f->reflections, scene, d->reflections_inputs, d->environment, f->ddgi, d->reflections_params, main)) != HR_OK) return s;
    }
    if (f->shadows) HR_HIP(hipStreamWaitEvent(main, f->ev_out[0], 0));
    if (f->ao) HR_HIP(hipStreamWaitEvent(main, f->ev_out[1], 0));
    if (f->ddgi) HR_HIP(hipStreamWaitEvent(main, f->ev_out[2], 0));
    return HR_OK;
}

} // namespace

extern "C" {

hr_status hr_hybrid_frame_create(hr_ctx* ctx, hr_shadows* shadows, hr_ao* ao, hr_ddgi* ddgi, hr_reflections* reflections, hr_hybrid_frame** out)
{
    HR_CHECK_ARG(ctx && out && (shadows || ao || ddgi || reflections));
    if (reflections && !ddgi) { set_last_error("hr_hybrid_frame_create: reflections need the DDGI pass they read (ray_traced_reflections.h:27)"); return HR_ERR_INVALID_ARG; }
    HR_HIP(hipSetDevice(ctx->device));
    hr_hybrid_frame* f = new (std::nothrow) hr_hybrid_frame();
    if (!f) return HR_ERR_OUT_OF_MEMORY;
    f->ctx = ctx; f->shadows = shadows; f->ao = ao; f->ddgi = ddgi; f->reflections = reflections;
    hipError_t e = hipSuccess;
    for (int i = 0; i < 3 && e == hipSuccess; i++) e = hipStreamCreateWithFlags(&f->side[i], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&f->capture, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_in, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_atlas, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_done, hipEventDisableTiming);
    for (int i = 0; i < 3 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&f->ev_out[i], hipEventDisableTiming);
    if (e != hipSuccess) { set_last_error(std::string("hr_hybrid_frame_create: ") + hipGetErrorString(e)); hr_hybrid_frame_destroy(f); return HR_ERR_HIP; }
    *out = f;
    return HR_OK;
}

hr_status hr_hybrid_frame_destroy(hr_hybrid_frame* f)
{
    if (!f) return HR_OK;
    (void)hipSetDevice(f->ctx->device);
    (void)hipDeviceSynchronize();
    if (f->exec) (void)hipGraphExecDestroy(f->exec);
    for (hipStream_t s : f->side) if (s) (void)hipStreamDestroy(s);
    if (f->capture) (void)hipStreamDestroy(f->capture);
    for (hipEvent_t e : { f->ev_in, f->ev_atlas, f->ev_done, f->ev_out[0], f->ev_out[1], f->ev_out[2] }) if (e) (void)hipEventDestroy(e);
    delete f;
    return HR_OK;
}

hr_status hr_hybrid_frame_render(hr_hybrid_frame* f, const hr_scene* scene, const hr_hybrid_frame_desc* d, hr_frame_mode mode, void* stream_)
{
    HR_CHECK_ARG(f && scene && d);
    HR_CHECK_ARG((!f->shadows || (d->shadows_inputs && d->shadows_params)) && (!f->ao || (d->ao_inputs && d->ao_params)) &&
                 (!f->ddgi || (d->ddgi_inputs && d->ddgi_params && d->environment)) && (!f->reflections || (d->reflections_inputs && d->reflections_params)));
    hipStream_t main = (hipStream_t)stream_;
    HR_HIP(hipSetDevice(f->ctx->device));
    hr_status s;
    if (mode == HR_FRAME_SERIAL)
    {
        // the reference's call order on one stream (main.cpp:80-83)
        if (f->shadows && (s = hr_shadows_render(f->shadows, scene, d->shadows_inputs, d->shadows_params, main)) != HR_OK) return s;
        if (f->ao && (s = hr_ao_render(f->ao, scene, d->ao_inputs, d->ao_params, main)) != HR_OK) return s;
        if (f->ddgi && (s = hr_ddgi_render(f->ddgi, scene, d->ddgi_inputs, d->environment, d->ddgi_params, main)) != HR_OK) return s;
        if (f->reflections && (s = hr_reflections_render(f->reflections, scene, d->reflections_inputs, d->environment, f->ddgi, d->reflections_params, main)) != HR_OK) return s;
        return HR_OK;
    }
    if (mode == HR_FRAME_STREAMS) return enqueue_forked(f, scene, d, main);
    if (mode != HR_FRAME_GRAPH) { set_last_error("hr_hybrid_frame_render: unknown mode"); return HR_ERR_INVALID_ARG; }
    if ((f->shadows && hr::profiling_enabled(f->shadows)) || (f->ao && hr::profiling_enabled(f->ao)) || (f->ddgi && hr::profiling_enabled(f->ddgi)) ||
        (f->reflections && hr::profiling_enabled(f->reflections)))
    {
        set_last_error("hr_hybrid_frame_render: HR_FRAME_GRAPH cannot carry the passes' per-stage profiling events (timing events are not capturable); switch profiling off or use HR_FRAME_STREAMS");
        return HR_ERR_INVALID_ARG;
    }
    // ---- one hipGraph per frame: capture the forked frame (the side streams join the capture through their event waits), then update
    // the instantiated graph in place — topology and kernels are those of the last frame, only the argument blocks differ
    HR_HIP(hipStreamBeginCapture(f->capture, hipStreamCaptureModeThreadLocal));
    s = enqueue_forked(f, scene, d, f->capture);
    hipGraph_t graph = nullptr;
    const hipError_t ce = hipStreamEndCapture(f->capture, &graph);
    if (s != HR_OK) { if (graph) (void)hipGraphDestroy(graph); reset_after_failed_frame(f); return s; }
    if (ce != hipSuccess || !graph) { reset_after_failed_frame(f); set_last_error(std::string("hr_hybrid_frame_render: stream capture failed: ") + hipGetErrorString(ce)); return HR_ERR_HIP; }
    bool fresh = f->exec == nullptr;
    if (!fresh)
    {
        hipGraphNode_t           bad = nullptr;
        hipGraphExecUpdateResult res;
        if (hipGraphExecUpdate(f->exec, graph, &bad, &res) != hipSuccess)
        {
            (void)hipGetLastError();
            // the old graph may still be running — on the stream it was LAUNCHED on, which need not be this frame's (topology changes are rare: a pass's first frame)
            if (f->last_launch && f->last_launch != main) (void)hipStreamSynchronize(f->last_launch);
            (void)hipStreamSynchronize(main);
            (void)hipGraphExecDestroy(f->exec);
            f->exec = nullptr;
            fresh = true;
        }
        else
            f->updates++;
    }
    if (fresh)
    {
        const hipError_t ie = hipGraphInstantiate(&f->exec, graph, nullptr, nullptr, 0);
        if (ie != hipSuccess) { (void)hipGraphDestroy(graph); f->exec = nullptr; reset_after_failed_frame(f); set_last_error(std::string("hipGraphInstantiate: ") + hipGetErrorString(ie)); return HR_ERR_HIP; }
        f->instantiations++;
    }
    (void)hipGraphDestroy(graph);
    {
        const hipError_t le = hipGraphLaunch(f->exec, main);
        if (le != hipSuccess) { reset_after_failed_frame(f); set_last_error(std::string("hipGraphLaunch: ") + hipGetErrorString(le)); return HR_ERR_HIP; }
    }
    f->last_launch = main;
    return HR_OK;
}

// Fork / join for hosts that enqueue the chains themselves (the row-tiled passes of include/hr/tiled.hpp post their neighbour exchanges
// from inside render(): hr::TiledHybridFrame).  fork: the three internal streams wait for everything enqueued on `main` so far and are
// handed out; join: `main` waits for everything enqueued on them since.
hr_status hr_hybrid_frame_fork(hr_hybrid_frame* f, void* main_, void** side_streams)
{
    HR_CHECK_ARG(f && side_streams);
    hipStream_t main = (hipStream_t)main_;
    HR_HIP(hipSetDevice(f->ctx->device));
    HR_HIP(hipEventRecord(f->ev_in, main));
    for (int i = 0; i < 3; i++)
    {
        HR_HIP(hipStreamWaitEvent(f->side[i], f->ev_in, 0));
        side_streams[i] = (void*)f->side[i];
    }
    return HR_OK;
}

hr_status hr_hybrid_frame_join(hr_hybrid_frame* f, void* main_)
{
    HR_CHECK_ARG(f);
    hipStream_t main = (hipStream_t)main_;
    HR_HIP(hipSetDevice(f->ctx->device));
    for (int i = 0; i < 3; i++)
    {
        HR_HIP(hipEventRecord(f->ev_out[i], f->side[i]));
        HR_HIP(hipStreamWaitEvent(main, f->ev_out[i], 0));
    }
    return HR_OK;
}

hr_status hr_hybrid_frame_graph_stats(hr_hybrid_frame* f, int32_t* instantiations, int32_t* updates)
{
    HR_CHECK_ARG(f);
    if (instantiations) *instantiations = f->instantiations;
    if (updates) *updates = f->updates;
    return HR_OK;
}

} // extern "C"
