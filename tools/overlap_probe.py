"""How much frame time is there to win by running the ray-trace stage of frame N+1 on a second HIP stream while
frame N is being denoised?  Timing probe only (no inter-stream dependencies, so the images are garbage)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hybrid_rendering_amd import api as hr, synth

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
sd = synth.sponza_like(1.0); ctx = hr.Context(0); sc = hr.Scene(ctx, sd)
cam = synth.sponza_camera(W / H); ubo = synth.make_ubo(cam, None, synth.sponza_light())
gb = sc.gbuffer(ubo, W, H); sob, sr = synth.blue_noise_tables()
sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
p = hr.RayTracedShadows(ctx, W, H)
s2 = torch.cuda.Stream()
main = torch.cuda.current_stream()


def run(frames, overlap):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(frames):
        fi = hr.frame_inputs(gb, gb, ubo, i, i & 1, sob_d, sr_d)
        p.ray_trace(sc, fi, stream=s2 if overlap else main)
        p.temporal(fi, stream=main)
        for k in range(4):
            p.atrous_iteration(fi, k, stream=main)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / frames * 1e3


for mode in (False, True, False, True):
    run(20, mode)
    print('cross-frame overlap=%s  %.4f ms/frame' % (mode, run(200, mode)))


def run_intra(frames, overlap, prio):
    # trace || temporal inside one frame (fork/join), a-trous after the join: upper bound for splitting the
    # mask-independent reprojection out of the temporal kernel
    s = torch.cuda.Stream(priority=prio)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(frames):
        fi = hr.frame_inputs(gb, gb, ubo, i, i & 1, sob_d, sr_d)
        if overlap:
            s.wait_stream(main)
            p.temporal(fi, stream=s)
            p.ray_trace(sc, fi, stream=main)
            main.wait_stream(s)
        else:
            p.ray_trace(sc, fi, stream=main)
            p.temporal(fi, stream=main)
        for k in range(4):
            p.atrous_iteration(fi, k, stream=main)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / frames * 1e3


for overlap, prio in ((False, 0), (True, 0), (True, -1), (False, 0), (True, 0)):
    run_intra(20, overlap, prio)
    print('intra-frame overlap=%s side-priority=%d  %.4f ms/frame' % (overlap, prio, run_intra(200, overlap, prio)))
