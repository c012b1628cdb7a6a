"""GPU developer probe: cross-frame pipelining of the shadow pass.  The trace of frame k+1 reads only the G-buffer (and writes the mask
the temporal stage of frame k has finished reading), so it may run UNDER the a-trous chain of frame k: trace on stream A, denoise on
stream B, two cross-stream events per frame.  Compares frame times with everything on one stream (same stage calls).
python tools/pipeline_probe.py [width height]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hybrid_rendering_amd import api as hr, synth

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
sd = synth.sponza_like(1.0)
ctx = hr.Context(0)
scene = hr.Scene(ctx, sd)
light = synth.sponza_light()
sob, sr = synth.blue_noise_tables()
sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
cams = [synth.sponza_camera(W / H, frame=f, dolly=0.5) for f in range(3)]
ubos = [synth.make_ubo(cams[i + 1], cams[i], light) for i in range(2)]
gbs = [scene.gbuffer(u, W, H) for u in ubos]
fis = [hr.frame_inputs(gbs[k & 1], gbs[(k + 1) & 1], ubos[k & 1], k, k & 1, sob_d, sr_d) for k in range(2)]
p = hr.RayTracedShadows(ctx, W, H)
p.params.exact = 0
A, B = torch.cuda.Stream(), torch.cuda.Stream()
ev_trace = [torch.cuda.Event() for _ in range(4)]
ev_temporal = [torch.cuda.Event() for _ in range(4)]


def serial_render(k):
    fis[k & 1].num_frames = k
    p.render(scene, fis[k & 1])


def serial_stages(k):
    fi = fis[k & 1]
    fi.num_frames = k
    p.ray_trace(scene, fi)
    p.temporal(fi)
    for i in range(4):
        p.atrous_iteration(fi, i)


def pipelined(k):
    fi = fis[k & 1]
    fi.num_frames = k
    A.wait_event(ev_temporal[(k - 1) & 3])      # the mask is free once temporal(k-1) has read it
    p.ray_trace(scene, fi, A)
    ev_trace[k & 3].record(A)
    B.wait_event(ev_trace[k & 3])
    p.temporal(fi, B)
    ev_temporal[k & 3].record(B)
    for i in range(4):
        p.atrous_iteration(fi, i, B)


def timed(fn, n=100):
    for e in ev_temporal:
        e.record(B)
    for k in range(8):
        fn(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(8, 8 + n):
        fn(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, fn in (("render() on one stream", serial_render), ("stage calls on one stream (unfused a-trous)", serial_stages), ("trace | denoise pipelined over two streams", pipelined)):
    print(f"{W}x{H} {name}: {min(timed(fn) for _ in range(3)):.4f} ms / frame", flush=True)
