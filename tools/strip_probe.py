"""GPU developer probe: can the tail of the shadow trace be hidden under the denoise of an EARLIER strip of the same frame?
Emulated with row bands (each band = its own pass instance + 24 halo rows, so slightly MORE work than real strips would do):
band k traces on its own stream (optionally with descending priorities) and runs its denoise chain right behind; compare the
frame time with the whole-frame pass on one stream.   python tools/strip_probe.py [bands] [priorities 0/1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hybrid_rendering_amd import api as hr, synth, tiling

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2
prio = int(sys.argv[2]) if len(sys.argv) > 2 else 1
W, H = 1920, 1080
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
whole = hr.RayTracedShadows(ctx, W, H)
whole.params.exact = 0
bounds = [((H // 8) * r // nb) * 8 for r in range(nb)] + [H]
bands = [hr.RayTracedShadows(ctx, W, H, 0, band=(bounds[r], bounds[r + 1], tiling.HALO, tiling.HISTORY_HALO)) for r in range(nb)]
for b in bands:
    b.params.exact = 0
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
streams = [torch.cuda.Stream(priority=(hi if (prio and r == 0) else 0)) for r in range(nb)]


def frame_whole(k):
    fis[k & 1].num_frames = k
    whole.render(scene, fis[k & 1])


def frame_bands(k):
    fi = fis[k & 1]
    fi.num_frames = k
    ev = torch.cuda.Event()
    ev.record()
    for r in range(nb):
        streams[r].wait_event(ev)
        bands[r].ray_trace(scene, fi, streams[r])
    for r in range(nb):
        bands[r].denoise(fi, streams[r])
    for r in range(nb):
        e = torch.cuda.Event()
        e.record(streams[r])
        torch.cuda.current_stream().wait_event(e)


def timed(fn, n=60):
    for k in range(8):
        fn(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(8, 8 + n):
        fn(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"whole frame, one stream: {timed(frame_whole):.4f} ms")
print(f"{nb} bands (+ halo rows) on {nb} streams, priorities {'on' if prio else 'off'}: {timed(frame_bands):.4f} ms")
