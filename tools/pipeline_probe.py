"""Developer probe: would two frames in flight pay for the shadow pass?  The trace of frame f + 1 only needs frame f's temporal accumulation to be
over (it rewrites the mask image that kernel reads, and takes its launch order from it); frame f's a-trous iterations could run beside it.
Three timings of the same launches (stage entry points of hr_api_stages.h, un-fused a-trous):
  serial       everything on one stream
  two streams  trace on stream A, denoise on stream B, events: trace(f) -> temporal(f) -> trace(f + 1)
Prints ms per frame for both and checks the final images agree bit for bit.

    python tools/pipeline_probe.py [--width 1920 --height 1080 --frames 400]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=400)
    args = ap.parse_args()
    import torch
    from hybrid_rendering_amd import api as hr, synth
    W, H = args.width, args.height
    sd = synth.sponza_like(1.0)
    ctx = hr.Context(0)
    scene = hr.Scene(ctx, sd)
    light = synth.sponza_light()
    sob, sr = synth.blue_noise_tables()
    sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
    cams = [synth.sponza_camera(W / H, frame=f, dolly=0.5) for f in range(5)]
    gbs, ubos = [], []
    for i in range(4):
        ubo = synth.make_ubo(cams[i + 1], cams[i], light)
        ubos.append(ubo)
        gbs.append(scene.gbuffer(ubo, W, H))
    fis = [hr.frame_inputs(gbs[k % 4], gbs[(k - 1) % 4], ubos[k % 4], k, k & 1, sob_d, sr_d) for k in range(8)]
    A, B = torch.cuda.Stream(), torch.cuda.Stream()

    def run(mode, fused):
        p = hr.RayTracedShadows(ctx, W, H)
        p.params.exact = 0
        n_it = p.params.filter_iterations
        ev_trace = [torch.cuda.Event() for _ in range(4)]
        ev_temp = [torch.cuda.Event() for _ in range(4)]

        def frame(k):
            fi = fis[k % 8]
            fi.num_frames = k
            if mode == "serial":
                if fused:
                    p.render(scene, fi, stream=A)
                else:
                    p.ray_trace(scene, fi, stream=A)
                    p.temporal(fi, stream=A)
                    for i in range(n_it):
                        p.atrous_iteration(fi, i, stream=A)
                return
            if k > 0:
                A.wait_event(ev_temp[(k - 1) % 4])
            p.ray_trace(scene, fi, stream=A)
            ev_trace[k % 4].record(A)
            B.wait_event(ev_trace[k % 4])
            p.temporal(fi, stream=B)
            ev_temp[k % 4].record(B)
            for i in range(n_it):
                p.atrous_iteration(fi, i, stream=B)
        for k in range(16):
            frame(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(16, 16 + args.frames):
            frame(k)
            if k % 16 == 0:
                torch.cuda.synchronize() if False else None
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.frames * 1e3
        out = p.output().clone()
        rays = p.ray_count()
        p.close()
        return ms, out, rays
    res = {}
    for mode, fused in (("serial", True), ("serial", False), ("streams", False), ("serial", False), ("streams", False)):
        ms, out, rays = run(mode, fused)
        print(f"{mode:8s} fused_atrous01={int(fused)}  {ms:.4f} ms per frame   rays {rays}")
        res.setdefault((mode, fused), []).append(out)
    a, b = res[("serial", False)][0], res[("streams", False)][0]
    print("images equal (serial vs two streams):", bool(torch.equal(a, b)))


if __name__ == "__main__":
    main()
