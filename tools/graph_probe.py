"""GPU developer tool: what HR_FRAME_GRAPH costs on the host against HR_FRAME_STREAMS — per-frame host time of hr_hybrid_frame_render
(no synchronisation inside the loop, so the call's own cost) next to the GPU frame time.   python tools/graph_probe.py [--size 1920x1080]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--frames", type=int, default=60)
    a = ap.parse_args()
    import torch
    from hybrid_rendering_amd import api as hr, synth
    from hybrid_rendering_amd.frame import HybridFrame
    W, H = (int(v) for v in a.size.split("x"))
    sd = synth.sponza_like(1.0)
    ctx = hr.Context(0)
    scene = hr.Scene(ctx, sd)
    hf = HybridFrame(ctx, scene, sd, W, H)
    for mode in ("serial", "streams", "graph"):
        hf.concurrent_streams(mode != "serial", mode if mode != "serial" else "streams")
        for k in range(8):
            hf.render(k)
        torch.cuda.synchronize()
        host = []
        t0 = time.perf_counter()
        for k in range(8, 8 + a.frames):
            h0 = time.perf_counter()
            hf.render(k)
            host.append(time.perf_counter() - h0)
            if k % 4 == 3:
                torch.cuda.synchronize()   # keep the queue shallow: the host time is the call, not queue back-pressure
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / a.frames * 1e3
        host.sort()
        print(f"{a.size} {mode:8s} wall {wall:.3f} ms/frame (sync every 4 frames)   host call: median {host[len(host)//2]*1e3:.3f} ms, p90 {host[int(len(host)*0.9)]*1e3:.3f} ms", flush=True)
        print(f"{a.size} {mode:8s} free-running {hf.time(24, 4, repeats=3):.3f} ms/frame", flush=True)


main()
