"""GPU developer tool: what the traversal loops of the trace kernels EXECUTE against what their lanes NEED.

Rebuilds the library with -DHR_TRACE_DIVERGENCE (traverse.h: every node step / triangle-pair step is counted per lane and once per
executing wave), renders a few hybrid frames and prints, per kernel and ray class:
  node / pair steps per ray            sum over lanes / rays
  executed wave steps                  what the SIMD issued
  lane utilisation                     needed / (64 * executed)
  'own pace' bound                     max over the wave's lanes of the lane's OWN total (a lane that starts its next ray without
                                       waiting for the wave, or refills from a queue, cannot beat this without regrouping rays)
The product build is restored afterwards.   usage (GPU box): python tools/divergence.py [--tier hard] [--size 1920x1080]"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tier", default="standard")
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--frames", type=int, default=3)
    args = ap.parse_args()
    env = dict(os.environ, HR_CFLAGS="-DHR_TRACE_DIVERGENCE")
    subprocess.check_call([sys.executable, "-m", "hybrid_rendering_amd.build", "--force"], cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        run(args)
    finally:
        subprocess.check_call([sys.executable, "-m", "hybrid_rendering_amd.build", "--force"], cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def run(args):
    import torch
    from hybrid_rendering_amd import api as hr, synth
    from hybrid_rendering_amd.frame import HybridFrame
    W, H = (int(v) for v in args.size.split("x"))
    sd = synth.sponza_like(1.0, tier=args.tier)
    ctx = hr.Context(0)
    scene = hr.Scene(ctx, sd)
    fr = HybridFrame(ctx, scene, sd, W, H)
    lib = hr.lib()
    for name in ("ao", "ddgi", "refl"):
        getattr(lib, "hr_debug_divergence_" + name).argtypes = [C.POINTER(C.c_uint64), C.c_int]
    for k in range(args.frames - 1):
        fr.render(k)
    torch.cuda.synchronize()
    for name in ("ao", "ddgi", "refl"):
        getattr(lib, "hr_debug_divergence_" + name)(None, 1)
    fr.render(args.frames - 1)
    torch.cuda.synchronize()
    rays = fr.ray_counts()
    print("rays", rays)

    def report(label, c):
        ln, lp, wn, wp, mn, mp, waves, lanes = c
        if not waves or not lanes:
            print(f"{label:24s} (no rays)")
            return
        print(f"{label:24s} waves {waves:6d} lanes/wave {lanes / waves:5.1f} | per traced lane: nodes {ln / lanes:6.2f} pairs {lp / lanes:6.2f} | per wave: executed nodes {wn / waves:6.1f} pairs {wp / waves:6.1f}"
              f" | utilisation nodes {ln / (64 * wn):.3f} pairs {lp / (64 * max(wp, 1)):.3f} | own-pace bound nodes {mn / waves:6.1f} pairs {mp / waves:6.1f} | ideal nodes {ln / 64 / waves:6.1f} pairs {lp / 64 / waves:6.1f}")

    buf = (C.c_uint64 * 16)()
    lib.hr_debug_divergence_ao(buf, 0)
    report("ao any-hit (spp rays)", list(buf)[:8])
    lib.hr_debug_divergence_ddgi(buf, 0)
    report("ddgi primary closest", list(buf)[:8])
    report("ddgi secondary any-hit", list(buf)[8:])
    lib.hr_debug_divergence_refl(buf, 0)
    report("refl primary closest", list(buf)[:8])
    report("refl secondary any-hit", list(buf)[8:])
    fr.close()


if __name__ == "__main__":
    main()
