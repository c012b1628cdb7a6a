"""Per-band work of the N-GPU bench frame, measured on ONE GPU: every band of the 1920 x (1080 N) frame is rendered in
turn by its own TiledShadows instance (no exchange), and its rays / frame time are printed.  The slowest band bounds the
N-GPU frame rate, so  sum(rays) / max(band ms)  vs  N x (single-GPU rays / ms)  predicts the weak-scaling efficiency
before communication.   python tools/band_balance.py [N ...]"""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hybrid_rendering_amd import api as hr, synth, tiling


def frame_setup(world, W=1920, h1=1080):
    """bench.py's weak-scaling frame: the same view with N x the pixels"""
    sc = math.sqrt(world)
    W, H = (int(round(W * sc / 8)) * 8, int(round(h1 * sc / 8)) * 8) if world > 1 else (W, h1)
    cams = [synth.sponza_camera(W / H, frame=f, dolly=0.5) for f in range(3)]
    return W, H, cams


def main():
    worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
    sd = synth.sponza_like(1.0)
    ctx = hr.Context(0)
    scene = hr.Scene(ctx, sd)
    light = synth.sponza_light()
    sob, sr = synth.blue_noise_tables()
    sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
    base = None
    for world in worlds:
        W, H, cams = frame_setup(world)
        ubos = [synth.make_ubo(cams[i + 1], cams[i], light) for i in range(2)]
        gbs = [scene.gbuffer(u, W, H) for u in ubos]
        rows = []
        bounds = None
        if world > 1 and not os.environ.get("UNIFORM_BANDS"):
            import numpy as np
            cal = hr.RayTracedShadows(ctx, W, H)
            cal.ray_trace(scene, hr.frame_inputs(gbs[0], gbs[1], ubos[0], 0, 0, sob_d, sr_d))
            cost = tiling.shadow_cost_per_tile_row(gbs[0]["depth"], cal.tile_ray_counts())
            cal.close()
            bounds = tiling.balanced_bounds(cost, world, H)
        for r in range(world):
            t = tiling.TiledShadows(ctx, W, H, r, world, bounds=bounds)
            t.world = 1   # no exchange: timing of the band's own work (band + halo)
            t.params.exact = int(os.environ.get("EXACT", "0"))   # the mode bench.py times by default
            fis = [hr.frame_inputs(gbs[k & 1], gbs[(k + 1) & 1], ubos[k & 1], k, k & 1, sob_d, sr_d) for k in range(2)]
            for k in range(6):
                fis[k & 1].num_frames = k
                t.render(scene, fis[k & 1])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 40
            for k in range(6, 6 + n):
                fis[k & 1].num_frames = k
                t.render(scene, fis[k & 1])
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            rays = int(t.pass_.tile_ray_counts()[t.b0 // 8:(t.b1 + 7) // 8].sum())   # band rows only: halo rays are overhead
            rows.append((r, t.b0, t.b1, rays, ms))
            del t
        tot = sum(x[3] for x in rows)
        worst = max(x[4] for x in rows)
        rate = tot / worst / 1e3
        if world == 1:
            base = rate
        print('N=%d  frame %dx%d  rays %d  slowest band %.3f ms  -> %.1f Mrays/s  predicted efficiency %.2f' % (
            world, W, H, tot, worst, rate, rate / (world * base) if base else 1.0))
        for r, b0, b1, rays, ms in rows:
            print('    band %d rows %5d-%5d  rays %8d  %.3f ms' % (r, b0, b1, rays, ms))
        del gbs


if __name__ == '__main__':
    main()
