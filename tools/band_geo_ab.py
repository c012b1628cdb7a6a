"""GPU developer tool (round 5): what the geometry records buy a BAND of a row-tiled frame.  One interior band of BASELINE configs[4]'s geometry
(3840 wide, 272 rows + halos) next to the whole 3840x816 frame, shadows and AO 4 spp, tolerance mode, the reference's G-buffer ping-pong; temporal
stage time with HR_GEO_HISTORY = 1 / 0 (read at create).  Prints µs and ns per computed pixel.     python tools/band_geo_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from hybrid_rendering_amd import api as hr, synth
    W, rows, world = 3840, 272, 3
    H = rows * world
    sd = synth.sponza_like(1.0)
    ctx = hr.Context(0)
    scene = hr.Scene(ctx, sd)
    light = synth.sponza_light()
    cams = [synth.sponza_camera(W / H, frame=f, dolly=0.5) for f in range(3)]
    ubos = [synth.make_ubo(cams[i + 1], cams[i], light) for i in range(2)]
    gbs = [scene.gbuffer(u, W, H) for u in ubos]
    sob, sr = synth.blue_noise_tables()
    sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
    fis = [hr.frame_inputs(gbs[k & 1], gbs[(k + 1) & 1], ubos[k & 1], k, k & 1, sob_d, sr_d, z_buffer_params=synth.z_buffer_params()) for k in range(2)]
    for geo in ("1", "0"):
        os.environ["HR_GEO_HISTORY"] = geo
        passes = {"shadows whole": hr.RayTracedShadows(ctx, W, H), "shadows band": hr.RayTracedShadows(ctx, W, H, 0, band=(rows, 2 * rows, 24, 40)),
                  "ao whole": hr.RayTracedAO(ctx, W, H, 0), "ao band": hr.RayTracedAO(ctx, W, H, 0, band=(rows, 2 * rows, 24, 24))}
        for name, p in passes.items():
            p.params.exact = 0
            if hasattr(p.params, "spp"):
                p.params.spp = 4
            for k in range(6):
                fis[k & 1].num_frames = k
                p.render(scene, fis[k & 1])
            p.set_profiling(True)
            acc, n = 0.0, 30
            for k in range(6, 6 + n):
                fis[k & 1].num_frames = k
                p.render(scene, fis[k & 1])
                acc += dict((s, ms) for s, ms, _ in p.stage_times())["temporal_accumulation"]
            px = W * (H if "whole" in name else rows + 48)
            print(f"HR_GEO_HISTORY={geo}  {name:14s} temporal {acc / n * 1e3:7.1f} us   {acc / n * 1e6 / px:6.4f} ns per computed pixel", flush=True)
            p.close()


if __name__ == "__main__":
    main()
