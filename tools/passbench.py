"""Per-stage timings of every pass at a given resolution (developer tool; prints a JSON line per pass).

    python tools/passbench.py [--width 1920 --height 1080 --frames 40] [--passes shadows,ao,reflections,ddgi,post]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--passes", default="shadows,ao,reflections,ddgi")
    ap.add_argument("--ao-spp", type=int, default=4)
    ap.add_argument("--ao-scale", type=int, default=0)
    ap.add_argument("--refl-scale", type=int, default=1)
    ap.add_argument("--probes", default="16,8,16")
    ap.add_argument("--rays-per-probe", type=int, default=256)
    ap.add_argument("--tier", default="standard", choices=["standard", "hard"])
    ap.add_argument("--detail", type=float, default=1.0, help="tessellation of the standard tier's scene (1.0 = the bench scene; 3.0 ~ 2.5 M triangles)")
    ap.add_argument("--light", default="", choices=["", "standard", "hard"], help="default: the tier's light")
    ap.add_argument("--exact", type=int, default=0, help="1 = bit-for-bit parity arithmetic, 0 = tolerance mode (the shipping mode)")
    args = ap.parse_args()
    import torch
    from hybrid_rendering_amd import api as hr, api_gi, api_reflections, synth, synth_env
    W, H = args.width, args.height
    sd = synth.sponza_like(args.detail, tier=args.tier)
    ctx = hr.Context(0)
    scene = hr.Scene(ctx, sd)
    light = synth.sponza_hard_light() if (args.light or args.tier) == "hard" else synth.sponza_light()
    sob, sr = synth.blue_noise_tables()
    sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
    cams = [synth.sponza_camera(W / H, frame=f, dolly=0.5) for f in range(5)]
    gbs, ubos = [], []
    for i in range(4):
        ubo = synth.make_ubo(cams[i + 1], cams[i], light)
        ubos.append(ubo)
        gbs.append(scene.gbuffer(ubo, W, H))

    def mip(g, lvl):
        s = 1 << lvl
        return {k: v[::s, ::s].contiguous() if v.dim() == 2 else v[::s, ::s, :].contiguous() for k, v in g.items()}

    zbp = synth.z_buffer_params()
    lo, hi = sd.bounds()
    ddgi_u = synth_env.ddgi_uniforms(lo, hi, probe_counts=tuple(int(v) for v in args.probes.split(",")), rays_per_probe=args.rays_per_probe, normal_bias=0.1)
    sky = synth_env.sky_cubemap(32)
    f16 = lambda a: torch.from_numpy(a).cuda().view(torch.float16)
    env = api_gi.environment(f16(sky), f16(synth_env.prefiltered_chain(sky, 5)), 32, 5, f16(synth_env.brdf_lut(32)))
    rng = np.random.RandomState(1)
    want = args.passes.split(",")
    ddgi = api_gi.DDGI(ctx, W, H, ddgi_u)

    def run(name, make_pass, render, scale):
        p = make_pass()
        p.params.exact = args.exact
        lv = [mip(g, scale) if scale else g for g in gbs]
        fis = [hr.frame_inputs(lv[k % 4], lv[(k - 1) % 4], ubos[k % 4], k, k & 1, sob_d, sr_d, cur_full=gbs[k % 4], z_buffer_params=zbp) for k in range(8)]
        for k in range(8):
            fis[k].num_frames = k
            render(p, fis[k % 8], k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(8, 8 + args.frames):
            fis[k % 8].num_frames = k
            render(p, fis[k % 8], k)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.frames * 1e3
        p.set_profiling(True)
        acc, rays = {}, 0
        for k in range(8 + args.frames, 8 + args.frames + 10):
            fis[k % 8].num_frames = k
            render(p, fis[k % 8], k)
            rays += p.ray_count()
            for n, t, b in p.stage_times():
                a = acc.setdefault(n, [0.0, b]); a[0] += t
        out = dict(pass_=name, exact=args.exact, res=f"{W >> scale}x{H >> scale}", ms_per_frame=round(ms, 4), rays_per_frame=rays // 10,
                   Mrays_per_s=round(rays / 10 / (ms * 1e-3) / 1e6, 1),
                   stages={n: dict(ms=round(v[0] / 10, 4), GBps=round(v[1] / (v[0] / 10 * 1e-3) / 1e9, 1) if v[0] > 0 else 0) for n, v in acc.items()})
        print(json.dumps(out))
        return p

    if "shadows" in want:
        run("shadows", lambda: hr.RayTracedShadows(ctx, W, H), lambda p, fi, k: p.render(scene, fi), 0)
    if "ao" in want:
        def mk():
            p = hr.RayTracedAO(ctx, W, H, args.ao_scale); p.params.spp = args.ao_spp; return p
        run(f"ao_{args.ao_spp}spp", mk, lambda p, fi, k: p.render(scene, fi), args.ao_scale)
    if "ddgi" in want or "reflections" in want:
        def rd(p, fi, k):
            p.render(scene, fi, env, synth_env.random_orientation(rng))
        run("ddgi", lambda: ddgi, rd, 0)
    if "reflections" in want:
        run("reflections", lambda: api_reflections.RayTracedReflections(ctx, W, H, args.refl_scale), lambda p, fi, k: p.render(scene, fi, env, ddgi), args.refl_scale)
    if "post" in want:
        # the SURVEY 8f rows: deferred composite, TAA, ground-truth accumulator (wall clock per call; one kernel each)
        from hybrid_rendering_amd import api_deferred, api_post
        sh = hr.RayTracedShadows(ctx, W, H); ao = hr.RayTracedAO(ctx, W, H, 0); refl = api_reflections.RayTracedReflections(ctx, W, H, 0)
        fi = hr.frame_inputs(gbs[1], gbs[0], ubos[1], 1, 1, sob_d, sr_d, cur_full=gbs[1], z_buffer_params=zbp)
        sh.render(scene, fi); ao.render(scene, fi); ddgi.render(scene, fi, env, synth_env.random_orientation(rng)); refl.render(scene, fi, env, ddgi)
        de, taa, gt = api_deferred.DeferredShading(ctx, W, H), api_post.TemporalAA(ctx, W, H), api_post.GroundTruthPathTracer(ctx, W, H)
        s_o, a_o, r_o, g_o = sh.output(hr.OUTPUT_ATROUS), ao.output(hr.OUTPUT_UPSAMPLE), refl.output(hr.OUTPUT_UPSAMPLE), ddgi.output()

        def timed(fn, n=args.frames):
            for _ in range(5): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): fn()
            torch.cuda.synchronize()
            return round((time.perf_counter() - t0) / n * 1e3, 4)
        out = dict(pass_="post", res=f"{W}x{H}")
        out["deferred_ms"] = timed(lambda: de.render(fi, env, s_o, a_o, r_o, g_o))
        color = de.output()
        taa.update(3)
        out["taa_ms"] = timed(lambda: taa.render(color, gbs[1], 1))
        out["ground_truth_ms"] = timed(lambda: gt.render(scene, ubos[1], env))
        out["ground_truth_Mrays_per_s"] = round(gt.ray_count() / (out["ground_truth_ms"] * 1e-3) / 1e6, 1)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
