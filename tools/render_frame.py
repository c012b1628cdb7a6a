"""Renders N frames of the whole pass chain (shadows, AO half-res, DDGI, reflections half-res, deferred composite, TAA, tone
map) of the procedural Sponza-like bench scene on one GPU and writes the last one as a PNG — a look at what the numbers in
bench.py are numbers of.   python tools/render_frame.py [--width 960 --height 540 --frames 24 --out gpurun_out/frame.png]"""
import argparse, os, struct, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_png(path, img):
    h, w, _ = img.shape
    raw = b"".join(b"\0" + img[y].tobytes() for y in range(h))
    chunk = lambda t, b: struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xffffffff)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--height", type=int, default=544)
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--exposure", type=float, default=1.0)
    ap.add_argument("--textured", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "frame.png"))
    a = ap.parse_args()
    import torch
    from hybrid_rendering_amd import api as hr, api_deferred, api_gi, api_post, api_reflections, synth, synth_env
    W, H = a.width, a.height
    sd = synth.sponza_like(1.0)
    if a.textured:
        sd = synth.with_textures(sd)
    ctx = hr.Context(0)
    scene = hr.Scene(ctx, sd)
    light = synth.sponza_light()
    sob, sr = synth.blue_noise_tables()
    sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
    lo, hi = sd.bounds()
    ddgi_u = synth_env.ddgi_uniforms(lo, hi, probe_counts=(16, 8, 16), rays_per_probe=256, normal_bias=0.1)
    sky = synth_env.sky_cubemap(32)
    f16 = lambda x: torch.from_numpy(x).cuda().view(torch.float16)
    env = api_gi.environment(f16(sky), f16(synth_env.prefiltered_chain(sky, 5)), 32, 5, f16(synth_env.brdf_lut(32)))
    zbp = synth.z_buffer_params()
    sh, ao = hr.RayTracedShadows(ctx, W, H), hr.RayTracedAO(ctx, W, H, hr.SCALE_HALF_RES)
    gi, rf = api_gi.DDGI(ctx, W, H, ddgi_u), api_reflections.RayTracedReflections(ctx, W, H, hr.SCALE_HALF_RES)
    df, taa = api_deferred.DeferredShading(ctx, W, H), api_post.TemporalAA(ctx, W, H)
    df.set_sh9(synth_env.sh9_from_cubemap(sky))
    df.params.use_ray_traced_shadows = df.params.use_ray_traced_ao = df.params.use_ray_traced_reflections = df.params.use_ddgi = 1
    taa.params.reset = 0
    rng = np.random.RandomState(1)
    cams = [synth.sponza_camera(W / H, frame=f, dolly=0.05) for f in range(a.frames + 1)]
    prev = None
    for f in range(a.frames):
        ubo = synth.make_ubo(cams[f + 1], cams[f], light)
        cur = scene.gbuffer(ubo, W, H)
        half = hr.gbuffer_mip(cur, 1)
        p, ph = (prev or (cur, half))
        fi = hr.frame_inputs(cur, p, ubo, f, f & 1, sob_d, sr_d, z_buffer_params=zbp)
        fh = hr.frame_inputs(half, ph, ubo, f, f & 1, sob_d, sr_d, cur_full=cur, z_buffer_params=zbp)
        sh.render(scene, fi); ao.render(scene, fh); gi.render(scene, fi, env, synth_env.random_orientation(rng)); rf.render(scene, fh, env, gi)
        df.render(fi, env, shadow=sh.output(hr.OUTPUT_UPSAMPLE), ao=ao.output(hr.OUTPUT_UPSAMPLE), reflections=rf.output(hr.OUTPUT_UPSAMPLE), gi=gi.output())
        taa.update(f)
        taa.render(df.output(), cur, f & 1)
        prev = (cur, half)
    _, ldr = api_post.tone_map(ctx, taa.output((a.frames - 1) & 1), False, a.exposure)
    torch.cuda.synchronize()
    img = ldr.cpu().numpy()[..., :3]
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    write_png(a.out, np.ascontiguousarray(img))
    print("wrote", a.out, img.shape, "mean", float(img.mean()))


if __name__ == "__main__":
    main()
