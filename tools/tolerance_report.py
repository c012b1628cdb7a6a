"""GPU developer tool: where does the tolerance mode (exact = 0) deviate from the exact mode, stage by stage?
Runs both modes of a pass side by side on the same frames (each with its own history) and prints, per frame and stage image,
the relative L2 error, the share of texels within 2 fp16 ulp, percentiles of the ulp distance and the worst texels.
    python tools/tolerance_report.py [shadows|ao|reflections|ddgi] [W H frames]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import helpers
from hybrid_rendering_amd import api as hr, api_gi, api_reflections, synth, synth_env
from oracle import pyoracle as oracle

which = sys.argv[1] if len(sys.argv) > 1 else "shadows"
W, H, N = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (320, 184, 4)
name = "sponza_small"


def key(bits):
    b = bits.astype(np.int32); mag = b & 0x7fff
    return np.where(b & 0x8000, -mag, mag)


def report(tag, got, ref):
    got, ref = np.atleast_3d(got), np.atleast_3d(ref)
    for c in range(got.shape[2]):
        g16, r16 = got[..., c], ref[..., c]
        g, r = g16.view(np.float16).astype(np.float64), r16.view(np.float16).astype(np.float64)
        if not (np.abs(r).max() > 0 or np.abs(g).max() > 0):
            continue
        du = np.abs(key(g16) - key(r16))
        rl2 = np.sqrt(((g - r) ** 2).sum()) / max(np.sqrt((r ** 2).sum()), 1e-30)
        worst = np.argsort(-np.abs(g - r), axis=None)[:3]
        ws = [(int(i // g.shape[1]), int(i % g.shape[1]), float(g.flat[i]), float(r.flat[i])) for i in worst]
        print(f"  {tag:28s} ch{c} relL2 {rl2:.2e}  within2ulp {(du <= 2).mean() * 100:7.3f}%  ulp p99 {np.percentile(du, 99):.0f} p99.9 {np.percentile(du, 99.9):.0f} max {du.max()}  "
              f"maxabs {np.abs(g - r).max():.2e}  worst (y,x,got,ref) {ws}")


sd = helpers.scene_data(name)
ctx = hr.Context(0)
osc, gsc = oracle.Scene(sd), hr.Scene(ctx, sd)
scale = 0
frames = helpers.make_frames(oracle, osc, name, W, H, N, 1.5, scale_mips=0)
sob, sr = synth.blue_noise_tables()
sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
zbp = synth.z_buffer_params()
b16 = helpers.bits16


def fi_of(f):
    cur, prev = frames[f]["gb"], frames[f - 1 if f else 0]["gb"]
    return hr.frame_inputs(helpers.to_cuda(cur), helpers.to_cuda(prev), frames[f]["ubo"], f, f & 1, sob_d, sr_d, cur_full=helpers.to_cuda(cur), z_buffer_params=zbp)


if which == "shadows":
    pe, pf = hr.RayTracedShadows(ctx, W, H), hr.RayTracedShadows(ctx, W, H)
    pf.params.exact = 0
    for f in range(N):
        fi = fi_of(f)
        pe.render(gsc, fi); pf.render(gsc, fi); torch.cuda.synchronize()
        print(f"frame {f}: tiles differ {(pe.image(pe.IMG_TILES) != pf.image(pf.IMG_TILES)).sum().item()}")
        report("temporal (vis,var)", b16(pf.image(pf.IMG_TEMPORAL)), b16(pe.image(pe.IMG_TEMPORAL)))
        m = pe.IMG_MOMENTS1 if f & 1 else pe.IMG_MOMENTS0
        report("moments (m1,m2,len)", b16(pf.image(m))[..., :3], b16(pe.image(m))[..., :3])
        report("feedback image", b16(pf.image(pf.IMG_PREV)), b16(pe.image(pe.IMG_PREV)))
        report("a-trous output", b16(pf.output(hr.OUTPUT_ATROUS)), b16(pe.output(hr.OUTPUT_ATROUS)))
        # stage isolation: the fast a-trous chain on the EXACT temporal output would need a stage API on shared state; instead report
        # how the exact chain reacts to the fast temporal image is left to the per-stage numbers above
elif which == "ao":
    spp = 4
    pe, pf = hr.RayTracedAO(ctx, W, H, 0), hr.RayTracedAO(ctx, W, H, 0)
    pe.params.spp = pf.params.spp = spp
    pf.params.exact = 0
    for f in range(N):
        fi = fi_of(f)
        pe.render(gsc, fi); pf.render(gsc, fi); torch.cuda.synchronize()
        print(f"frame {f}: tiles differ {(pe.image(pe.IMG_TILES) != pf.image(pf.IMG_TILES)).sum().item()}")
        a = pe.IMG_AO1 if f & 1 else pe.IMG_AO0
        report("temporal AO", b16(pf.image(a)), b16(pe.image(a)))
        l = pe.IMG_LEN1 if f & 1 else pe.IMG_LEN0
        report("history length", b16(pf.image(l)), b16(pe.image(l)))
        if os.environ.get("HR_FUSE", "1") == "0":   # the fused X + Y launch of the tolerance mode keeps the X image in LDS: nothing to compare
            report("blur x", b16(pf.image(pf.IMG_BLUR0)), b16(pe.image(pe.IMG_BLUR0)))
        report("blur y", b16(pf.image(pf.IMG_BLUR1)), b16(pe.image(pe.IMG_BLUR1)))
elif which in ("ddgi", "reflections"):
    lo, hi = sd.bounds()
    ddgi = synth_env.ddgi_uniforms(lo, hi, probe_counts=(5, 3, 4), rays_per_probe=64, normal_bias=0.1)
    sky = synth_env.sky_cubemap(16)
    pre, lut = synth_env.prefiltered_chain(sky, 5), synth_env.brdf_lut(16)
    f16 = lambda a: torch.from_numpy(a).cuda().view(torch.float16)
    env = api_gi.environment(f16(sky), f16(pre), 16, 5, f16(lut))
    ge, gf = api_gi.DDGI(ctx, W, H, ddgi), api_gi.DDGI(ctx, W, H, ddgi)
    gf.params.exact = 0
    re_, rf = api_reflections.RayTracedReflections(ctx, W, H, 0), api_reflections.RayTracedReflections(ctx, W, H, 0)
    rf.params.exact = 0
    rng = np.random.RandomState(7)
    for f in range(N):
        fi = fi_of(f)
        orient = synth_env.random_orientation(rng)
        ge.render(gsc, fi, env, orient); gf.render(gsc, fi, env, orient)
        for r_ in (re_, rf):
            r_.set_camera_delta((0.0, 0.0, 0.0) if f == 0 else (-1.5, 0.0, 0.0))
        re_.render(gsc, fi, env, ge); rf.render(gsc, fi, env, ge)      # both read the exact pass's atlases
        torch.cuda.synchronize()
        print(f"frame {f}: reflection tiles differ {(re_.image(re_.IMG_TILES) != rf.image(rf.IMG_TILES)).sum().item()}")
        report("ddgi sample", b16(gf.output())[..., :3], b16(ge.output())[..., :3])
        c = re_.IMG_COLOR1 if f & 1 else re_.IMG_COLOR0
        report("refl temporal", b16(rf.image(c)), b16(re_.image(c)))
        m = re_.IMG_MOMENTS1 if f & 1 else re_.IMG_MOMENTS0
        report("refl moments", b16(rf.image(m))[..., :3], b16(re_.image(m))[..., :3])
        report("refl a-trous", b16(rf.output(hr.OUTPUT_ATROUS)), b16(re_.output(hr.OUTPUT_ATROUS)))
