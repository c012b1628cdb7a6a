"""GPU developer tool: one configuration of tools/fuzz_tolerance.py, reflections pass; prints the neighbourhood of a texel in the stage images of the
tolerance-mode kernels and of the oracle (where does a reflections a-trous outlier come from?).   python tools/refl_outlier_probe.py <seed> <trial> <frame> <y> <x>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import helpers
from hybrid_rendering_amd import api as hr, api_gi, api_reflections, synth_env
from oracle import pyoracle as oracle, pyoracle_ddgi as od, pyoracle_reflections as orf
import test_gpu_tolerance as tol
seed, want, fwant, py, px = (int(v) for v in sys.argv[1:6])
rng = np.random.RandomState(seed)
for trial in range(want + 1):
    name = str(rng.choice(["cornell", "sponza_small"]))
    W, H = int(rng.randint(160, 360)), int(rng.randint(120, 220))
    light = str(rng.choice(["default", "point", "spot"]) if name != "cornell" else rng.choice(["default", "soft"]))
    dolly = float(rng.uniform(0.2, 2.5))
    scale = int(rng.choice([0, 1, 1, 2]))
    rp = None
    if trial % 2:
        [rng.uniform(0.005, 0.3), rng.uniform(0.05, 0.5), rng.uniform(1, 20), rng.choice([8.0, 32.0, 64.0, 12.5]), rng.uniform(0.3, 3), rng.choice([0.0, 1.2, 2.0]), rng.choice([1, 2]), rng.choice([1, 3, 5]), rng.choice([0, 1])]
        [rng.choice([2, 4, 6]), rng.uniform(0.005, 0.3), rng.uniform(5, 60)]
        rp = dict(alpha=float(rng.uniform(0.005, 0.3)), moments_alpha=float(rng.uniform(0.05, 0.5)), phi_color=float(rng.uniform(1, 20)),
                  phi_normal=float(rng.choice([32.0, 8.0, 12.5])), sigma_depth=float(rng.uniform(0.3, 3)), radius=int(rng.choice([1, 2])),
                  filter_iterations=int(rng.choice([1, 3, 5])), feedback_iteration=int(rng.choice([0, 1])))
    rng.randint(1, 5)
scale = min(scale, 1)
print("config", trial, name, (W, H), light, "scale", scale, "dolly %.2f" % dolly, rp, flush=True)
ctx = hr.Context(0)
sd = helpers.scene_data(name)
osc, gsc = oracle.Scene(sd), hr.Scene(ctx, sd)
lo, hi = sd.bounds()
ddgi = synth_env.ddgi_uniforms(lo, hi, probe_counts=(5, 3, 4), rays_per_probe=64, normal_bias=0.1)
sky = synth_env.sky_cubemap(16)
pre, lut = synth_env.prefiltered_chain(sky, 5), synth_env.brdf_lut(16)
env_np = dict(sky=sky, prefiltered=pre, pre_size=16, pre_levels=5, lut=lut)
f16 = lambda a: torch.from_numpy(a).cuda().view(torch.float16)
env = api_gi.environment(f16(sky), f16(pre), 16, 5, f16(lut))
frames = helpers.make_frames(oracle, osc, name, W, H, fwant + 1, dolly, scale_mips=scale)
r01, r003 = np.float16(0.1).view(np.uint16), np.float16(0.03).view(np.uint16)
for fr in frames:
    for g in [fr["gb"]] + fr.get("mips", [])[1:]:
        ch = g["gb3"][..., 0]
        ch[ch == r01] = r003
sob, sr, sob_d, sr_d = tol._tables()
w, h = W >> scale, H >> scale
g_ddgi, o_ddgi = api_gi.DDGI(ctx, W, H, ddgi), od.DDGIPass(ddgi)
g_ddgi.params.exact = 0
gp = api_reflections.RayTracedReflections(ctx, W, H, scale)
kw = dict(rp or {})
for k, v in kw.items():
    setattr(gp.params, k, v)
gp.params.exact = 0
op = orf.ReflectionsPass(w, h, **kw)
rr = np.random.RandomState(7)
for f in range(fwant + 1):
    lvl = (lambda fr: fr["mips"][scale] if scale else fr["gb"])
    cur, prev, full = lvl(frames[f]), lvl(frames[f - 1 if f else 0]), frames[f]["gb"]
    orient = synth_env.random_orientation(rr)
    cam_delta = (0.0, 0.0, 0.0) if f == 0 else (-dolly, 0.0, 0.0)
    o_ddgi.render(osc, frames[f]["ubo"], full, sky, orient, f)
    irr, dep = o_ddgi.current_read()
    op.render(osc, frames[f]["ubo"], ddgi, cur, prev, sob, sr, f, env_np, irr, dep, camera_delta=cam_delta, full=full if scale else None, ping_pong=bool(f & 1))
    full_d = helpers.to_cuda(full)
    g_ddgi.render(gsc, hr.frame_inputs(full_d, None, frames[f]["ubo"], f, f & 1, sob_d, sr_d), env, orient)
    gp.set_camera_delta(cam_delta)
    gp.render(gsc, hr.frame_inputs(helpers.to_cuda(cur), helpers.to_cuda(prev), frames[f]["ubo"], f, f & 1, sob_d, sr_d, cur_full=full_d), env, g_ddgi)
    torch.cuda.synchronize()
st = op.stages
np.set_printoptions(precision=6, linewidth=220, suppress=True)
def nb(a, r=1):
    return a[max(py - r, 0):py + r + 1, max(px - r, 0):px + r + 1].view(np.float16).astype(np.float32)
f = fwant
imgs = [("trace", helpers.bits16(gp.image(gp.IMG_TRACE)), st["trace"]),
        ("temporal", helpers.bits16(gp.image(gp.IMG_COLOR1 if f & 1 else gp.IMG_COLOR0)), st["temporal"]),
        ("atrous out", helpers.bits16(gp.output(hr.OUTPUT_ATROUS)), st["atrous"][-1])]
for i, a in enumerate(st["atrous"][:-1]):
    imgs.append((f"oracle atrous {i}", a, a))
print("gb3 (roughness, curvature | id, z):", nb(cur["gb3"])[1, 1], " tiles got/ref", gp.image(gp.IMG_TILES).cpu().numpy()[py >> 3, px >> 3], st["tiles"][py >> 3, px >> 3])
for label, got, ref in imgs:
    g, r = nb(got), nb(ref)
    for c in range(4):
        d = np.abs(g[..., c] - r[..., c]).max()
        print(f"{label} ch{c}: max |got-ref| in 3x3 = {d:.3e}")
        print("  got", g[..., c].ravel())
        print("  ref", r[..., c].ravel())

# ---- whole-image view: where do the inputs of the a-trous chain differ, and is the chain itself faithful?
def ulps(a, b):
    k = lambda v: np.where(v & 0x8000, 0x8000 - (v & 0x7fff).astype(np.int32), 0x8000 + (v & 0x7fff).astype(np.int32))
    return np.abs(k(a.astype(np.int32)) - k(b.astype(np.int32)))
g_tmp = helpers.bits16(gp.image(gp.IMG_COLOR1 if f & 1 else gp.IMG_COLOR0))
for label, got, ref in imgs[:3]:
    u = ulps(got, ref)
    w_ = np.argwhere(u > 0)
    print(f"[whole image] {label}: {len(w_)} of {u.size} values differ, {int((u > 2).sum())} by more than 2 ulp, worst {int(u.max())} ulp; first: {w_[:8].tolist()}")
# the ORACLE's a-trous chain run on the GPU's temporal image: what is left against the GPU's a-trous output is the a-trous kernels' own arithmetic,
# what it differs by from the oracle's output is the reference's filter amplifying the (tolerated) input differences
p = op.p
img = g_tmp.copy()
for i in range(p["filter_iterations"]):
    img = orf.atrous(img, cur, st["tiles"], 1 << i, p["radius"], p["phi_color"], p["phi_normal"], p["sigma_depth"], p["approximate_with_ddgi"])
g_out = helpers.bits16(gp.output(hr.OUTPUT_ATROUS))
for label, a, b in (("GPU a-trous output vs ORACLE chain on the GPU's temporal image", g_out, img), ("ORACLE chain on the GPU's temporal image vs oracle output", img, st["atrous"][-1]),
                    ("GPU a-trous output vs oracle output", g_out, st["atrous"][-1])):
    u = ulps(a, b)
    print(f"[whole image] {label}: {int((u > 0).sum())} values differ, {int((u > 2).sum())} by more than 2 ulp ({100.0 * (u > 2).mean():.3f} %), worst {int(u.max())} ulp")

# ---- which temporal texels are off by more than 2 ulp, and what do their moments / history lengths say?
g_mom = helpers.bits16(gp.image(gp.IMG_MOMENTS1 if f & 1 else gp.IMG_MOMENTS0))
u = ulps(g_tmp, st["temporal"])
bad = np.argwhere((u > 2).any(axis=2))
f16 = lambda v: v.view(np.float16).astype(np.float32)
print(f"[temporal texels beyond 2 ulp] {len(bad)} texels")
for (yy, xx) in bad[:24]:
    print(f"  ({yy:3d},{xx:3d}) roughness {float(f16(cur['gb3'][yy, xx, 0:1])[0]):.3f} curv {float(f16(cur['gb3'][yy, xx, 1:2])[0]):.3f} ray_len {float(f16(st['trace'][yy, xx, 3:4])[0]):8.3f} | "
          f"got {np.round(f16(g_tmp[yy, xx]), 6).tolist()} ref {np.round(f16(st['temporal'][yy, xx]), 6).tolist()} | moments got {np.round(f16(g_mom[yy, xx, :3]), 5).tolist()} ref {np.round(f16(st['moments'][yy, xx, :3]), 5).tolist()}")
