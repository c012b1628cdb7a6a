"""CPU tool (oracle only, no GPU): how far does the REFERENCE's a-trous chain spread a perturbation of its input that the tolerance rule allows?
For one configuration of tools/fuzz_tolerance.py (same draws) the oracle renders the reflections sequence; on the chosen frame the temporal image is
perturbed by +-1 fp16 ulp on a random fraction of its values (default 0.4 %: what the tolerance-mode temporal kernel differs by on such frames, see
tools/refl_outlier_probe.py) and the oracle's own a-trous chain runs on both.  Printed: the share of OUTPUT values that moved by more than 2 fp16 ulp — the
figure the end-to-end population bound (>= 99.9 % within 2 ulp) would have to absorb with perfectly faithful a-trous kernels.
    python tools/refl_atrous_conditioning.py <seed> <trial> <frame> [fraction] [repeats]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from hybrid_rendering_amd import synth, synth_env
from oracle import pyoracle as oracle, pyoracle_ddgi as od, pyoracle_reflections as orf

seed, want, fwant = (int(v) for v in sys.argv[1:4])
frac = float(sys.argv[4]) if len(sys.argv) > 4 else 0.004
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
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
sd = helpers.scene_data(name)
osc = oracle.Scene(sd)
lo, hi = sd.bounds()
ddgi = synth_env.ddgi_uniforms(lo, hi, probe_counts=(5, 3, 4), rays_per_probe=64, normal_bias=0.1)
sky = synth_env.sky_cubemap(16)
pre, lut = synth_env.prefiltered_chain(sky, 5), synth_env.brdf_lut(16)
env_np = dict(sky=sky, prefiltered=pre, pre_size=16, pre_levels=5, lut=lut)
frames = helpers.make_frames(oracle, osc, name, W, H, fwant + 1, dolly, scale_mips=scale)
r01, r003 = np.float16(0.1).view(np.uint16), np.float16(0.03).view(np.uint16)
for fr in frames:
    for g in [fr["gb"]] + fr.get("mips", [])[1:]:
        ch = g["gb3"][..., 0]
        ch[ch == r01] = r003
sob, sr = synth.blue_noise_tables()
w, h = W >> scale, H >> scale
o_ddgi = od.DDGIPass(ddgi)
op = orf.ReflectionsPass(w, h, **dict(rp or {}))
rr = np.random.RandomState(7)
for f in range(fwant + 1):
    lvl = (lambda fr: fr["mips"][scale] if scale else fr["gb"])
    cur, prev, full = lvl(frames[f]), lvl(frames[f - 1 if f else 0]), frames[f]["gb"]
    orient = synth_env.random_orientation(rr)
    o_ddgi.render(osc, frames[f]["ubo"], full, sky, orient, f)
    irr, dep = o_ddgi.current_read()
    op.render(osc, frames[f]["ubo"], ddgi, cur, prev, sob, sr, f, env_np, irr, dep, camera_delta=(0.0, 0.0, 0.0) if f == 0 else (-dolly, 0.0, 0.0),
              full=full if scale else None, ping_pong=bool(f & 1))
st, p = op.stages, op.p


def chain(img):
    for i in range(p["filter_iterations"]):
        img = orf.atrous(img, cur, st["tiles"], 1 << i, p["radius"], p["phi_color"], p["phi_normal"], p["sigma_depth"], p["approximate_with_ddgi"])
    return img


def ulps(a, b):
    k = lambda v: np.where(v & 0x8000, 0x8000 - (v & 0x7fff).astype(np.int32), 0x8000 + (v & 0x7fff).astype(np.int32))
    return np.abs(k(a.astype(np.int32)) - k(b.astype(np.int32)))


ref = chain(st["temporal"])
assert np.array_equal(ref, st["atrous"][-1])
print(f"config {want} {name} ({W}, {H}) {light} scale {scale} dolly {dolly:.2f} {'random' if rp else 'default'} params; frame {fwant}; a-trous image {w}x{h}, "
      f"{p['filter_iterations']} iterations, phi_color {p['phi_color']:.2f}")
var = st["temporal"][..., 3].view(np.float16).astype(np.float32)
geo = cur["depth"] != 1.0
print(f"  temporal variance channel: {100.0 * (var[geo] == 0).mean():.1f} % of the surface texels exactly 0, {100.0 * (var[geo] < 1e-6).mean():.1f} % below 1e-6, median {np.median(var[geo]):.2e}")
pr = np.random.RandomState(123)
for what, chans in (("colour channels", (0, 1, 2)), ("all four channels (colour + variance)", (0, 1, 2, 3))):
    out = []
    for _ in range(reps):
        t = st["temporal"].copy()
        m = np.zeros(t.shape, bool)
        m[..., list(chans)] = pr.random_sample(t[..., list(chans)].shape) < frac
        m &= geo[..., None]
        step = np.where(pr.random_sample(t.shape) < 0.5, 1, -1)
        v = t.astype(np.int32)
        v[m] = np.clip(v[m] + step[m], 0, 0x7bff)      # +-1 ulp on positive fp16 values (colour and variance are >= 0)
        moved = ulps(chain(v.astype(np.uint16)), ref)
        out.append(100.0 * (moved[..., :3] > 2).mean())
    print(f"  +-1 fp16 ulp on {100 * frac:.2f} % of the {what}: {np.mean(out):.3f} % of the output colour values move by more than 2 ulp (runs: {', '.join('%.3f' % o for o in out)})")
# the variance is m2 - m1^2 of two STORED fp16 moments (reflections_denoise_reprojection.comp): one ulp of a stored moment moves it by the fp16 spacing at m2,
# whatever its own magnitude — a large RELATIVE change where the variance is small.  That is what a 1-ulp difference in the moments history (tolerated) does.
m2 = st["moments"][..., 1]
spacing = (np.minimum(m2.astype(np.int32) + 1, 0x7bff).astype(np.uint16).view(np.float16).astype(np.float32) - m2.view(np.float16).astype(np.float32))
out = []
for _ in range(reps):
    t = st["temporal"].copy()
    m = (pr.random_sample(var.shape) < frac) & geo
    v = var.copy()
    v[m] = np.maximum(0.0, v[m] + np.where(pr.random_sample(int(m.sum())) < 0.5, 1.0, -1.0) * spacing[m])
    t[..., 3] = v.astype(np.float16).view(np.uint16)
    moved = ulps(chain(t), ref)
    out.append(100.0 * (moved[..., :3] > 2).mean())
print(f"  variance moved by the fp16 spacing of its second moment (one ulp of a stored m2) on {100 * frac:.2f} % of the texels: {np.mean(out):.3f} % of the output colour values "
      f"move by more than 2 ulp (runs: {', '.join('%.3f' % o for o in out)})")
