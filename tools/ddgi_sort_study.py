"""GPU developer study: would regrouping the 256 rays of a DDGI probe into waves of similar LENGTH pay?
Needs a variant library built with  HR_CFLAGS="-DHR_TRACE_DIVERGENCE -DHR_DDGI_DUMP_STEPS" python -m hybrid_rendering_amd.build --variant ddgisteps
(the primary ray's node steps are written into the unused .w of the radiance image).  Prints the node steps a wave executes
(= its slowest lane) for: the kernel's ray order, the ideal regrouping (sorted by the true step count) and a regrouping by a
prediction from the PREVIOUS frame (per probe, an 8x8 octahedral map of the steps seen per world-space direction bin)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HR_LIBRARY"] = os.path.join(ROOT, "hybrid_rendering_amd", "variants", "libhybrid_rendering_amd.ddgisteps.so")
import torch
from hybrid_rendering_amd import api as hr, api_gi, synth, synth_env

sd = synth.sponza_like(1.0)
ctx = hr.Context(0)
scene = hr.Scene(ctx, sd)
W, H = 640, 360
light = synth.sponza_light()
cams = [synth.sponza_camera(W / H, frame=f, dolly=0.5) for f in range(4)]
lo, hi = sd.bounds()
u = synth_env.ddgi_uniforms(lo, hi, probe_counts=(16, 8, 16), rays_per_probe=256, normal_bias=0.1)
sky = synth_env.sky_cubemap(32)
env = api_gi.environment(torch.from_numpy(sky).cuda().view(torch.float16))
sob, sr = synth.blue_noise_tables()
sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
g = api_gi.DDGI(ctx, W, H, u)
rng = np.random.RandomState(1)
steps, dirs = [], []
for f in range(3):
    ubo = synth.make_ubo(cams[f + 1], cams[f], light)
    gb = scene.gbuffer(ubo, W, H)
    g.render(scene, hr.frame_inputs(gb, None, ubo, f, f & 1, sob_d, sr_d), env, synth_env.random_orientation(rng))
    torch.cuda.synchronize()
    rad = g.image(g.IMG_RADIANCE).float().cpu().numpy().reshape(2048, 256, 4)
    dd = g.image(g.IMG_DIRDIST).float().cpu().numpy().reshape(2048, 256, 4)
    steps.append(rad[..., 3].copy()); dirs.append(dd[..., :3].copy())


def wave_cost(s):      # s: [probes, 256] in wave order -> mean over waves of the slowest lane
    return s.reshape(2048, 4, 64).max(axis=2).mean()


def octbin(d, n=8):
    d = d / np.abs(d).sum(axis=-1, keepdims=True)
    x, y = d[..., 0].copy(), d[..., 1].copy()
    neg = d[..., 2] < 0
    x2 = (1 - np.abs(y)) * np.sign(x + 1e-30); y2 = (1 - np.abs(x)) * np.sign(y + 1e-30)
    x[neg], y[neg] = x2[neg], y2[neg]
    ix = np.clip(((x * 0.5 + 0.5) * n).astype(int), 0, n - 1); iy = np.clip(((y * 0.5 + 0.5) * n).astype(int), 0, n - 1)
    return iy * n + ix


for f in (1, 2):
    s = steps[f]
    print(f"frame {f}: node steps per ray mean {s.mean():.2f}  p99 {np.percentile(s, 99):.0f}  max {s.max():.0f}")
    print(f"  kernel order (64 consecutive Fibonacci indices per wave): executed per wave {wave_cost(s):6.2f}")
    print(f"  ideal regrouping (sorted by the true step count):          executed per wave {wave_cost(np.sort(s, axis=1)):6.2f}")
    for n in (4, 8, 16):
        pb, cb = octbin(dirs[f - 1], n), octbin(dirs[f], n)
        pred = np.zeros((2048, n * n)); cnt = np.zeros((2048, n * n))
        np.add.at(pred, (np.arange(2048)[:, None], pb), steps[f - 1]); np.add.at(cnt, (np.arange(2048)[:, None], pb), 1)
        mean_all = steps[f - 1].mean(axis=1, keepdims=True)
        pred = np.where(cnt > 0, pred / np.maximum(cnt, 1), mean_all)
        key = np.take_along_axis(pred, cb, axis=1)
        order = np.argsort(key, axis=1, kind="stable")
        print(f"  regrouped by last frame's {n}x{n} direction-bin means:            executed per wave {wave_cost(np.take_along_axis(s, order, axis=1)):6.2f}")
