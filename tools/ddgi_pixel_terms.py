"""CPU diagnostic (no GPU): the per-probe terms of sample_irradiance at ONE pixel of one frame of a tools/fuzz_tolerance.py configuration, from the
oracle (orc_ddgi_sample_pixel_terms).  Used in round 5 to find out what the few DDGI outlier pixels of the tolerance mode have in common.
    python tools/ddgi_pixel_terms.py <seed> <trial> <frame> <y> <x>"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from hybrid_rendering_amd import synth_env
from oracle import pyoracle as oracle, pyoracle_ddgi as od

seed, trial_want, frame_want, py, px = (int(v) for v in sys.argv[1:6])
rng = np.random.RandomState(seed)
for trial in range(trial_want + 1):     # the draws of tools/fuzz_tolerance.py, in its order
    name = str(rng.choice(["cornell", "sponza_small"]))
    W, H = int(rng.randint(160, 360)), int(rng.randint(120, 220))
    light = str(rng.choice(["default", "point", "spot"]) if name != "cornell" else rng.choice(["default", "soft"]))
    dolly = float(rng.uniform(0.2, 2.5))
    scale = int(rng.choice([0, 1, 1, 2]))
    if trial % 2:
        [rng.uniform(0.005, 0.3), rng.uniform(0.05, 0.5), rng.uniform(1, 20), rng.choice([8.0, 32.0, 64.0, 12.5]), rng.uniform(0.3, 3), rng.choice([0.0, 1.2, 2.0]), rng.choice([1, 2]), rng.choice([1, 3, 5]), rng.choice([0, 1])]
        [rng.choice([2, 4, 6]), rng.uniform(0.005, 0.3), rng.uniform(5, 60)]
        [rng.uniform(0.005, 0.3), rng.uniform(0.05, 0.5), rng.uniform(1, 20), rng.choice([32.0, 8.0, 12.5]), rng.uniform(0.3, 3), rng.choice([1, 2]), rng.choice([1, 3, 5]), rng.choice([0, 1])]
    rng.randint(1, 5)                   # the AO runner's spp
print("config", trial, name, (W, H), light, "scale", scale, "dolly %.2f" % dolly)
scale = min(scale, 1)
sd = helpers.scene_data(name)
osc = oracle.Scene(sd)
lo, hi = sd.bounds()
ddgi = synth_env.ddgi_uniforms(lo, hi, probe_counts=(5, 3, 4), rays_per_probe=64, normal_bias=0.1)
sky = synth_env.sky_cubemap(16)
frames = helpers.make_frames(oracle, osc, name, W, H, frame_want + 1, dolly, scale_mips=scale)
o_ddgi = od.DDGIPass(ddgi)
r = np.random.RandomState(7)
for f in range(frame_want + 1):
    orient = synth_env.random_orientation(r)
    o_ddgi.render(osc, frames[f]["ubo"], frames[f]["gb"], sky, orient, f)
irr, dep = o_ddgi.current_read()
full = frames[frame_want]["gb"]
out, pnw = np.zeros((8, 16), np.float32), np.zeros(9, np.float32)
fp = C.POINTER(C.c_float)
oracle.lib().orc_ddgi_sample_pixel_terms(oracle._ubo_ptr(frames[frame_want]["ubo"]), od._ddgi_ptr(ddgi), C.c_int(W), C.c_int(H), C.c_int(px), C.c_int(py),
                                         full["depth"].ctypes.data_as(fp), full["gb2"].ctypes.data_as(C.POINTER(C.c_uint16)), dep.ctypes.data_as(C.POINTER(C.c_uint16)),
                                         out.ctypes.data_as(fp), pnw.ctypes.data_as(fp))
np.set_printoptions(precision=9, suppress=False, linewidth=200)
print("P", pnw[:3], "N", pnw[3:6], "Wo", pnw[6:9], "depth", full["depth"][py, px])
print("output (oracle)", o_ddgi.stages["output"][py, px].view(np.float16))
print("probe: dist, mean, m2, variance, dist-mean, vis, weight, final weight")
for i in range(8):
    print(i, out[i, :8], "atlas xy", out[i, 8:10], "texels", out[i, 10:14], " var/mean^2 %.3e  (dist-mean)/dist %.3e" % (out[i, 3] / max(out[i, 1] ** 2, 1e-30), out[i, 4] / max(out[i, 0], 1e-30)))
