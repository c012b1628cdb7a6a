"""CPU diagnostic (no GPU): how well conditioned is the weighted mean of sample_irradiance over a frame?  Per pixel the oracle reports sum_w (the
normaliser) and the largest probe weight before its trilinear factor; a tolerance-mode trilinear factor of ~1e-7 where the parity arithmetic has an
exact 0 (shading point on a probe plane) leaks max_w * 1e-7 into a sum of sum_w.   python tools/ddgi_conditioning.py [bench|small] [W H]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from hybrid_rendering_amd import synth, synth_env
from oracle import pyoracle as oracle, pyoracle_ddgi as od

which = sys.argv[1] if len(sys.argv) > 1 else "bench"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (480, 270)
if which == "bench":
    sd = synth.sponza_like(1.0)
    counts, rays = (16, 8, 16), 256
else:
    sd = helpers.scene_data("sponza_small")
    counts, rays = (5, 3, 4), 64
osc = oracle.Scene(sd)
lo, hi = sd.bounds()
ddgi = synth_env.ddgi_uniforms(lo, hi, probe_counts=counts, rays_per_probe=rays, normal_bias=0.1)
sky = synth_env.sky_cubemap(16)
light = synth.sponza_light()
cams = [synth.sponza_camera(W / H, frame=f, dolly=0.5) for f in range(4)]
o_ddgi = od.DDGIPass(ddgi)
r = np.random.RandomState(7)
for f in range(3):
    ubo = synth.make_ubo(cams[f + 1], cams[f], light)
    gb = osc.gbuffer(ubo, W, H)
    o_ddgi.render(osc, ubo, gb, sky, synth_env.random_orientation(r), f)
irr, dep = o_ddgi.current_read()
out = np.zeros((H, W, 2), np.float32)
fp = C.POINTER(C.c_float)
oracle.lib().orc_ddgi_sample_conditioning(oracle._ubo_ptr(ubo), od._ddgi_ptr(ddgi), C.c_int(W), C.c_int(H), gb["depth"].ctypes.data_as(fp),
                                          gb["gb2"].ctypes.data_as(C.POINTER(C.c_uint16)), dep.ctypes.data_as(C.POINTER(C.c_uint16)), out.ctypes.data_as(fp))
geo = out[..., 0] >= 0
sw, mx = out[..., 0][geo], out[..., 1][geo]
print(f"{which}: {geo.sum()} geometry pixels of {W}x{H}; depth atlas inf share {np.isinf(dep.view(np.float16).astype(np.float32)).mean():.3f}")
for thr in (1e-1, 1e-2, 1e-3, 1e-4, 1e-6, 1e-10):
    print(f"  sum_w < {thr:g}: {(sw < thr).mean() * 100:.3f} % of the geometry pixels;  leak-sensitive (max_w * 1e-7 > 1e-3 * sum_w): {((mx * 1e-7) > 1e-3 * sw).mean() * 100:.3f} %" if thr == 1e-1 else f"  sum_w < {thr:g}: {(sw < thr).mean() * 100:.3f} %")
