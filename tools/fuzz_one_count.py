"""GPU developer tool: tools/fuzz_one.py with a count of the PIXELS beyond the hard cap for every compared image that has any (the counted allowance of the a-trous
chains is in pixels).   python tools/fuzz_one_count.py <seed> <trial>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hybrid_rendering_amd import api as hr
from oracle import pyoracle as oracle
import test_gpu_tolerance as tol
import helpers
seed, want = int(sys.argv[1]), int(sys.argv[2])
c = helpers.fuzz_config(seed, want)
print("config", want, c["name"], (c["W"], c["H"]), c["light"], "scale", c["scale"], "dolly %.2f" % c["dolly"], c["reflections"], flush=True)
orig = tol.compare16


def counting(got, ref, what, *a, **kw):
    g, r = got.view(np.float16).astype(np.float64), ref.view(np.float16).astype(np.float64)
    u, d = np.abs(tol._key(got) - tol._key(ref)), np.abs(g - r)
    beyond = (u > tol.CAP_ULPS) & (d > tol.CAP_ABS)
    for ch in kw.get("variance_channels", ()):
        beyond[..., ch] &= d[..., ch] > max(tol.CAP_ABS, tol.VARIANCE_FLOOR)
    ex = kw.get("exclude")
    if ex is not None:
        beyond &= ~(ex if beyond.ndim == 2 else ex[..., None])
    px = beyond.reshape(beyond.shape[0], beyond.shape[1], -1).any(axis=2)
    if px.any():
        n_pixels = beyond.shape[0] * beyond.shape[1]
        allowed = int(max(4, kw.get("outlier_pixels", 0) * n_pixels) * kw.get("outlier_scale", 1)) if kw.get("outlier_pixels", 0) else 0
        print(f"[count] {what}: {int(px.sum())} pixels ({int(beyond.sum())} texels) beyond the cap, allowance {allowed}; worst {int(u[beyond].max())} ulp / {d[beyond].max():.3e}; at {np.argwhere(px)[:10].tolist()}", flush=True)
    try:
        return orig(got, ref, what, *a, **kw)
    except AssertionError as e:
        print("   FAILED:", str(e)[:200], flush=True)
        return 0.0, 0.0


tol.compare16 = counting
ctx = hr.Context(0)
tol.test_reflections_and_ddgi_sample_tolerance(oracle, hr, ctx, c["name"], c["W"], c["H"], min(c["scale"], 1), c["dolly"], c["reflections"])
print("done")
