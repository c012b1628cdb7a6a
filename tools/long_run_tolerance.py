"""GPU developer tool: does the tolerance mode (exact = 0) DRIFT over long runs?  The temporal stages feed their own fp16 output back with
alpha = 0.01, so a rounding difference survives ~1 / alpha frames.  Runs the tolerance tests' runners (tests/test_gpu_tolerance.py: every
frame of every image against the oracle, 2 fp16 ulp on >= 99.9 % of the texels, rel-L2 <= 1e-3) for many frames with a static / slowly
moving camera, where the history is never reset.   python tools/long_run_tolerance.py [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hybrid_rendering_amd import api as hr
from oracle import pyoracle as oracle
import test_gpu_tolerance as tol

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = hr.Context(0)
bad = 0
for name, w, h, dolly, light in (("cornell", 320, 184, 0.0, "soft"), ("cornell", 320, 184, 0.02, "default"), ("sponza_small", 288, 168, 0.03, "default"), ("sponza_small", 288, 168, 0.0, "point")):
    for label, fn in (("shadows", lambda: tol.test_shadows_tolerance(oracle, hr, ctx, name, w, h, dolly, light, None, n_frames=n)),
                      ("ao 2 spp", lambda: tol.test_ao_tolerance(oracle, hr, ctx, name, w, h, 0, 2, None, n_frames=n)),
                      ("reflections + ddgi", lambda: tol.test_reflections_and_ddgi_sample_tolerance(oracle, hr, ctx, name, w, h, 0, dolly, None, n_frames=min(n, 40)))):
        try:
            fn()
            print(name, (w, h), "dolly", dolly, light, label, f"{n} frames ok", flush=True)
        except AssertionError as e:
            msg = str(e)
            if msg.lstrip().startswith("frame"):
                bad += 1
                print(name, (w, h), "dolly", dolly, light, label, "OUT OF TOLERANCE:", msg[:300], flush=True)
            else:
                print(name, (w, h), "dolly", dolly, light, label, f"{n} frames ok (coverage check n/a)", flush=True)
print("out of tolerance:", bad)
