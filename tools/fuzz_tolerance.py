"""GPU developer tool: the tolerance-mode runners of tests/test_gpu_tolerance.py (exact = 0 against the oracle within the stated
tolerance — masks bit-exact, >= 99.9 % of the texels within 2 fp16 ulp, rel-L2 <= 1e-3) on random image sizes,
scenes, lights, camera speeds, resolution scales (full / half / quarter) and pass parameters; `frames` > the tests' 5-6 checks that the
bound holds once the temporal feedback has run to its steady state.   python tools/fuzz_tolerance.py [seed] [n_configs] [frames] [hard]
(a 4th argument "hard" draws the scenes from bench.py's hard-tier geometry — layered fabric, foliage cards — and adds its grazing sun)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hybrid_rendering_amd import api as hr
from oracle import pyoracle as oracle
import test_gpu_tolerance as tol

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n_frames = int(sys.argv[3]) if len(sys.argv) > 3 else 0
kw = dict(n_frames=n_frames) if n_frames else {}
hard = len(sys.argv) > 4 and sys.argv[4] == "hard"
import helpers
ctx = hr.Context(0)
bad = 0
for c in helpers.fuzz_configs(seed, n, hard):
    trial, name, W, H, light, dolly, scale, sp, ap, rp = (c[k] for k in ("trial", "name", "W", "H", "light", "dolly", "scale", "shadows", "ao", "reflections"))
    res = []
    for label, fn in (("shadows", lambda: tol.test_shadows_tolerance(oracle, hr, ctx, name, W, H, dolly, light, sp, **kw)),
                      ("ao", lambda: tol.test_ao_tolerance(oracle, hr, ctx, name, W, H, scale, c["ao_spp"], ap, **kw)),
                      ("reflections+ddgi", lambda: tol.test_reflections_and_ddgi_sample_tolerance(oracle, hr, ctx, name, W, H, min(scale, 1), dolly, rp, **kw))):
        try:
            fn()
            res.append(label + " ok")
        except AssertionError as e:
            msg = str(e).splitlines()[0][:160]
            # the runners' scene-coverage checks carry no "frame" message
            if "frame" in msg or "tile classes" in msg:
                res.append(label + " OUT OF TOLERANCE: " + msg); bad += 1
            else:
                res.append(label + " ok")
        except Exception as e:
            res.append(label + " ERROR: " + repr(e)[:160]); bad += 1
    print(trial, name, (W, H), light, "scale", scale, "dolly %.2f" % dolly, "random params" if sp else "default params", res, flush=True)
print("out of tolerance:", bad)
sys.exit(1 if bad else 0)
