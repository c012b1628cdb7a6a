"""GPU developer tool: the parity runners of tests/test_gpu_*.py (HIP through the C ABI vs the oracle, every stage image
bit for bit) on random image sizes / scenes / lights / resolution scales.   python tools/fuzz_gpu.py [seed] [n_configs]"""
import os, sys, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hybrid_rendering_amd import api as hr
from oracle import pyoracle as oracle
import test_gpu_shadows, test_gpu_ao, test_gpu_ddgi, test_gpu_reflections

rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = hr.Context(0)
bad = 0
for trial in range(n):
    name = str(rng.choice(["cornell", "sponza_small"]))
    W, H = int(rng.randint(9, 200)), int(rng.randint(9, 140))
    kind = str(rng.choice(["default", "point", "spot"]) if name != "cornell" else rng.choice(["default", "soft"]))
    dolly = float(rng.uniform(0, 2.0))
    scale = int(rng.choice([0, 1, 2]))
    # every other configuration also draws the GUI parameters at random (the kernels' specialised fast paths — phi_normal 32,
    # sigma_depth 1, radius 1 — and their generic paths)
    sp = ap = rp = dpar = None
    if trial % 2:
        sp = dict(bias=float(rng.uniform(0.05, 1.0)), alpha=float(rng.uniform(0.005, 0.3)), moments_alpha=float(rng.uniform(0.05, 0.5)),
                  phi_visibility=float(rng.uniform(1, 20)), phi_normal=float(rng.choice([8.0, 32.0, 64.0, 12.5, 128.0])), sigma_depth=float(rng.uniform(0.2, 3)),
                  power=float(rng.choice([0.0, 1.2, 2.0, 0.7])), radius=int(rng.choice([1, 2])), filter_iterations=int(rng.choice([1, 3, 5])),
                  feedback_iteration=int(rng.choice([0, 1])))
        ap = dict(bias=float(rng.uniform(0.05, 1.0)), ray_length=float(rng.uniform(5, 100)), alpha=float(rng.uniform(0.005, 0.3)), blur_radius=int(rng.choice([1, 2, 4, 6])))
        rp = dict(sample_gi=bool(rng.randint(2)), approximate_with_ddgi=bool(rng.randint(2)), gi_intensity=float(rng.uniform(0.1, 1)),
                  rough_ddgi_intensity=float(rng.uniform(0.1, 1)), ibl_indirect_specular_intensity=float(rng.uniform(0, 0.2)), bias=float(rng.uniform(0.05, 1)),
                  trim=float(rng.uniform(0.3, 1)), alpha=float(rng.uniform(0.005, 0.3)), moments_alpha=float(rng.uniform(0.05, 0.5)),
                  blur_as_input=bool(rng.randint(2)), phi_color=float(rng.uniform(1, 20)), phi_normal=float(rng.choice([32.0, 8.0, 12.5, 128.0])),
                  sigma_depth=float(rng.uniform(0.2, 3)), radius=int(rng.choice([1, 2])), filter_iterations=int(rng.choice([1, 3, 5])),
                  feedback_iteration=int(rng.choice([0, 1])))
        dpar = dict(infinite_bounces=bool(rng.randint(2)), infinite_bounce_intensity=float(rng.uniform(0.5, 2.0)), gi_intensity=float(rng.uniform(0.3, 2.0)))
    res = []
    for label, fn in (("shadows", lambda: test_gpu_shadows._run_case(oracle, hr, ctx, name, W, H, 3, dolly, light_kind=kind, params=sp)),
                      ("ao", lambda: test_gpu_ao._run_case(oracle, hr, ctx, name, W, H, scale, 3, dolly, params=ap)),
                      ("ddgi", lambda: test_gpu_ddgi._run(oracle, hr, ctx, name, W, H, (3, 2, 3), 24, 2, light_kind=kind, params=dpar)),
                      ("reflections", lambda: test_gpu_reflections._run(oracle, hr, ctx, name, W, H, min(scale, 1), 2, dolly, params=rp, counts=(3, 2, 3)))):
        try:
            fn()
            res.append(label + " ok")
        except Exception as e:
            if not isinstance(e, AssertionError):
                res.append(label + " ERROR: " + repr(e)[:120]); bad += 1
                continue
            if not str(e).strip():                        # the runners' own scene-coverage checks carry no message: not a parity failure
                res.append(label + " ok")
            else:
                res.append(label + " MISMATCH: " + str(e)[:100])
                bad += 1
    print(trial, name, (W, H), kind, "scale", scale, "dolly %.2f" % dolly, "random params" if sp else "default params", res, flush=True)
print("mismatches:", bad)
