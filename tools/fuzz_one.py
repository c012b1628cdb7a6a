"""GPU developer tool: ONE configuration of tools/fuzz_tolerance.py (same draws), the reflections + DDGI runner only.   python tools/fuzz_one.py <seed> <trial> [frames]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hybrid_rendering_amd import api as hr
from oracle import pyoracle as oracle
import test_gpu_tolerance as tol
seed, want = int(sys.argv[1]), int(sys.argv[2])
kw = dict(n_frames=int(sys.argv[3])) if len(sys.argv) > 3 else {}
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
print("config", trial, name, (W, H), light, "scale", scale, "dolly %.2f" % dolly, flush=True)
ctx = hr.Context(0)
try:
    tol.test_reflections_and_ddgi_sample_tolerance(oracle, hr, ctx, name, W, H, min(scale, 1), dolly, rp, **kw)
    print("ok")
except AssertionError as e:
    print("FAILED:", str(e)[:600])
