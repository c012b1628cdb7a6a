"""GPU developer tool: ONE configuration of tools/fuzz_tolerance.py (same draws), the reflections + DDGI runner only.   python tools/fuzz_one.py <seed> <trial> [frames]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hybrid_rendering_amd import api as hr
from oracle import pyoracle as oracle
import test_gpu_tolerance as tol
import helpers
seed, want = int(sys.argv[1]), int(sys.argv[2])
kw = dict(n_frames=int(sys.argv[3])) if len(sys.argv) > 3 else {}
c = helpers.fuzz_config(seed, want)
print("config", want, c["name"], (c["W"], c["H"]), c["light"], "scale", c["scale"], "dolly %.2f" % c["dolly"], flush=True)
ctx = hr.Context(0)
try:
    tol.test_reflections_and_ddgi_sample_tolerance(oracle, hr, ctx, c["name"], c["W"], c["H"], min(c["scale"], 1), c["dolly"], c["reflections"], **kw)
    print("ok")
except AssertionError as e:
    print("FAILED:", str(e)[:600])
