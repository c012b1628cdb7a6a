"""CPU developer tool: thread scaling of the oracle's any-hit replay (bench.py cpu_baseline.trace_replay).  OMP_NUM_THREADS=n python tools/cpu_replay.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybrid_rendering_amd import synth
from oracle import pyoracle as po
sd = synth.sponza_like(1.0)
osc = po.Scene(sd)
W, H = 1920, 1080
cam = synth.sponza_camera(W / H)
ubo = synth.make_ubo(cam, None, synth.sponza_light())
gb = osc.gbuffer(ubo, 480, 270)
sob, sr = synth.blue_noise_tables()
rays = po.shadows_gen_rays(ubo, gb["depth"], gb["gb2"], sob, sr, 0.5, 1)
rays = np.ascontiguousarray(rays[rays[:, 7] > 0]); rays[:, 7] = 0.01
rays = np.tile(rays, (8, 1))
osc.any_hit(rays[:1000])
t0 = time.perf_counter(); occ, st = osc.any_hit(rays, stats=True); dt = time.perf_counter() - t0
print("threads", os.environ.get("OMP_NUM_THREADS"), "rays", len(rays), "Mrays/s %.3f" % (len(rays) / dt / 1e6), "nodes/ray %.1f" % (st[0] / len(rays)))
