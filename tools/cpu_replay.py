"""CPU developer tool: thread scaling of the oracle's any-hit replay (bench.py cpu_baseline.trace_replay) on this host.
    python tools/cpu_replay.py [threads ...]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybrid_rendering_amd import synth
from oracle import pyoracle as po
sd = synth.sponza_like(1.0)
osc = po.Scene(sd)
W, H = 1920, 1080
cam = synth.sponza_camera(W / H)
ubo = synth.make_ubo(cam, None, synth.sponza_light())
gb = osc.gbuffer(ubo, 960, 540)
sob, sr = synth.blue_noise_tables()
rays = po.shadows_gen_rays(ubo, gb["depth"], gb["gb2"], sob, sr, 0.5, 1)
rays = np.ascontiguousarray(rays[rays[:, 7] > 0]); rays[:, 7] = 0.01
rays = np.tile(rays, (4, 1))
print("visible cpus", os.cpu_count(), "effective (affinity / cgroup quota)", po.effective_cpus(), "rays", len(rays))
for t in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32]:
    po.set_threads(t)
    best = 1e9
    t_all = time.perf_counter()
    for rep in range(6):
        t0 = time.perf_counter(); c0 = time.process_time()
        osc.any_hit(rays)
        dt, cpu = time.perf_counter() - t0, time.process_time() - c0
        best = min(best, dt)
    print("threads %3d: best %.3f s = %.2f Mrays/s  (last call: cpu/wall %.1f)" % (t, best, len(rays) / best / 1e6, cpu / dt))
