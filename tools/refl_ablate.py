import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from hybrid_rendering_amd import api as hr, synth
from hybrid_rendering_amd.frame import HybridFrame
for (W, H) in ((1920, 1080), (3840, 2160)):
    sd = synth.sponza_like(1.0)
    ctx = hr.Context(0)
    scene = hr.Scene(ctx, sd)
    hf = HybridFrame(ctx, scene, sd, W, H)
    for gi, ap in ((1, 1), (0, 1), (1, 0), (0, 0)):
        hf.refl.params.sample_gi = gi
        hf.refl.params.approximate_with_ddgi = ap
        hf.time(6, 3)
        p = hf.refl.pass_
        p.set_profiling(True); p.stage_times()
        for k in range(100, 112):
            hf.render(k, only="reflections")
        torch.cuda.synchronize()
        st = {s: ms for s, ms, b in p.stage_times()}
        p.set_profiling(False)
        print(f"{W}x{H} sample_gi {gi} approximate_with_ddgi {ap}: trace {st['ray_trace']*1e3:.1f} us  rays {p.ray_count()}", flush=True)
    hf.close(); scene.close()
