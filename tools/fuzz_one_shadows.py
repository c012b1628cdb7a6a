"""GPU developer tool: the shadows runner of ONE configuration of tools/fuzz_tolerance.py (same draws), with a per-stage report of the texels beyond 2 fp16 ulp.
    python tools/fuzz_one_shadows.py <seed> <trial> [frames]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from hybrid_rendering_amd import api as hr, synth
from oracle import pyoracle as oracle
import test_gpu_tolerance as tol
import helpers

seed, want = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
c = helpers.fuzz_config(seed, want)
name, w, h, dolly, light, params = c["name"], c["W"], c["H"], c["dolly"], c["light"], c["shadows"]
print("config", want, name, (w, h), light, "dolly %.2f" % dolly, params, flush=True)
ctx = hr.Context(0)
sd = helpers.scene_data(name)
osc, gsc = oracle.Scene(sd), hr.Scene(ctx, sd)
frames = helpers.make_frames(oracle, osc, name, w, h, n, dolly, light)
sob, sr, sob_d, sr_d = tol._tables()
kw = dict(params or {})
gp, op = hr.RayTracedShadows(ctx, w, h), oracle.ShadowsPass(w, h, **kw)
for k, v in kw.items():
    setattr(gp.params, k, v)
gp.params.exact = 0


def ulps(a, b):
    return np.abs(tol._key(a).astype(np.int64) - tol._key(b).astype(np.int64))


for f in range(n):
    cur, prev = frames[f]["gb"], frames[f - 1 if f else 0]["gb"]
    op.render(osc, frames[f]["ubo"], cur, prev, sob, sr, f)
    gp.render(gsc, hr.frame_inputs(helpers.to_cuda(cur), helpers.to_cuda(prev), frames[f]["ubo"], f, f & 1, sob_d, sr_d))
    torch.cuda.synchronize()
    st = op.stages
    tiles_g, tiles_o = gp.image(gp.IMG_TILES).cpu().numpy(), st["tiles"]
    print(f"frame {f}: tile classes differing: {int((tiles_g.reshape(-1)[:tiles_o.size] != tiles_o.reshape(-1)).sum())}")
    for what, got, ref in (("temporal", helpers.bits16(gp.image(gp.IMG_TEMPORAL)), st["temporal"]),
                           ("moments", helpers.bits16(gp.image(gp.IMG_MOMENTS1 if f & 1 else gp.IMG_MOMENTS0)), st["moments"]),
                           ("output", helpers.bits16(gp.output(hr.OUTPUT_ATROUS)), st["output"]),
                           ("feedback", helpers.bits16(gp.image(gp.IMG_PREV)), op.prev_image)):
        got = got.reshape(ref.shape)
        u = ulps(got, ref)
        d = np.abs(oracle.f16(got).astype(np.float64) - oracle.f16(ref).astype(np.float64))
        bad = np.argwhere((u > 32) & (d > 2.0 ** -10))
        print(f"  {what:9s} > 2 ulp: {int((u > 2).sum())} ({100.0 * (u > 2).mean():.3f} %)  max {int(u.max())} ulp / {d.max():.3e}; beyond the cap: {len(bad)}", flush=True)
        for idx in bad[:8]:
            t = tuple(int(v) for v in idx)
            y, x = t[0], t[1]
            print("       ", t, "gpu", oracle.f16(got[y, x]).tolist(), "oracle", oracle.f16(ref[y, x]).tolist(), "depth", float(cur["depth"][y, x]),
                  "neighbourhood temporal gpu/oracle", oracle.f16(helpers.bits16(gp.image(gp.IMG_TEMPORAL)).reshape(st["temporal"].shape)[y, x]).tolist(), oracle.f16(st["temporal"][y, x]).tolist())
