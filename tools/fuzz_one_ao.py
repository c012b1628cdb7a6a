"""GPU developer tool: the AO runner of ONE configuration of tools/fuzz_tolerance.py (same draws), with a per-stage report of the texels beyond 2 fp16 ulp.
    python tools/fuzz_one_ao.py <seed> <trial> [frames]"""
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
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
c = helpers.fuzz_config(seed, want)
name, W, H, scale, spp, params = c["name"], c["W"], c["H"], c["scale"], c["ao_spp"], c["ao"]
print("config", want, name, (W, H), "scale", scale, "spp", spp, params, flush=True)
ctx = hr.Context(0)
sd = helpers.scene_data(name)
osc, gsc = oracle.Scene(sd), hr.Scene(ctx, sd)
frames = helpers.make_frames(oracle, osc, name, W, H, n, 1.5, scale_mips=scale)
sob, sr, sob_d, sr_d = tol._tables()
zbp = synth.z_buffer_params()
w, h = W >> scale, H >> scale
kw = dict(params or {})
gp, op = hr.RayTracedAO(ctx, W, H, scale), oracle.AOPass(w, h, spp=spp, zbp=zbp, **kw)
for k, v in kw.items():
    setattr(gp.params, k, v)
gp.params.spp, gp.params.exact = spp, 0


def ulps(a, b):
    return np.abs(tol._key(a).astype(np.int64) - tol._key(b).astype(np.int64))


for f in range(n):
    lvl = (lambda fr: fr["mips"][scale] if scale else fr["gb"])
    cur, prev, full = lvl(frames[f]), lvl(frames[f - 1 if f else 0]), frames[f]["gb"]
    op.render(osc, frames[f]["ubo"], cur, prev, sob, sr, f, full=full if scale else None)
    gp.render(gsc, hr.frame_inputs(helpers.to_cuda(cur), helpers.to_cuda(prev), frames[f]["ubo"], f, f & 1, sob_d, sr_d,
                                   cur_full=helpers.to_cuda(full) if scale else None, z_buffer_params=zbp))
    torch.cuda.synchronize()
    st = op.stages
    for what, got, ref in (("temporal", helpers.bits16(gp.image(gp.IMG_AO1 if f & 1 else gp.IMG_AO0)), st["temporal"]),
                           ("blur", helpers.bits16(gp.image(gp.IMG_BLUR1)), st["blur1"]),
                           ("output", helpers.bits16(gp.output(hr.OUTPUT_UPSAMPLE)), st["output"] if st["output"].ndim == 2 else st["output"][..., 0])):
        got = got.reshape(ref.shape)
        u = ulps(got, ref)
        bad = np.argwhere(u > 2)
        print(f"frame {f} {what:8s} shape {ref.shape}  > 2 ulp: {len(bad)} ({100.0 * len(bad) / u.size:.3f} %)  max {int(u.max())} ulp", flush=True)
        for y, x in bad[:12]:
            print("     ", (int(y), int(x)), "gpu", float(oracle.f16(got[y, x])), "oracle", float(oracle.f16(ref[y, x])), "ulp", int(u[y, x]))
    if scale:
        # the upsample kernel alone: the oracle's upsample of the GPU's own low-res image
        try:
            ref_up = op.upsample_of(helpers.bits16(gp.image(gp.IMG_BLUR1)).reshape(st["blur1"].shape), cur, full)
            u = ulps(helpers.bits16(gp.output(hr.OUTPUT_UPSAMPLE)).reshape(ref_up.shape), ref_up)
            print(f"frame {f} upsample kernel against the oracle's upsample of the same image: > 2 ulp {int((u > 2).sum())}, max {int(u.max())}")
        except AttributeError:
            pass
