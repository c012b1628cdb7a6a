"""Generates the committed golden fixtures from the CPU oracle (run from the repo root):

    python tests/golden/make_golden.py

The reference (a Vulkan application) cannot be built or run here, but its SHADERS can: make_ref_golden.py computes the
same cases with the reference's own shader sources through oracle/refshim, and tests/test_ref_shaders.py requires the
two to agree bit for bit — these vectors are therefore reference-shader outputs; the oracle and the HIP path
are checked against them.  One small multi-frame case per pass; everything is stored as exact bit patterns.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def build_cases():
    """Returns {name: {array_name: np.ndarray}} — shared by the generator and tests/test_golden.py."""
    import helpers
    from hybrid_rendering_amd import synth, synth_env
    from oracle import pyoracle as po, pyoracle_ddgi as od, pyoracle_reflections as orf
    out = {}
    sob, sr = synth.blue_noise_tables()
    # ---- shadows: Cornell-32, 64x64, soft light, 3 frames with a moving camera ------------------------------
    sd = helpers.scene_data("cornell")
    sc = po.Scene(sd)
    w = h = 64
    frames = helpers.make_frames(po, sc, "cornell", w, h, 3, 1.5, "soft")
    sp = po.ShadowsPass(w, h)
    for f in range(3):
        sp.render(sc, frames[f]["ubo"], frames[f]["gb"], frames[f - 1]["gb"] if f else frames[f]["gb"], sob, sr, f)
    out["shadows_cornell64"] = dict(gb2=frames[2]["gb"]["gb2"], depth=frames[2]["gb"]["depth"], mask=sp.stages["mask"], temporal=sp.stages["temporal"],
                                    moments=sp.stages["moments"], tiles=sp.stages["tiles"], output=sp.stages["output"])
    # ---- AO: same frames, half resolution + upsample ------------------------------------------------------------
    ap = po.AOPass(w // 2, h // 2, zbp=synth.z_buffer_params())
    for f in range(3):
        cur, prev = helpers.nearest_mip(frames[f]["gb"], 1), helpers.nearest_mip(frames[f - 1]["gb"] if f else frames[f]["gb"], 1)
        ap.render(sc, frames[f]["ubo"], cur, prev, sob, sr, f, full=frames[f]["gb"])
    out["ao_cornell64_half"] = dict(mask=ap.stages["mask"], temporal=ap.stages["temporal"], blur1=ap.stages["blur1"], output=ap.stages["output"])
    # ---- DDGI + reflections: small Sponza, 48x32, 2 frames ------------------------------------------------------
    sd2 = helpers.scene_data("sponza_small")
    sc2 = po.Scene(sd2)
    w2, h2 = 48, 32
    fr2 = helpers.make_frames(po, sc2, "sponza_small", w2, h2, 2, 1.0)
    lo, hi = sd2.bounds()
    ddgi = synth_env.ddgi_uniforms(lo, hi, probe_counts=(3, 2, 3), rays_per_probe=32, normal_bias=0.1)
    sky = synth_env.sky_cubemap(8)
    env = dict(sky=sky, prefiltered=synth_env.prefiltered_chain(sky, 4), pre_size=8, pre_levels=4, lut=synth_env.brdf_lut(8))
    dp, rp = od.DDGIPass(ddgi), orf.ReflectionsPass(w2, h2)
    rng = np.random.RandomState(3)
    for f in range(2):
        dp.render(sc2, fr2[f]["ubo"], fr2[f]["gb"], sky, synth_env.random_orientation(rng), f)
        irr, dep = dp.current_read()
        rp.render(sc2, fr2[f]["ubo"], ddgi, fr2[f]["gb"], fr2[f - 1]["gb"] if f else fr2[f]["gb"], sob, sr, f, env, irr, dep, camera_delta=(-1.0, 0, 0) if f else (0, 0, 0))
    out["ddgi_sponza"] = dict(radiance=dp.stages["radiance"], direction_distance=dp.stages["direction_distance"], irradiance=dp.stages["irradiance"],
                              depth=dp.stages["depth"], output=dp.stages["output"])
    out["reflections_sponza"] = dict(trace=rp.stages["trace"], temporal=rp.stages["temporal"], tiles=rp.stages["tiles"], output=rp.stages["output"])
    # ---- ground-truth accumulator (3 frames) + TAA (2 frames, real history) on the small Sponza frames --------------
    from oracle import pyoracle_post as opost
    gt = opost.GroundTruthPass(w2, h2)
    for f in range(3):
        gt_out = gt.render(sc2, fr2[0]["ubo"], sky).copy()
    out["ground_truth_sponza"] = dict(output=gt_out, rays=np.array([gt.rays], np.uint64))
    taa = opost.TAAPass(w2, h2, reset=False)
    for f in range(2):
        jit = taa.update(f).copy()
        taa.render(hdr_color(fr2[f]["gb"]), fr2[f]["gb"], f & 1)
    out["taa_sponza"] = dict(jitter=jit, output=taa.output(1).copy())
    return out


def hdr_color(gb):
    """A deterministic HDR stand-in for the deferred composite: albedo x 1.7 where geometry, 0.25 on the sky."""
    c = np.zeros(gb["gb1"].shape[:2] + (4,), np.float32)
    c[..., :3] = gb["gb1"][..., :3].astype(np.float32) / 255.0 * 1.7
    c[gb["depth"] == 1.0, :3] = 0.25
    c[..., 3] = 1.0
    return np.ascontiguousarray(c.astype(np.float16)).view(np.uint16)


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    cases = build_cases()
    index = {}
    for name, arrs in cases.items():
        np.savez_compressed(os.path.join(here, name + ".npz"), **arrs)
        index[name] = {k: sha(v) for k, v in arrs.items()}
        print(name, {k: (v.shape, str(v.dtype)) for k, v in arrs.items()})
    import json
    json.dump(index, open(os.path.join(here, "index.json"), "w"), indent=1, sort_keys=True)
