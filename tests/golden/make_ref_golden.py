"""The golden cases of make_golden.py computed with the REFERENCE'S OWN SHADERS (oracle/refshim + oracle/ref_harness.py)
instead of the oracle restatement.  tests/test_ref_shaders.py::test_golden_fixtures_are_reference_shader_outputs checks
that this reproduces the committed tests/golden/*.npz bit for bit — so the fixtures the GPU tests compare the HIP
kernels with (tests/test_gpu_golden.py, on a box without /root/reference) ARE outputs of the reference's shaders.

Only the inputs come from elsewhere: the synthetic scene and G-buffer (helpers.make_frames), the BVH answer to each ray
(the oracle's pinned watertight test — the reference's traversal is the Vulkan driver) and the parameter defaults.

    python tests/golden/make_ref_golden.py        # prints which arrays match the committed fixtures
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build_cases():
    import helpers
    from oracle import ref_harness as rh
    from make_golden import hdr_color
    from hybrid_rendering_amd import synth, synth_env
    from oracle import pyoracle as po, pyoracle_post as opost
    out = {}
    sob, sr = synth.blue_noise_tables()
    # ---- shadows + AO: Cornell-32, 64x64, soft light, 3 frames ----------------------------------------------------
    sd = helpers.scene_data("cornell")
    sc = po.Scene(sd)
    w = h = 64
    frames = helpers.make_frames(po, sc, "cornell", w, h, 3, 1.5, "soft")
    sp = rh.RefShadowsPass(w, h)
    for f in range(3):
        sp.render(sc, frames[f]["ubo"], frames[f]["gb"], frames[f - 1]["gb"] if f else frames[f]["gb"], sob, sr, f)
    out["shadows_cornell64"] = dict(gb2=frames[2]["gb"]["gb2"], depth=frames[2]["gb"]["depth"], mask=sp.stages["mask"], temporal=sp.stages["temporal"],
                                    moments=sp.stages["moments"], tiles=sp.stages["tiles"], output=sp.stages["output"])
    zbp = synth.z_buffer_params()
    ap = rh.RefAOPass(w // 2, h // 2, zbp)
    for f in range(3):
        cur, prev = helpers.nearest_mip(frames[f]["gb"], 1), helpers.nearest_mip(frames[f - 1]["gb"] if f else frames[f]["gb"], 1)
        ap.render(sc, frames[f]["ubo"], cur, prev, sob, sr, f)
    up = rh.upsample("ao/ao_upsample.comp", [frames[2]["gb"], helpers.nearest_mip(frames[2]["gb"], 1)], 1, ap.stages["blur1"], "r16f", power=ap.p["power"])
    out["ao_cornell64_half"] = dict(mask=ap.stages["mask"][None], temporal=ap.stages["temporal"], blur1=ap.stages["blur1"], output=up)
    # ---- DDGI + reflections: small Sponza, 48x32, 2 frames --------------------------------------------------------
    sd2 = helpers.scene_data("sponza_small")
    sc2 = po.Scene(sd2)
    w2, h2 = 48, 32
    fr2 = helpers.make_frames(po, sc2, "sponza_small", w2, h2, 2, 1.0)
    lo, hi = sd2.bounds()
    ddgi = synth_env.ddgi_uniforms(lo, hi, probe_counts=(3, 2, 3), rays_per_probe=32, normal_bias=0.1)
    sky = synth_env.sky_cubemap(8)
    env = dict(sky=sky, prefiltered=synth_env.prefiltered_chain(sky, 4), pre_size=8, pre_levels=4, lut=synth_env.brdf_lut(8))
    dp, rp = rh.RefDDGIPass(ddgi, sd2), rh.RefReflectionsPass(w2, h2, sd2)
    rng = np.random.RandomState(3)
    for f in range(2):
        dp.render(sc2, fr2[f]["ubo"], fr2[f]["gb"], sky, synth_env.random_orientation(rng), f)
        irr, dep = dp.current_read()
        rp.render(sc2, fr2[f]["ubo"], ddgi, fr2[f]["gb"], fr2[f - 1]["gb"] if f else fr2[f]["gb"], sob, sr, f, env, irr, dep,
                  camera_delta=(-1.0, 0, 0) if f else (0, 0, 0))
    out["ddgi_sponza"] = dict(radiance=dp.stages["radiance"], direction_distance=dp.stages["direction_distance"], irradiance=dp.stages["irradiance"],
                              depth=dp.stages["depth"], output=dp.stages["output"])
    out["reflections_sponza"] = dict(trace=rp.stages["trace"], temporal=rp.stages["temporal"], tiles=rp.stages["tiles"], output=rp.stages["output"])
    # ---- ground-truth accumulator (3 frames) + TAA (2 frames) ------------------------------------------------------
    rsc = rh.RefScene(sd2)
    imgs, pp = [np.zeros((h2, w2, 4), np.uint16) for _ in range(2)], False
    for f in range(3):
        if f == 0:
            pp = False
        imgs[int(not pp)] = rh.ground_truth(sc2, rsc, fr2[0]["ubo"], sky, w2, h2, f, imgs[int(pp)])
        pp = not pp
    out["ground_truth_sponza"] = dict(output=imgs[int(pp)])
    taa = opost.TAAPass(w2, h2, reset=False)            # jitter bookkeeping (temporal_aa.cpp:64-81) only
    timg = [np.zeros((h2, w2, 4), np.uint16) for _ in range(2)]
    for f in range(2):
        jit = taa.update(f).copy()
        timg[f & 1] = rh.taa_resolve(hdr_color(fr2[f]["gb"]), timg[1 - (f & 1)], fr2[f]["gb"], jit, taa.feedback_min, taa.feedback_max, taa.sharpen)
    out["taa_sponza"] = dict(jitter=jit, output=timg[1])
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    for name, arrs in build_cases().items():
        gold = np.load(os.path.join(here, name + ".npz"))
        for k, v in arrs.items():
            same = gold[k].shape == v.shape and np.array_equal(gold[k].view(np.uint8), np.ascontiguousarray(v).view(np.uint8))
            print("%-22s %-20s %s" % (name, k, "== committed fixture" if same else "DIFFERS"))
