"""CPU developer tool: random image sizes / scenes / lights / camera motion / resolution scales, every pass (shadows, AO,
DDGI, reflections) on the oracle AND on the reference's own shaders (oracle/refshim), all stage images compared bit for
bit.   python tools/fuzz_ref_shaders.py [seed] [n_configs]      (needs /root/reference or a prebuilt oracle/_ref)"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from oracle import pyoracle as oracle, ref_harness as rh, pyoracle_ddgi as od, pyoracle_reflections as orf
from hybrid_rendering_amd import synth, synth_env
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
sob, sr = synth.blue_noise_tables(); zbp = synth.z_buffer_params()
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    name = rng.choice(["cornell", "sponza_small"])
    W, H = int(rng.randint(9, 140)), int(rng.randint(9, 100))
    kind = rng.choice(["default", "point", "spot"]) if name != "cornell" else rng.choice(["default", "soft"])
    dolly = float(rng.uniform(0, 2.0)) * (0.05 if name == "cornell" else 1.0)
    scale = int(rng.choice([0, 0, 1]))
    sd = helpers.scene_data(name); osc = oracle.Scene(sd)
    frames = helpers.make_frames(oracle, osc, name, W, H, 3, dolly, kind, scale_mips=max(scale, 1))
    w, h = W >> scale, H >> scale
    lo, hi = sd.bounds()
    ddgi = synth_env.ddgi_uniforms(lo, hi, probe_counts=(3, 2, 3), rays_per_probe=24, normal_bias=1.0 if name == "cornell" else 0.1)
    sky = synth_env.sky_cubemap(8)
    env = dict(sky=sky, prefiltered=synth_env.prefiltered_chain(sky, 4), pre_size=8, pre_levels=4, lut=synth_env.brdf_lut(8))
    op, rp = oracle.ShadowsPass(W, H), rh.RefShadowsPass(W, H)
    oa, ra = oracle.AOPass(w, h, zbp=zbp), rh.RefAOPass(w, h, zbp)
    if name != "cornell": oa.p["ray_length"] = ra.p["ray_length"] = 60.0
    dp, rdp = od.DDGIPass(ddgi), rh.RefDDGIPass(ddgi, sd)
    orp, rrp = orf.ReflectionsPass(w, h), rh.RefReflectionsPass(w, h, sd)
    r1, r2 = np.random.RandomState(trial), np.random.RandomState(trial)
    res = []
    for k, fr in enumerate(frames):
        lvl = (lambda f: f["mips"][scale] if scale else f["gb"])
        cur, prev, full = lvl(fr), lvl(frames[k-1] if k else fr), fr["gb"]
        op.render(osc, fr["ubo"], full, frames[k-1]["gb"] if k else full, sob, sr, k); rp.render(osc, fr["ubo"], full, frames[k-1]["gb"] if k else full, sob, sr, k)
        oa.render(osc, fr["ubo"], cur, prev, sob, sr, k, full=full if scale else None); ra.render(osc, fr["ubo"], cur, prev, sob, sr, k)
        dp.render(osc, fr["ubo"], full, sky, synth_env.random_orientation(r1), k); rdp.render(osc, fr["ubo"], full, sky, synth_env.random_orientation(r2), k)
        irr, dep = dp.current_read()
        cd = (0, 0, 0) if k == 0 else (-dolly, 0, 0)
        orp.render(osc, fr["ubo"], ddgi, cur, prev, sob, sr, k, env, irr, dep, camera_delta=cd, full=full if scale else None)
        rrp.render(osc, fr["ubo"], ddgi, cur, prev, sob, sr, k, env, irr, dep, camera_delta=cd, full_mips=fr["mips"][:scale+1] if scale else None)
        a, b = op.stages, rp.stages
        n = int((a["mask"] != b["mask"]).sum()) + int((a["tiles"] != b["tiles"]).sum()) + int((a["temporal"] != b["temporal"]).sum()) + sum(int((x != y).sum()) for x, y in zip(a["atrous"], b["atrous"]))
        c, d = oa.stages, ra.stages
        n2 = int((c["mask"][0] != d["mask"]).sum()) + int((c["tiles"] != d["tiles"]).sum()) + int((c["temporal"] != d["temporal"]).sum()) + int((c["blur1"] != d["blur1"]).sum())
        n3 = sum(int((dp.stages[q] != rdp.stages[q]).sum()) for q in ("radiance", "direction_distance", "irradiance", "depth", "output"))
        e, f_ = orp.stages, rrp.stages
        n4 = sum(int((e[q] != f_[q]).sum()) for q in ("trace", "temporal", "moments", "tiles", "output"))
        res.append((n, n2, n3, n4))
    ok = all(sum(r) == 0 for r in res)
    bad += (not ok)
    print(trial, name, (W, H), kind, "scale", scale, "dolly %.2f" % dolly, "OK" if ok else ("MISMATCH " + str(res)), flush=True)
print("mismatching configs:", bad)
