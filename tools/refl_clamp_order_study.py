"""CPU tool (oracle only, no GPU): how much of the reflections' temporal image — and of the a-trous output behind it — depends on nothing but the ORDER of
the fp32 sums of the 17x17 neighbourhood clamp (reflections_denoise_reprojection.comp:133-157: sd = sqrt(E[x^2] - E[x]^2), the history is clipped to mean +- sd)?
The tolerance-mode GPU kernel sums separably (34 instead of 289 LDS reads per pixel); the oracle has a study switch that does the same (ORC_STUDY_SEPARABLE_STATS, never
set by tests / bench / smoke).  This tool renders one configuration of tools/fuzz_tolerance.py twice — reference order / separable order — and compares the stage
images of the chosen frame under the tolerance rule's measure (values beyond 2 fp16 ulp).     python tools/refl_clamp_order_study.py <seed> <trial> <frame>"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 4:      # worker: render and save
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    from hybrid_rendering_amd import synth, synth_env
    from oracle import pyoracle as oracle, pyoracle_ddgi as od, pyoracle_reflections as orf
    seed, want, fwant = (int(v) for v in sys.argv[1:4])
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
    scale = min(scale, 1)
    sd = helpers.scene_data(name)
    osc = oracle.Scene(sd)
    lo, hi = sd.bounds()
    ddgi = synth_env.ddgi_uniforms(lo, hi, probe_counts=(5, 3, 4), rays_per_probe=64, normal_bias=0.1)
    sky = synth_env.sky_cubemap(16)
    pre, lut = synth_env.prefiltered_chain(sky, 5), synth_env.brdf_lut(16)
    env_np = dict(sky=sky, prefiltered=pre, pre_size=16, pre_levels=5, lut=lut)
    frames = helpers.make_frames(oracle, osc, name, W, H, fwant + 1, dolly, scale_mips=scale)
    r01, r003 = np.float16(0.1).view(np.uint16), np.float16(0.03).view(np.uint16)
    for fr in frames:
        for g in [fr["gb"]] + fr.get("mips", [])[1:]:
            ch = g["gb3"][..., 0]
            ch[ch == r01] = r003
    sob, sr = synth.blue_noise_tables()
    w, h = W >> scale, H >> scale
    o_ddgi, op = od.DDGIPass(ddgi), orf.ReflectionsPass(w, h, **dict(rp or {}))
    rr = np.random.RandomState(7)
    for f in range(fwant + 1):
        lvl = (lambda fr: fr["mips"][scale] if scale else fr["gb"])
        cur, prev, full = lvl(frames[f]), lvl(frames[f - 1 if f else 0]), frames[f]["gb"]
        orient = synth_env.random_orientation(rr)
        o_ddgi.render(osc, frames[f]["ubo"], full, sky, orient, f)
        irr, dep = o_ddgi.current_read()
        op.render(osc, frames[f]["ubo"], ddgi, cur, prev, sob, sr, f, env_np, irr, dep, camera_delta=(0.0, 0.0, 0.0) if f == 0 else (-dolly, 0.0, 0.0),
                  full=full if scale else None, ping_pong=bool(f & 1))
    st = op.stages
    np.savez(sys.argv[4], trace=st["trace"], temporal=st["temporal"], moments=st["moments"], atrous=st["atrous"][-1], output=st["output"],
             info=np.array(f"config {want} {name} ({W}, {H}) {light} scale {scale} dolly {dolly:.2f} {'random' if rp else 'default'} params, frame {fwant}"))
    sys.exit(0)

seed, want, fwant = sys.argv[1:4]
with tempfile.TemporaryDirectory() as td:
    paths = {}
    for tag, extra in (("reference", {}), ("separable", {"ORC_STUDY_SEPARABLE_STATS": "1"})):
        paths[tag] = os.path.join(td, tag + ".npz")
        env = dict(os.environ, **extra)
        env.pop("ORC_STUDY_SEPARABLE_STATS", None) if not extra else None
        subprocess.check_call([sys.executable, os.path.abspath(__file__), seed, want, fwant, paths[tag]], env=env, stderr=subprocess.DEVNULL)
    a, b = np.load(paths["reference"]), np.load(paths["separable"])

    def ulps(x, y):
        k = lambda v: np.where(v & 0x8000, 0x8000 - (v & 0x7fff).astype(np.int32), 0x8000 + (v & 0x7fff).astype(np.int32))
        return np.abs(k(x.astype(np.int32)) - k(y.astype(np.int32)))
    print(str(a["info"]))
    for n in ("trace", "temporal", "moments", "atrous", "output"):
        u = ulps(a[n], b[n])
        col = u[..., :3] if u.ndim == 3 and u.shape[2] == 4 and n != "moments" else u
        print(f"  {n:9s}: {int((u > 0).sum()):5d} of {u.size} values differ between the two summation orders, {int((u > 2).sum()):4d} by more than 2 fp16 ulp "
              f"({100.0 * (u > 2).mean():.3f} %; colour channels alone {100.0 * (col > 2).mean():.3f} %), worst {int(u.max())} ulp")
