"""GPU developer tool: random instanced scenes through hr_scene_create_instanced / hr_scene_update_instances (csrc/instances.hip).
Per configuration: 2-5 meshes (Cornell room, cubes, pyramids, a tessellated sphere, optionally the small Sponza-like building), 2-400 instances with
random rotations, non-uniform / negative / zero scales and shears, 3-6 updates in which a random subset moves (small steps, large jumps, an occasional
forced top-level re-build); after every update 20 k any-hit and closest-hit queries (origins inside and around the scene, short and long rays) must equal
hr_scene_create over the flattened world vertices bit for bit, and every few configurations the shadows / AO masks + DDGI radiance + reflections trace
image are compared with the oracle's instanced scene.   python tools/fuzz_instances.py [seed] [n_configs]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from hybrid_rendering_amd import api as hr, api_gi, api_reflections, synth, synth_env
from oracle import pyoracle as oracle, pyoracle_ddgi as od, pyoracle_reflections as orf
import helpers

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = np.random.RandomState(seed)
ctx = hr.Context(0)
base = synth.instanced_cornell(2)
room, cube, pyr = base.meshes


def sphere(k):
    b = synth._Builder()
    th, ph = np.linspace(0, np.pi, k + 1), np.linspace(0, 2 * np.pi, 2 * k + 1)
    P = lambda i, j: (np.sin(th[i]) * np.cos(ph[j]), np.cos(th[i]), np.sin(th[i]) * np.sin(ph[j]))
    tris = []
    for i in range(k):
        for j in range(2 * k):
            a, b_, c, d = P(i, j), P(i + 1, j), P(i + 1, j + 1), P(i, j + 1)
            if i > 0: tris.append([a, b_, d])
            if i < k - 1: tris.append([b_, c, d])
    b.add(np.array(tris, np.float32), None, int(rng.randint(0, 4)))
    return b.finish(base.materials, "sphere")


def random_matrix(big):
    m = synth.model_matrix(rng.uniform(5, 95, 3) if not big else rng.uniform(-300, 400, 3), rng.uniform(-1, 1, 3), rng.uniform(0, 6.28),
                           rng.uniform(2, 25, 3) * rng.choice([1, 1, 1, -1], 3) * (0.0 if rng.rand() < 0.03 else 1.0)).reshape(4, 4).T.copy()
    if rng.rand() < 0.2:     # shear
        sh = np.eye(4, dtype=np.float32); sh[0, 1] = rng.uniform(-0.7, 0.7); sh[2, 0] = rng.uniform(-0.5, 0.5)
        m = (m @ sh).astype(np.float32)
    return np.ascontiguousarray(m.T.reshape(16), np.float32)


bad = 0
sob, sr = synth.blue_noise_tables()
sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
for trial in range(n):
    meshes = [room, cube, pyr, sphere(int(rng.randint(3, 9)))]
    materials = base.materials
    if rng.rand() < 0.25:
        meshes.append(synth.sponza_like(0.1 + 0.1 * rng.rand()))
        materials = meshes[-1].materials       # 11 materials: the other meshes' indices (0..4) stay valid
    I = int(rng.choice([2, 5, 9, 17, 64, 130, 400]))
    inst = [(synth.model_matrix(), 0, 1)]
    for i in range(I - 1):
        k = int(rng.randint(1, len(meshes)))
        m = random_matrix(big=rng.rand() < 0.1)
        if k == 4:
            m = synth.model_matrix(rng.uniform(-50, 50, 3), (0, 1, 0), rng.uniform(0, 6.28), rng.uniform(0.05, 0.15))
        inst.append((m, k, 2 + i))
    isd = synth.InstancedSceneData(meshes=meshes, instances=inst, materials=materials)
    msg = []
    try:
        g = hr.InstancedScene(ctx, isd)
        mats = isd.matrices().copy()
        for step in range(int(rng.randint(3, 7))):
            if step:
                move = rng.rand(I) < rng.choice([0.1, 0.5, 1.0])
                move[0] = rng.rand() < 0.1
                for i in np.nonzero(move)[0]:
                    if rng.rand() < 0.7:
                        mats[i, 12:15] += rng.uniform(-4, 4, 3).astype(np.float32)
                    else:
                        mats[i] = random_matrix(big=rng.rand() < 0.2)
                g.update(mats)
                if rng.rand() < 0.15:
                    g.rebuild_top_level()
            flat_sd = isd.flatten(mats)
            gf = hr.Scene(ctx, flat_sd)
            lo, hi = flat_sd.bounds()
            rays = np.zeros((20000, 8), np.float32)
            rays[:, :3] = rng.uniform(np.maximum(lo, -100) - 5, np.minimum(hi, 200) + 5, (20000, 3))
            d = rng.normal(size=(20000, 3)); rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
            rays[:, 3] = np.where(rng.rand(20000) < 0.3, rng.uniform(1, 30, 20000), 1e4); rays[:, 7] = 0.01
            rd = torch.from_numpy(rays).cuda()
            a, (ta, pa) = g.any_hit(rd).cpu().numpy(), [t.cpu().numpy() for t in g.closest_hit(rd)]
            b, (tb, pb) = gf.any_hit(rd).cpu().numpy(), [t.cpu().numpy() for t in gf.closest_hit(rd)]
            if not (np.array_equal(a, b) and np.array_equal(pa, pb) and np.array_equal(ta.view(np.uint32), tb.view(np.uint32))):
                msg.append(f"step {step}: queries differ (any {int((a != b).sum())}, prim {int((pa != pb).sum())})")
            gi, fi_ = g.refresh_info(), gf.info
            if list(gi.bounds_lo) != list(fi_.bounds_lo) or list(gi.bounds_hi) != list(fi_.bounds_hi):
                msg.append(f"step {step}: bounds differ")
            gf.close()
        if trial % 4 == 0 and I <= 130:
            # the passes against the oracle's instanced scene on the last state
            W, H = 96, 72
            osc = oracle.InstancedScene(isd, mats)
            cams = helpers.cameras("cornell", W / H, 2, 1.0)
            ubo = synth.make_ubo(cams[1], cams[0], helpers.light_for("cornell", "soft"))
            cur = osc.gbuffer(ubo, W, H)
            got = g.gbuffer(ubo, W, H)
            for k in cur:
                t = got[k].cpu().numpy()
                if not np.array_equal(t.view(np.uint16) if t.dtype == np.float16 else t, cur[k]):
                    msg.append(f"G-buffer {k} differs")
            fi = hr.frame_inputs(helpers.to_cuda(cur), helpers.to_cuda(cur), ubo, 0, 0, sob_d, sr_d, z_buffer_params=synth.z_buffer_params())
            gs, os_ = hr.RayTracedShadows(ctx, W, H), oracle.ShadowsPass(W, H)
            gs.render(g, fi); os_.render(osc, ubo, cur, cur, sob, sr, 0)
            torch.cuda.synchronize()
            if not np.array_equal(gs.image(gs.IMG_MASK).cpu().numpy().view(np.uint32), os_.stages["mask"]): msg.append("shadow mask differs from the oracle")
            flo, fhi = isd.flatten(mats).bounds()
            ddgi = synth_env.ddgi_uniforms(np.maximum(flo, -50), np.minimum(fhi, 150), probe_counts=(3, 3, 3), rays_per_probe=32, normal_bias=1.0)
            sky = synth_env.sky_cubemap(8)
            pre, lut = synth_env.prefiltered_chain(sky, 4), synth_env.brdf_lut(8)
            f16 = lambda a_: torch.from_numpy(a_).cuda().view(torch.float16)
            env = api_gi.environment(f16(sky), f16(pre), 8, 4, f16(lut))
            gd, odd = api_gi.DDGI(ctx, W, H, ddgi), od.DDGIPass(ddgi)
            orient = synth_env.random_orientation(rng)
            gd.render(g, fi, env, orient); odd.render(osc, ubo, cur, sky, orient, 0)
            torch.cuda.synchronize()
            if not np.array_equal(helpers.bits16(gd.image(gd.IMG_RADIANCE)), odd.stages["radiance"]): msg.append("DDGI radiance differs from the oracle")
            irr, dep = odd.current_read()
            gr, orr = api_reflections.RayTracedReflections(ctx, W, H, 0), orf.ReflectionsPass(W, H)
            gr.render(g, fi, env, gd)
            orr.render(osc, ubo, ddgi, cur, cur, sob, sr, 0, dict(sky=sky, prefiltered=pre, pre_size=8, pre_levels=4, lut=lut), irr, dep, ping_pong=False)
            torch.cuda.synchronize()
            if not np.array_equal(helpers.bits16(gr.image(gr.IMG_TRACE)), orr.stages["trace"]): msg.append("reflections trace image differs from the oracle")
            for p in (gs, gd, gr): p.close()
        rb = g.top_level_rebuilds
        g.close()
    except Exception as e:
        msg.append("ERROR " + repr(e)[:200]); rb = -1
    bad += 1 if msg else 0
    print(trial, f"{len(meshes)} meshes, {I} instances, {rb} top-level re-builds:", msg if msg else "ok", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
