"""GPU developer tool: what hr_scene_update_instances costs and what the refitted tree costs the trace kernels.
Scene: the bench building (sponza_like(detail), identity instance) + `--movers` instances of a cube / pyramid mesh flying through it.
  * update: wall clock of N back-to-back updates (matrix upload + vertex transform + reference gather + per-level refit), per update;
  * quality: the shadows trace stage on the instanced scene after `--frames` updates of motion against hr_scene_create over the same world
    vertices (a fresh SAH build), same G-buffer — masks must be equal, the time ratio is the price of never rebuilding.
    python tools/instances_probe.py [--detail 1.0 --movers 200 --frames 30 --width 1920 --height 1080]"""
import argparse, json, math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--movers", type=int, default=200)
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    a = ap.parse_args()
    import torch
    from hybrid_rendering_amd import api as hr, synth
    W, H = a.width, a.height
    building = synth.sponza_like(a.detail)
    small = synth.instanced_cornell(2)
    cube, pyr = small.meshes[1], small.meshes[2]
    lo, hi = building.bounds()
    rng = np.random.RandomState(1)
    base = [(rng.uniform(lo + 0.15 * (hi - lo), hi - 0.15 * (hi - lo)), rng.uniform(-1, 1, 3), rng.uniform(0, 6.28), rng.uniform(6, 30, 3), rng.uniform(-2, 2, 3)) for _ in range(a.movers)]

    def instances(f):
        out = [(synth.model_matrix(), 0, 1)]
        for i, (p, ax, ang, sc, vel) in enumerate(base):
            out.append((synth.model_matrix(p + vel * f, ax, ang + 0.05 * f, sc), 1 + (i & 1), 2 + i))
        return out
    isd = synth.InstancedSceneData(meshes=[building, cube, pyr], instances=instances(0), materials=building.materials)
    ctx = hr.Context(0)
    t0 = time.perf_counter()
    g = hr.InstancedScene(ctx, isd)
    t_create = time.perf_counter() - t0
    mats = [synth.InstancedSceneData(isd.meshes, instances(f), isd.materials).matrices() for f in range(a.frames + 1)]
    for m in mats[:3]:
        g.update(m)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for m in mats:
        g.update(m)
    torch.cuda.synchronize()
    upd_ms = (time.perf_counter() - t0) / len(mats) * 1e3
    info = g.refresh_info()
    res = dict(tris=info.n_tris, nodes=info.n_nodes, depth=info.max_depth, instances=a.movers + 1, create_s=round(t_create, 2), update_ms=round(upd_ms, 4),
               top_level_rebuilds=g.top_level_rebuilds, auto_rebuild=os.environ.get("HR_TOP_LEVEL_REBUILD", "1"))
    # quality of the refitted tree after `frames` updates
    light = synth.sponza_light()
    cams = [synth.sponza_camera(W / H, frame=f, dolly=0.5) for f in range(2)]
    ubo = synth.make_ubo(cams[1], cams[0], light)
    sob, sr = synth.blue_noise_tables()
    sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
    flat = hr.Scene(ctx, isd.flatten(mats[-1]))
    gb = flat.gbuffer(ubo, W, H)
    fi = hr.frame_inputs(gb, gb, ubo, 0, 0, sob_d, sr_d)
    out = {}
    for tag, sc in (("refitted", g), ("rebuilt", flat)):
        p = hr.RayTracedShadows(ctx, W, H)
        p.params.exact = 0
        for k in range(6):
            fi.num_frames = k
            p.render(sc, fi)
        p.set_profiling(True)
        acc = {}
        for k in range(6, 26):
            fi.num_frames = k
            p.render(sc, fi)
            for n, t, b in p.stage_times():
                acc[n] = acc.get(n, 0.0) + t / 20
        torch.cuda.synchronize()
        out[tag] = (p.image(p.IMG_MASK).cpu().numpy().copy(), acc)
        p.close()
    assert np.array_equal(out["refitted"][0], out["rebuilt"][0]), "masks differ between the refitted and the rebuilt tree"
    res["shadow_trace_ms_refitted"] = round(out["refitted"][1]["ray_trace"], 4)
    res["shadow_trace_ms_rebuilt"] = round(out["rebuilt"][1]["ray_trace"], 4)
    res["masks_equal"] = True
    print(json.dumps(res))


if __name__ == "__main__":
    main()
