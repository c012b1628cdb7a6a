"""Edge cases of the hot path against the oracle (bit for bit, both arithmetic modes where they differ): frames without a single
ray (all sky; a light that reaches nothing), the smallest images (1x1 ... 9x9: every thread of the only tile is an edge thread),
and degenerate scenes (ONE triangle: the BVH root is a leaf; coincident duplicate triangles)."""
import numpy as np
import pytest

import helpers
from hybrid_rendering_amd import synth

pytestmark = pytest.mark.gpu


def _tables():
    import torch
    sob, sr = synth.blue_noise_tables()
    return sob, sr, torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()


def _run_shadows_and_ao(oracle, hr, ctx, osc, gsc, frames, w, h, what, exact=1, spp=1):
    """shadows (full res) and AO (full res) over `frames` against the oracle; returns the rays fired per frame"""
    import torch
    sob, sr, sob_d, sr_d = _tables()
    zbp = synth.z_buffer_params()
    o_sh, g_sh = oracle.ShadowsPass(w, h), hr.RayTracedShadows(ctx, w, h)
    o_ao, g_ao = oracle.AOPass(w, h, spp=spp, zbp=zbp), hr.RayTracedAO(ctx, w, h, 0)
    g_sh.params.exact = g_ao.params.exact = exact
    g_ao.params.spp = spp
    rays = []
    for f, fr in enumerate(frames):
        cur, prev = fr["gb"], frames[f - 1]["gb"] if f else fr["gb"]
        o_sh.render(osc, fr["ubo"], cur, prev, sob, sr, f)
        o_ao.render(osc, fr["ubo"], cur, prev, sob, sr, f)
        fi = hr.frame_inputs(helpers.to_cuda(cur), helpers.to_cuda(prev), fr["ubo"], f, f & 1, sob_d, sr_d, z_buffer_params=zbp)
        g_sh.render(gsc, fi)
        g_ao.render(gsc, fi)
        torch.cuda.synchronize()
        st, sa = o_sh.stages, o_ao.stages
        assert np.array_equal(g_sh.image(g_sh.IMG_MASK).cpu().numpy().view(np.uint32), st["mask"]), f"{what} frame {f}: shadow mask"
        assert g_sh.ray_count() == st["rays"], f"{what} frame {f}: shadow rays"
        mh = (h + 3) // 4
        assert np.array_equal(g_ao.image(g_ao.IMG_MASK).cpu().numpy().view(np.uint32)[:spp * mh].reshape(spp, mh, -1), sa["mask"].reshape(spp, mh, -1)), f"{what} frame {f}: AO mask"
        assert g_ao.ray_count() == sa["rays"], f"{what} frame {f}: AO rays"
        if exact:
            assert np.array_equal(g_sh.image(g_sh.IMG_TILES).cpu().numpy(), st["tiles"]), f"{what} frame {f}: shadow tile classes"
            assert np.array_equal(helpers.bits16(g_sh.output(hr.OUTPUT_ATROUS)), st["output"]), f"{what} frame {f}: denoised shadows"
            assert np.array_equal(helpers.bits16(g_sh.image(g_sh.IMG_MOMENTS1 if f & 1 else g_sh.IMG_MOMENTS0)), st["moments"]), f"{what} frame {f}: shadow moments"
            assert np.array_equal(g_ao.image(g_ao.IMG_TILES).cpu().numpy(), sa["tiles"]), f"{what} frame {f}: AO tile classes"
            assert np.array_equal(helpers.bits16(g_ao.image(g_ao.IMG_BLUR1)), sa["blur1"]), f"{what} frame {f}: blurred AO"
        else:
            import test_gpu_tolerance as tol
            tol.compare16(helpers.bits16(g_sh.output(hr.OUTPUT_ATROUS)), st["output"], f"{what} frame {f}: denoised shadows (exact = 0)", variance_channels=(1,))
            tol.compare16(helpers.bits16(g_ao.image(g_ao.IMG_BLUR1)), sa["blur1"], f"{what} frame {f}: blurred AO (exact = 0)")
        rays.append((st["rays"], sa["rays"]))
    g_sh.close(); g_ao.close()
    return rays


@pytest.mark.parametrize("exact", [1, 0])
def test_frame_without_a_single_ray(oracle, hr, ctx, exact):
    """every pixel is sky (depth == 1): no ray is fired, masks / tile classes / images are the cleared ones — then geometry comes
    back on the next frame (history of an all-sky frame) and goes again"""
    sd = helpers.scene_data("cornell")
    osc, gsc = oracle.Scene(sd), hr.Scene(ctx, sd)
    w, h = 72, 40
    frames = helpers.make_frames(oracle, osc, "cornell", w, h, 4, 0.3)
    for f in (0, 2):
        gb = {k: v.copy() for k, v in frames[f]["gb"].items()}
        gb["depth"][...] = 1.0
        frames[f] = dict(frames[f], gb=gb)
    rays = _run_shadows_and_ao(oracle, hr, ctx, osc, gsc, frames, w, h, "all-sky", exact)
    assert rays[0] == (0, 0) and rays[2] == (0, 0) and rays[1][0] > 0 and rays[1][1] > 0
    gsc.close()


def test_light_that_reaches_nothing(oracle, hr, ctx):
    """a spot light pointing away from the scene: attenuation 0 everywhere, so the shadow pass fires no ray and every pixel is dark"""
    sd = helpers.scene_data("cornell")
    osc, gsc = oracle.Scene(sd), hr.Scene(ctx, sd)
    w, h = 64, 48
    frames = helpers.make_frames(oracle, osc, "cornell", w, h, 2, 0.2)
    lo, hi = sd.bounds()
    away = synth.make_light(synth.LIGHT_SPOT, direction_to_light=(0.0, -1.0, 0.0), position=(float(lo[0]) - 500.0, float(hi[1]) + 500.0, 0.0), radius=0.5,
                            intensity=10.0, cone_inner_deg=5.0, cone_outer_deg=8.0)
    cams = helpers.cameras("cornell", w / h, 2, 0.2)
    for f in range(2):
        frames[f] = dict(frames[f], ubo=synth.make_ubo(cams[f], cams[f - 1] if f else None, away))
    rays = _run_shadows_and_ao(oracle, hr, ctx, osc, gsc, frames, w, h, "light away")
    assert rays[0][0] == 0 and rays[1][0] == 0 and rays[0][1] > 0
    gsc.close()


@pytest.mark.parametrize("w,h", [(1, 1), (3, 5), (8, 8), (9, 9), (7, 33)])
@pytest.mark.parametrize("exact", [1, 0])
def test_smallest_images(oracle, hr, ctx, w, h, exact):
    sd = helpers.scene_data("cornell")
    osc, gsc = oracle.Scene(sd), hr.Scene(ctx, sd)
    frames = helpers.make_frames(oracle, osc, "cornell", w, h, 3, 0.4)
    _run_shadows_and_ao(oracle, hr, ctx, osc, gsc, frames, w, h, f"{w}x{h}", exact, spp=2)
    gsc.close()


@pytest.mark.parametrize("copies", [1, 3])
def test_one_triangle_scene(oracle, hr, ctx, copies):
    """ONE triangle (the BVH root is a leaf with a single slot), and the same triangle three times over (coincident duplicates:
    equal t, the tie rule decides) — camera and light of the Cornell set-up, so most pixels are sky"""
    sd = helpers.scene_data(f"one_triangle_x{copies}")
    lo, hi = helpers.scene_data("cornell").bounds()
    c, e = (lo + hi) * 0.5, (hi - lo) * 0.35
    osc, gsc = oracle.Scene(sd), hr.Scene(ctx, sd)
    w, h = 96, 64
    frames = helpers.make_frames(oracle, osc, sd.name, w, h, 3, 0.3)
    geom = (frames[0]["gb"]["depth"] != 1.0).mean()
    assert 0.02 < geom < 0.9, geom
    _run_shadows_and_ao(oracle, hr, ctx, osc, gsc, frames, w, h, sd.name, 1, spp=2)
    # closest-hit side: any-hit / closest-hit queries through the single leaf equal the oracle's
    import torch
    rng = np.random.RandomState(3)
    o = (c + rng.uniform(-1, 1, (4096, 3)) * e * 2).astype(np.float32)
    d = rng.normal(size=(4096, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, np.full((4096, 1), 1e4, np.float32), d, np.full((4096, 1), 1e-3, np.float32)], 1).astype(np.float32)   # origin, t_max, dir, t_min
    tuv, prim = osc.closest_hit(rays)
    gt, gp = gsc.closest_hit(torch.from_numpy(rays).cuda())
    assert np.array_equal(gp.cpu().numpy(), prim), "closest-hit primitive ids"
    hit = prim >= 0
    assert hit.sum() > 20
    assert np.array_equal(gt.cpu().numpy()[hit].view(np.uint32), tuv[hit].view(np.uint32)), "closest-hit t, u, v"
    assert np.array_equal(gsc.any_hit(torch.from_numpy(rays).cuda()).cpu().numpy(), osc.any_hit(rays)), "any-hit"
    gsc.close()


@pytest.mark.parametrize("name,w,h", [("one_triangle_x1", 64, 40), ("one_triangle_x3", 40, 24), ("cornell", 1, 1), ("cornell", 3, 5), ("cornell", 8, 8)])
def test_ddgi_and_reflections_on_degenerate_inputs(oracle, hr, ctx, name, w, h):
    """the hit-shading passes (DDGI probe trace / updates / sample, reflections with DDGI feeding them) on the one-triangle scenes
    (nearly every probe ray and reflection ray misses -> sky) and on the smallest images — the parity runners of tests/test_gpu_ddgi.py
    and tests/test_gpu_reflections.py, every stage image bit for bit"""
    import test_gpu_ddgi, test_gpu_reflections
    for label, fn in (("ddgi", lambda: test_gpu_ddgi._run(oracle, hr, ctx, name, w, h, (3, 2, 3), 24, 2)),
                      ("reflections", lambda: test_gpu_reflections._run(oracle, hr, ctx, name, w, h, 0, 2, 0.3, counts=(3, 2, 3)))):
        try:
            fn()
        except AssertionError as e:
            # the runners' parity assertions carry a "frame N: ..." message; their scene-COVERAGE checks (every roughness regime present,
            # some rays hit ...) carry none and cannot hold on a one-triangle scene or a 1x1 image
            if str(e).lstrip().startswith("frame"):
                raise AssertionError(f"{label} on {name} {w}x{h}: {e}")


@pytest.mark.parametrize("dolly", [0.0, 0.05])
def test_forty_frames_of_temporal_feedback(oracle, hr, ctx, dolly):
    """long runs: 40 frames with a static / slowly moving camera — the history length saturates at 32 (`min(32, len + 1)`, reached only
    after 32 successful reprojections), alpha falls to its floor, ping-pong parity and the a-trous feedback run 40 times; DDGI's
    hysteresis blend and its infinite-bounce feedback run 12 frames.  Every stage image of every frame bit for bit."""
    import test_gpu_shadows, test_gpu_ao, test_gpu_ddgi, test_gpu_reflections
    for label, fn in (("shadows", lambda: test_gpu_shadows._run_case(oracle, hr, ctx, "cornell", 56, 40, 40, dolly)),
                      ("ao", lambda: test_gpu_ao._run_case(oracle, hr, ctx, "cornell", 56, 40, 0, 40, dolly, spp=2)),
                      ("ddgi", lambda: test_gpu_ddgi._run(oracle, hr, ctx, "cornell", 40, 24, (3, 2, 3), 32, 12)),
                      ("reflections", lambda: test_gpu_reflections._run(oracle, hr, ctx, "cornell", 56, 40, 0, 36, dolly, counts=(3, 2, 3)))):
        try:
            fn()
        except AssertionError as e:
            if str(e).lstrip().startswith("frame"):   # parity assertions carry "frame N: ..."; scene-coverage checks do not apply here
                raise AssertionError(f"{label}: {e}")


def test_hybrid_frame_modes_survive_parameter_changes(hr, ctx):
    """hr_hybrid_frame with the knobs turned BETWEEN frames: AO spp 4 -> 1 -> 3, shadows filter_iterations 4 -> 2 (fewer launches: the
    captured graph's topology changes, hipGraphExecUpdate must be refused and the graph re-instantiated), shadows denoise off and on
    again, reflections a-trous radius 1 -> 2, the arithmetic mode flipped for one pass, mixed serial / streams / graph frames on one
    object.  Every pass output of every frame equals the plain serial render() calls of a twin set of passes, bit for bit."""
    import torch
    from hybrid_rendering_amd import api_frame
    from hybrid_rendering_amd.frame import HybridFrame
    sd = synth.sponza_like(0.25)
    scene = hr.Scene(ctx, sd)
    ref = HybridFrame(ctx, scene, sd, 328, 184, probes=(5, 3, 4), rays_per_probe=64)
    sched = {2: lambda f: setattr(f.ao.params, "spp", 1),
             3: lambda f: setattr(f.shadows.params, "filter_iterations", 2),
             4: lambda f: setattr(f.shadows.params, "denoise", 0),
             5: lambda f: (setattr(f.shadows.params, "denoise", 1), setattr(f.ao.params, "spp", 3)),
             6: lambda f: setattr(f.refl.params, "radius", 2),
             7: lambda f: setattr(f.ao.params, "exact", 1),
             8: lambda f: (setattr(f.ao.params, "exact", 0), setattr(f.shadows.params, "filter_iterations", 4), setattr(f.refl.params, "radius", 1))}
    modes = ["graph", "graph", "graph", "graph", "streams", "graph", "graph", "serial", "graph", "graph", "streams", "graph"]
    twin = HybridFrame(ctx, scene, sd, 328, 184, probes=(5, 3, 4), rays_per_probe=64)
    for k, mode in enumerate(modes):
        for f in (ref, twin):
            if k in sched:
                sched[k](f)
        ref.render(k)
        twin.concurrent_streams(True, mode)
        twin.render(k)
        torch.cuda.synchronize()
        for n, p in ref.passes().items():
            assert torch.equal(p.output(), twin.passes()[n].output()), f"frame {k} ({mode}): {n} output differs from the serial render() calls"
    inst, upd = twin._native.graph_stats()
    assert inst >= 4 and upd >= 2, (inst, upd)   # topology changes re-instantiate, argument-only changes update in place
    ref.close(); twin.close(); scene.close()


def test_create_destroy_does_not_leak_device_memory(hr, ctx):
    """60 rounds of create -> one frame -> destroy of every pass object (+ the frame orchestrator in graph mode, a scene, a band
    instance): the device's free memory comes back to where it was (hipMalloc'd images, events, streams, graphs)"""
    import torch
    from hybrid_rendering_amd.frame import HybridFrame
    sd = synth.sponza_like(0.25)

    def one_round(k):
        scene = hr.Scene(ctx, sd)
        f = HybridFrame(ctx, scene, sd, 256, 144, probes=(4, 2, 4), rays_per_probe=64)
        f.concurrent_streams(True, ("graph", "streams")[k & 1])
        f.render(0); f.render(1)
        band = hr.RayTracedShadows(ctx, 256, 144, 0, band=(48, 96, 24, 24))
        torch.cuda.synchronize()
        band.close(); f.close(); scene.close()

    def outside_torch():
        # bytes of device memory in use that are NOT torch's caching pool (the library's hipMalloc'd images, events, streams, graphs)
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        return total - free - torch.cuda.memory_reserved()

    import gc
    for k in range(4):          # module loading, graph caches, HIP's own pools settle
        one_round(k)
    gc.collect()
    used0 = outside_torch()
    for k in range(60):
        one_round(k)
    gc.collect()
    used1 = outside_torch()
    assert used1 - used0 < 4 << 20, f"{(used1 - used0) / 2**20:.1f} MiB of device memory did not come back after 60 create / destroy rounds"
