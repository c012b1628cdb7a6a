"""Instanced scenes with a per-frame update (scene_descriptor_set.glsl:30-34 Instance, :102-160 fetch_hit_info / fetch_triangle /
transform_vertex; main.cpp:74 build_tlas) — the CPU half: the oracle's instanced scene against its flattened form, and THE PIN of the
instanced hit shading against the reference's own hit shaders run over several instances through oracle/refshim.  The GPU half is
tests/test_gpu_instances.py."""
import ctypes as C

import numpy as np
import pytest

import helpers
from hybrid_rendering_amd import synth, synth_env
from oracle import pyref


def _frames(oracle, osc, w, h, n, dolly=1.0, light="soft"):
    return helpers.make_frames(oracle, osc, "cornell", w, h, n, dolly, light)


def test_flatten_is_pinned_arithmetic(oracle):
    """orc_instances_flatten (C, -ffp-contract=off) == the numpy restatement, bit for bit (InstancedScene asserts it while building); rotations,
    non-uniform scales; an identity instance returns the object-space vertices unchanged"""
    isd = synth.instanced_cornell(9, seed=11, frame=2)
    osc = oracle.InstancedScene(isd)
    flat = isd.flatten()
    assert flat.n_tris == sum(isd.meshes[k].n_tris for _, k, _ in isd.instances) == len(osc.verts)
    assert np.array_equal(flat.verts[:12].view(np.uint32), np.asarray(isd.meshes[0].verts, np.float32).view(np.uint32))   # instance 0: identity
    assert not np.array_equal(flat.verts[12:24], isd.meshes[1].verts)
    first, mbase, mid, n = isd.layout()
    assert list(first[:3]) == [0, 12, 24] and list(flat.tri_mesh_id[:13]) == [1] * 12 + [2]


def test_identity_instance_equals_flat_scene(oracle):
    """ONE instance with the identity matrix: every query and every pass image equals the flat scene's (x * 1 + y * 0 + z * 0 + 0 is exact)"""
    from oracle import pyoracle_ddgi as od, pyoracle_reflections as orf
    sd = synth.cornell32()
    one = synth.InstancedSceneData(meshes=[sd], instances=[(synth.model_matrix(), 0, 1)], materials=sd.materials)
    o0, o1 = oracle.Scene(sd), oracle.InstancedScene(one)
    W, H = 64, 48
    fr = _frames(oracle, o0, W, H, 2)
    sob, sr = synth.blue_noise_tables()
    p0, p1 = oracle.ShadowsPass(W, H), oracle.ShadowsPass(W, H)
    lo, hi = sd.bounds()
    ddgi = synth_env.ddgi_uniforms(lo, hi, probe_counts=(3, 3, 3), rays_per_probe=32, normal_bias=1.0)
    sky = synth_env.sky_cubemap(8)
    env = dict(sky=sky, prefiltered=synth_env.prefiltered_chain(sky, 4), pre_size=8, pre_levels=4, lut=synth_env.brdf_lut(8))
    d0, d1 = od.DDGIPass(ddgi), od.DDGIPass(ddgi)
    rng = np.random.RandomState(3)
    for f in range(2):
        cur, prev = fr[f]["gb"], fr[f - 1 if f else 0]["gb"]
        # mesh ids differ by construction (sd carries one per box, the instance one per instance): compare on the flat scene's G-buffer
        p0.render(o0, fr[f]["ubo"], cur, prev, sob, sr, f)
        p1.render(o1, fr[f]["ubo"], cur, prev, sob, sr, f)
        for k in ("mask", "temporal", "output"):
            assert np.array_equal(p0.stages[k], p1.stages[k]), (f, k)
        orient = synth_env.random_orientation(rng)
        d0.render(o0, fr[f]["ubo"], cur, sky, orient, f)
        d1.render(o1, fr[f]["ubo"], cur, sky, orient, f)
        assert np.array_equal(d0.stages["radiance"], d1.stages["radiance"]) and np.array_equal(d0.stages["direction_distance"], d1.stages["direction_distance"])
        irr, dep = d0.current_read()
        tp = orf.TraceParams(0.5, 0.8, f, 1, 1, 0.5, 0.5, 0.05)
        a, ra = orf.ray_trace(o0, fr[f]["ubo"], ddgi, cur, sob, sr, tp, env, irr, dep)
        b, rb = orf.ray_trace(o1, fr[f]["ubo"], ddgi, cur, sob, sr, tp, env, irr, dep)
        assert np.array_equal(a, b) and ra == rb


def test_moving_instances_change_the_answers_and_match_a_fresh_flat_scene(oracle):
    """update(matrices) = a fresh oracle scene over the newly flattened vertices: queries equal those of oracle.Scene(flatten(matrices)); the shadow
    mask of the next frame differs from the one before the move"""
    isd = synth.instanced_cornell(5)
    osc = oracle.InstancedScene(isd)
    W, H = 80, 60
    sob, sr = synth.blue_noise_tables()
    masks = []
    rng = np.random.RandomState(1)
    rays = np.zeros((3000, 8), np.float32)
    rays[:, :3] = rng.uniform(5, 95, (3000, 3))
    d = rng.normal(size=(3000, 3))
    rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 3], rays[:, 7] = 1e4, 0.01
    for f in range(3):
        mats = synth.InstancedSceneData(isd.meshes, synth.instanced_cornell_instances(5, frame=f), isd.materials).matrices()
        osc.update(mats)
        flat = oracle.Scene(isd.flatten(mats))
        ta, pa = osc.closest_hit(rays)
        tb, pb = flat.closest_hit(rays)
        assert np.array_equal(ta.view(np.uint32), tb.view(np.uint32)) and np.array_equal(pa, pb)
        assert np.array_equal(osc.any_hit(rays), flat.any_hit(rays))
        fr = _frames(oracle, osc, W, H, 1, 0.0)[0]
        assert all(np.array_equal(fr["gb"][k], flat.gbuffer(fr["ubo"], W, H)[k]) for k in fr["gb"])
        p = oracle.ShadowsPass(W, H)
        p.render(osc, fr["ubo"], fr["gb"], fr["gb"], sob, sr, 0)
        masks.append(p.stages["mask"].copy())
    assert not np.array_equal(masks[0], masks[1]) and not np.array_equal(masks[1], masks[2])


@pytest.mark.skipif(not pyref.available(), reason="neither /root/reference nor a prebuilt oracle/_ref")
@pytest.mark.parametrize("approx,textured", [(1, False), (0, False), (1, True)])
def test_instanced_hit_shading_against_the_reference_hit_shaders(oracle, approx, textured):
    """THE PIN for instances: reflections_ray_trace.{rgen,rchit,rmiss} and gi_ray_trace.{rgen,rchit,rmiss} — the reference's own fetch_hit_info /
    fetch_triangle / interpolated_vertex / transform_vertex (scene_descriptor_set.glsl:102-160) over 10 instances of 3 meshes with rotations and
    non-uniform scales, two of them moving between the frames — against the oracle's instanced surface_at, bit for bit; once with textured materials."""
    from oracle import ref_harness as rh, pyoracle_ddgi as od, pyoracle_reflections as orf
    isd = synth.instanced_cornell(9, seed=5, textured=textured)   # textured: fetch_albedo / fetch_normal / ... through per-mesh uvs and tangents (TBN = (T, T, N))
    osc = oracle.InstancedScene(isd)
    W, H = 56, 40
    lo, hi = isd.flatten().bounds()
    ddgi = synth_env.ddgi_uniforms(lo, hi, probe_counts=(3, 2, 3), rays_per_probe=32, normal_bias=1.0)
    sky = synth_env.sky_cubemap(8)
    env = dict(sky=sky, prefiltered=synth_env.prefiltered_chain(sky, 4), pre_size=8, pre_levels=4, lut=synth_env.brdf_lut(8))
    sob, sr = synth.blue_noise_tables()
    dp = od.DDGIPass(ddgi)
    rng = np.random.RandomState(7)
    hit_instances = set()
    for f in range(2):
        mats = synth.InstancedSceneData(isd.meshes, synth.instanced_cornell_instances(9, seed=5, frame=f), isd.materials).matrices()
        osc.update(mats)
        rsc = rh.RefInstancedScene(isd, mats)
        fr = _frames(oracle, osc, W, H, 1, 0.0)[0]
        cur = fr["gb"]
        ch = cur["gb3"][..., 0]
        ch[ch == np.float16(0.8).view(np.uint16)] = np.float16(0.03).view(np.uint16)      # white surfaces become mirrors: every instance gets reflected
        orient = synth_env.random_orientation(rng)
        rd = int(not dp.ping_pong)
        pirr, pdep, inf = dp.irr[rd].copy(), dp.dep[rd].copy(), dp.p["infinite_bounces"] and not dp.first_frame
        dp.render(osc, fr["ubo"], cur, sky, orient, f)
        rad, dd = rh.ddgi_ray_trace(osc, rsc, fr["ubo"], ddgi, orient, f, inf, dp.p["infinite_bounce_intensity"], sky, pirr, pdep)
        assert np.array_equal(dd, dp.stages["direction_distance"]) and np.array_equal(rad, dp.stages["radiance"]), f"frame {f}: gi_ray_trace"
        irr, dep = dp.current_read()
        tp = orf.TraceParams(0.5, 0.8, f, 1, approx, 0.5, 0.5, 0.05)
        a, rays = orf.ray_trace(osc, fr["ubo"], ddgi, cur, sob, sr, tp, env, irr, dep)
        b = rh.reflections_ray_trace(osc, rsc, fr["ubo"], ddgi, cur, sob, sr, tp, env, irr, dep)
        assert np.array_equal(a, b), f"frame {f}: {int((a != b).any(-1).sum())} texels of the reflections trace image differ"
        assert rays > 0
        hit_instances |= set(np.unique(cur["gb3"][..., 2]).tolist())
    assert len(hit_instances) >= 6   # mesh ids (fp16 bits) of the instances the camera sees


def test_instanced_abi_without_a_gpu():
    from hybrid_rendering_amd import api
    assert C.sizeof(api.hr_instance) == 72 and C.sizeof(api.hr_mesh_desc) == 48 and C.sizeof(api.hr_instanced_scene_desc) == 72
    L = api.lib()
    h = C.c_void_p()
    assert L.hr_scene_create_instanced(None, None, C.byref(h)) == 1            # HR_ERR_INVALID_ARG, never an exception
    assert L.hr_scene_update_instances(None, None, None) == 1
    assert L.hr_scene_instance_count(None) == 0
