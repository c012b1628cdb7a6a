"""The native transport (include/hr_comm.h, libhr_comm.so) through its LOOPBACK back end: all ranks of the row-tiled frame in one
process on the one GPU of the test box (RCCL refuses two ranks on one device).  The plan, the row ranges, the event ordering and the
per-pass conveniences are the code the RCCL back end runs; only the wire differs (hipMemcpyAsync instead of ncclSend / ncclRecv)."""
import numpy as np
import pytest

import helpers
from hybrid_rendering_amd import synth, synth_env, tiling

pytestmark = pytest.mark.gpu


def test_exchange_rows_and_allgather_loopback(hr, ctx):
    import torch
    from hybrid_rendering_amd import comm
    H, W, world, rows = 96, 40, 3, 16
    bounds = [0, 24, 64, 96]
    comms = [comm.NativeComm(ctx, world, r, loopback_name="t1") for r in range(world)]
    # every rank's copy holds its own band rows = rank + 1, everything else = -1
    imgs = [[torch.full((H, W, 2), -1.0, dtype=torch.float16, device="cuda"), torch.full((H, W), -1, dtype=torch.int32, device="cuda")] for _ in range(world)]
    for r in range(world):
        for t in imgs[r]:
            t[bounds[r]:bounds[r + 1]] = r + 1
    # rank 2 posts first, then 0, then 1: any interleaving must work; nothing is waited for until wait()
    for r in (2, 0, 1):
        comms[r].exchange_rows(imgs[r], bounds, rows)
    for r in range(world):
        comms[r].wait()
    torch.cuda.synchronize()
    for r in range(world):
        for t in imgs[r]:
            v = t[..., 0] if t.dim() == 3 else t
            exp = np.full(H, -1.0)
            exp[bounds[r]:bounds[r + 1]] = r + 1
            if r > 0:
                exp[max(bounds[r] - rows, bounds[r - 1]):bounds[r]] = r          # the upper neighbour's last rows
            if r < world - 1:
                exp[bounds[r + 1]:min(bounds[r + 1] + rows, bounds[r + 2])] = r + 2
            assert np.array_equal(v[:, 0].float().cpu().numpy(), exp), f"rank {r}"
    # ragged all-gather (one empty slab)
    rb = [1, 1, 30, 50]
    atl = [torch.zeros((52, 8), dtype=torch.float32, device="cuda") for _ in range(world)]
    for r in range(world):
        atl[r][rb[r]:rb[r + 1]] = r + 1
    for r in range(world):
        comms[r].allgather_rows(atl[r], rb)
    for r in range(world):
        comms[r].wait()
    torch.cuda.synchronize()
    exp = np.zeros(52); exp[1:30] = 2; exp[30:50] = 3
    for r in range(world):
        assert np.array_equal(atl[r][:, 0].cpu().numpy(), exp)
    for c in comms:
        c.close()


@pytest.mark.parametrize("world,name", [(2, "sponza_small"), (3, "sponza_small"), (2, "cornell"), (3, "cornell")])
def test_tiled_frame_native_comm_equals_untiled(oracle, hr, ctx, world, name):
    """shadows + AO + DDGI + reflections on `world` bands through the C-ABI exchange / all-gather calls: every band row equals the
    un-tiled render bit for bit (exact mode) over 6 frames with camera motion — sideways in the Sponza-like scene, with a VERTICAL component
    in the Cornell box (history rows then really cross the band boundaries: the reflections' colour history must be the exchanged one)."""
    import torch
    from hybrid_rendering_amd import api_gi, api_reflections, comm
    W, H, n_frames = 192, 264, 6
    sd = helpers.scene_data(name)
    osc, gsc = oracle.Scene(sd), hr.Scene(ctx, sd)
    frames = helpers.make_frames(oracle, osc, name, W, H, n_frames, 2.0)
    sob, sr = synth.blue_noise_tables()
    sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
    zbp = synth.z_buffer_params()
    lo, hi = sd.bounds()
    ddgi_u = synth_env.ddgi_uniforms(lo, hi, probe_counts=(4, 3, 6), rays_per_probe=32, normal_bias=0.1)
    sky = synth_env.sky_cubemap(16)
    f16 = lambda a: torch.from_numpy(a).cuda().view(torch.float16)
    env = api_gi.environment(f16(sky), f16(synth_env.prefiltered_chain(sky, 5)), 16, 5, f16(synth_env.brdf_lut(16)))
    bounds = [tiling.band_rows(H, world, r)[0] for r in range(world)] + [H]
    comms = [comm.NativeComm(ctx, world, r, loopback_name=f"frame{world}{name}") for r in range(world)]
    band = lambda r: (bounds[r], bounds[r + 1], tiling.HALO, tiling.HISTORY_HALO)
    w_sh, w_ao, w_gi, w_rf = hr.RayTracedShadows(ctx, W, H), hr.RayTracedAO(ctx, W, H, 0), api_gi.DDGI(ctx, W, H, ddgi_u), api_reflections.RayTracedReflections(ctx, W, H, 0)
    t_sh = [hr.RayTracedShadows(ctx, W, H, 0, band=band(r)) for r in range(world)]
    t_ao = [hr.RayTracedAO(ctx, W, H, 0, band=(bounds[r], bounds[r + 1], tiling.HALO, tiling.HALO)) for r in range(world)]
    t_rf = [api_reflections.RayTracedReflections(ctx, W, H, 0, band=(bounds[r], bounds[r + 1], tiling.HALO, tiling.HALO)) for r in range(world)]
    t_gi = [api_gi.DDGI(ctx, W, H, ddgi_u) for _ in range(world)]
    for r, g in enumerate(t_gi):
        z0, z1 = tiling.probe_slabs(6, world, r)
        g.set_shard(z0, z1, bounds[r], bounds[r + 1])
    rng = np.random.RandomState(3)
    tk = {k: [0] * world for k in ("sh", "ao", "gi", "rf")}
    for f in range(n_frames):
        cur, prev = helpers.to_cuda(frames[f]["gb"]), helpers.to_cuda(frames[f - 1 if f else 0]["gb"])
        fi = hr.frame_inputs(cur, prev, frames[f]["ubo"], f, f & 1, sob_d, sr_d, cur_full=cur, z_buffer_params=zbp)
        orient = synth_env.random_orientation(rng)
        w_sh.render(gsc, fi); w_ao.render(gsc, fi); w_gi.render(gsc, fi, env, orient); w_rf.render(gsc, fi, env, w_gi)
        # one rank after the other on ONE host thread, each driving its passes as hr::Tiled* do (include/hr/tiled.hpp): a pass waits for
        # the TICKET of the exchange it posted last frame only — never for what another pass posted a moment ago, which the neighbour
        # (driven later by this same thread) has not answered yet (round-2 review: a shared pending flag dead-locked exactly this loop)
        for r in range(world):
            c = comms[r]
            c.wait(ticket=tk["sh"][r])
            t_sh[r].render(gsc, fi)
            tk["sh"][r] = c.exchange_shadows(t_sh[r], bounds, f & 1, tiling.HISTORY_HALO)
            c.wait(ticket=tk["ao"][r])
            t_ao[r].render(gsc, fi)
            tk["ao"][r] = c.exchange_ao(t_ao[r], bounds, f & 1, tiling.HALO)
            g = t_gi[r]
            g.set_orientation(orient)
            g.ray_trace(gsc, fi, env); g.probe_update()
            tk["gi"][r] = c.allgather_ddgi(g)
        for r in range(world):                                  # the gather completes when the last rank has posted
            comms[r].wait(ticket=tk["gi"][r])
            g = t_gi[r]
            g.sample_probe_grid(fi); g.end_frame()
            comms[r].wait(ticket=tk["rf"][r])
            t_rf[r].render(gsc, fi, env, g)
            tk["rf"][r] = comms[r].exchange_reflections(t_rf[r], bounds, f & 1, tiling.HALO)
            assert tk["rf"][r] > tk["gi"][r] > tk["ao"][r] > tk["sh"][r] > 0
        torch.cuda.synchronize()
        for r in range(world):
            b0, b1 = bounds[r], bounds[r + 1]
            assert torch.equal(t_sh[r].output(hr.OUTPUT_ATROUS)[b0:b1], w_sh.output(hr.OUTPUT_ATROUS)[b0:b1]), f"frame {f} rank {r}: shadows"
            assert torch.equal(t_ao[r].output(hr.OUTPUT_UPSAMPLE)[b0:b1], w_ao.output(hr.OUTPUT_UPSAMPLE)[b0:b1]), f"frame {f} rank {r}: AO"
            gi_r, gd_r = t_gi[r].current_read()
            wi, wd = w_gi.current_read()
            assert torch.equal(gi_r, wi) and torch.equal(gd_r, wd), f"frame {f} rank {r}: gathered atlases"
            assert torch.equal(t_gi[r].output()[b0:b1], w_gi.output()[b0:b1]), f"frame {f} rank {r}: DDGI sample"
            assert torch.equal(t_rf[r].output(hr.OUTPUT_UPSAMPLE)[b0:b1], w_rf.output(hr.OUTPUT_UPSAMPLE)[b0:b1]), f"frame {f} rank {r}: reflections"
    for c in comms:
        c.close()


def test_exchange_refuses_bands_shorter_than_the_apron(hr, ctx):
    """a band shorter than the exchanged apron would leave rows of the SECOND neighbour stale: HR_ERR_INVALID_ARG, not a silent clamp"""
    import torch
    from hybrid_rendering_amd import comm
    world, H = 3, 96
    comms = [comm.NativeComm(ctx, world, r, loopback_name="short") for r in range(world)]
    img = torch.zeros((H, 8), dtype=torch.float32, device="cuda")
    with pytest.raises(hr.HRError, match="shorter than"):
        comms[0].exchange_rows([img], [0, 40, 48, 96], 16)       # band 1 holds 8 rows
    assert comms[0].exchange_rows([img], [0, 40, 56, 96], 16) > 0
    # a second rank 0 of the same group is refused, and the failed create leaves nothing behind
    with pytest.raises(hr.HRError, match="already joined"):
        comm.NativeComm(ctx, world, 0, loopback_name="short")
    for c in comms:
        c.close()


def test_rccl_backend_loads_and_initialises_on_one_gpu(hr, ctx):
    """what CAN run of the RCCL back end on a one-GPU box: librccl is dlopen'ed, every symbol resolves, ncclGetUniqueId and
    ncclCommInitRank (world 1) succeed through the C ABI, the communicator tears down.  The send / recv path needs two devices:
    tests/test_gpu_multi.py."""
    import torch
    from hybrid_rendering_amd import comm
    uid = comm.NativeComm.unique_id()
    assert len(uid) == comm.HR_COMM_ID_BYTES and any(uid)
    c = comm.NativeComm(ctx, 1, 0, unique_id=uid)
    img = torch.zeros((64, 8), dtype=torch.float32, device="cuda")
    assert c.exchange_rows([img], [0, 64], 16) == 0     # a single band has no neighbour: nothing posted, ticket 0
    c.wait()
    c.close()


def test_4k_eight_bands_native_comm_equals_untiled(hr, ctx):
    """BASELINE configs[4]'s decomposition in full: the 3840x2160 hybrid frame (shadows + AO 4 spp + DDGI 16x8x16 x 256 rays + half-res
    reflections, exact = 0 — the mode and parameters bench.py times) cut into EIGHT row bands / probe slabs, all eight ranks driven through
    the native transport (loopback wire, per-pass tickets) on the one GPU, three moving frames: every band row of every pass equals the
    un-tiled render bit for bit."""
    import torch
    from hybrid_rendering_amd import api_gi, api_reflections, comm
    W, H, world, n_frames = 3840, 2160, 8, 3
    sd = helpers.scene_data("sponza")
    gsc = hr.Scene(ctx, sd)
    light = synth.sponza_light()
    cams = [synth.sponza_camera(W / H, frame=f, dolly=0.5) for f in range(n_frames + 1)]
    ubos = [synth.make_ubo(cams[i + 1], cams[i], light) for i in range(n_frames)]
    gbs = [gsc.gbuffer(u, W, H) for u in ubos]
    lows = [hr.gbuffer_mip(g, 1) for g in gbs]
    sob, sr = synth.blue_noise_tables()
    sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
    zbp = synth.z_buffer_params()
    lo, hi = sd.bounds()
    ddgi_u = synth_env.ddgi_uniforms(lo, hi, probe_counts=(16, 8, 16), rays_per_probe=256, normal_bias=0.1)
    sky = synth_env.sky_cubemap(32)
    f16 = lambda a: torch.from_numpy(a).cuda().view(torch.float16)
    env = api_gi.environment(f16(sky), f16(synth_env.prefiltered_chain(sky, 5)), 32, 5, f16(synth_env.brdf_lut(32)))
    bounds = [((H // 16) * r // world) * 16 for r in range(world)] + [H]      # 16-row cuts: the half-res bands stay on the 8-row tile grid
    hb = [b // 2 for b in bounds]
    comms = [comm.NativeComm(ctx, world, r, loopback_name="frame4k") for r in range(world)]

    def mk(cls, scale, b, r, hh):
        return cls(ctx, W, H, scale, band=(b[r], b[r + 1], tiling.HALO, hh))
    w_sh, w_ao, w_gi, w_rf = hr.RayTracedShadows(ctx, W, H), hr.RayTracedAO(ctx, W, H, 0), api_gi.DDGI(ctx, W, H, ddgi_u), api_reflections.RayTracedReflections(ctx, W, H, 1)
    t_sh = [mk(hr.RayTracedShadows, 0, bounds, r, tiling.HISTORY_HALO) for r in range(world)]
    t_ao = [mk(hr.RayTracedAO, 0, bounds, r, tiling.HALO) for r in range(world)]
    t_rf = [mk(api_reflections.RayTracedReflections, 1, hb, r, tiling.HALO) for r in range(world)]
    t_gi = [api_gi.DDGI(ctx, W, H, ddgi_u) for _ in range(world)]
    for r, g in enumerate(t_gi):
        z0, z1 = tiling.probe_slabs(16, world, r)
        g.set_shard(z0, z1, bounds[r], bounds[r + 1])
    for p in [w_sh, w_ao, w_gi, w_rf] + t_sh + t_ao + t_rf + t_gi:
        p.params.exact = 0
    for p in [w_ao] + t_ao:
        p.params.spp = 4
    rng = np.random.RandomState(4)
    tk = {k: [0] * world for k in ("sh", "ao", "gi", "rf")}
    for f in range(n_frames):
        fi = hr.frame_inputs(gbs[f], gbs[f - 1 if f else 0], ubos[f], f, f & 1, sob_d, sr_d, cur_full=gbs[f], z_buffer_params=zbp)
        fl = hr.frame_inputs(lows[f], lows[f - 1 if f else 0], ubos[f], f, f & 1, sob_d, sr_d, cur_full=gbs[f], z_buffer_params=zbp)
        orient = synth_env.random_orientation(rng)
        w_sh.render(gsc, fi); w_ao.render(gsc, fi); w_gi.render(gsc, fi, env, orient); w_rf.render(gsc, fl, env, w_gi)
        for r in range(world):
            c = comms[r]
            c.wait(ticket=tk["sh"][r]); t_sh[r].render(gsc, fi); tk["sh"][r] = c.exchange_shadows(t_sh[r], bounds, f & 1, tiling.HISTORY_HALO)
            c.wait(ticket=tk["ao"][r]); t_ao[r].render(gsc, fi); tk["ao"][r] = c.exchange_ao(t_ao[r], bounds, f & 1, tiling.HALO)
            g = t_gi[r]
            g.set_orientation(orient)
            g.ray_trace(gsc, fi, env); g.probe_update()
            tk["gi"][r] = c.allgather_ddgi(g)
        for r in range(world):
            c, g = comms[r], t_gi[r]
            c.wait(ticket=tk["gi"][r])
            g.sample_probe_grid(fi); g.end_frame()
            c.wait(ticket=tk["rf"][r]); t_rf[r].render(gsc, fl, env, g); tk["rf"][r] = c.exchange_reflections(t_rf[r], hb, f & 1, tiling.HALO)
        torch.cuda.synchronize()
        for r in range(world):
            b0, b1 = bounds[r], bounds[r + 1]
            assert torch.equal(t_sh[r].output(hr.OUTPUT_ATROUS)[b0:b1], w_sh.output(hr.OUTPUT_ATROUS)[b0:b1]), f"frame {f} rank {r}: shadows"
            assert torch.equal(t_ao[r].output(hr.OUTPUT_UPSAMPLE)[b0:b1], w_ao.output(hr.OUTPUT_UPSAMPLE)[b0:b1]), f"frame {f} rank {r}: AO"
            gi_r, gd_r = t_gi[r].current_read()
            wi, wd = w_gi.current_read()
            assert torch.equal(gi_r, wi) and torch.equal(gd_r, wd), f"frame {f} rank {r}: gathered atlases"
            assert torch.equal(t_gi[r].output()[b0:b1], w_gi.output()[b0:b1]), f"frame {f} rank {r}: DDGI sample"
            assert torch.equal(t_rf[r].output(hr.OUTPUT_ATROUS)[hb[r]:hb[r + 1]], w_rf.output(hr.OUTPUT_ATROUS)[hb[r]:hb[r + 1]]), f"frame {f} rank {r}: reflections (a-trous)"
            assert torch.equal(t_rf[r].output(hr.OUTPUT_UPSAMPLE)[b0:b1], w_rf.output(hr.OUTPUT_UPSAMPLE)[b0:b1]), f"frame {f} rank {r}: reflections (upsampled)"
        assert not any(t.history_apron_exceeded() for t in t_sh + t_ao + t_rf), f"frame {f}: a history tap left the apron"
    for c in comms:
        c.close()
    for p in [w_sh, w_ao, w_gi, w_rf] + t_sh + t_ao + t_rf + t_gi:
        p.close()
    gsc.close()
