"""The NATIVE RCCL path (include/hr_comm.h: hr_comm_create_rccl, hr_*_exchange_history, hr_ddgi_allgather_atlases, tickets) on real
devices, one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/native_comm_check.py

torch.distributed only carries the RCCL unique id to the ranks (NativeComm.from_torch_distributed) and the final verdict; every
transfer goes through libhr_comm.so.  Each rank renders its row band of shadows / AO / reflections and its probe slab of DDGI for a few
moving frames (exact mode) and compares its band rows with an un-tiled render of the same frames on its own GPU — the loopback test
(tests/test_gpu_comm.py) with ncclSend / ncclRecv as the wire.  Needs >= 2 GPUs (RCCL refuses two ranks on one device); run by
tests/test_gpu_multi.py when they are visible.  Prints one JSON line on rank 0."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl")
    from hybrid_rendering_amd import api as hr, api_gi, api_reflections, comm, synth, synth_env, tiling
    W, H, n_frames = 192, 264 if world <= 3 else 88 * world, 4
    sd = synth.sponza_like(0.25)
    ctx = hr.Context(local)
    scene = hr.Scene(ctx, sd)
    nc = comm.NativeComm.from_torch_distributed(ctx)
    assert (nc.world, nc.rank) == (world, rank)
    light = synth.sponza_light()
    cams = [synth.sponza_camera(W / H, frame=f, dolly=2.0) for f in range(n_frames + 1)]
    ubos = [synth.make_ubo(cams[i + 1], cams[i], light) for i in range(n_frames)]
    gbs = [scene.gbuffer(u, W, H) for u in ubos]
    sob, sr = synth.blue_noise_tables()
    sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
    zbp = synth.z_buffer_params()
    lo, hi = sd.bounds()
    cz = max(6, world)
    ddgi_u = synth_env.ddgi_uniforms(lo, hi, probe_counts=(4, 3, cz), rays_per_probe=32, normal_bias=0.1)
    sky = synth_env.sky_cubemap(16)
    f16 = lambda a: torch.from_numpy(a).cuda().view(torch.float16)
    env = api_gi.environment(f16(sky), f16(synth_env.prefiltered_chain(sky, 5)), 16, 5, f16(synth_env.brdf_lut(16)))
    bounds = [tiling.band_rows(H, world, r)[0] for r in range(world)] + [H]
    b0, b1 = bounds[rank], bounds[rank + 1]
    w_sh, w_ao, w_gi, w_rf = hr.RayTracedShadows(ctx, W, H), hr.RayTracedAO(ctx, W, H, 0), api_gi.DDGI(ctx, W, H, ddgi_u), api_reflections.RayTracedReflections(ctx, W, H, 0)
    t_sh = hr.RayTracedShadows(ctx, W, H, 0, band=(b0, b1, tiling.HALO, tiling.HISTORY_HALO))
    t_ao = hr.RayTracedAO(ctx, W, H, 0, band=(b0, b1, tiling.HALO, tiling.HALO))
    t_rf = api_reflections.RayTracedReflections(ctx, W, H, 0, band=(b0, b1, tiling.HALO, tiling.HALO))
    t_gi = api_gi.DDGI(ctx, W, H, ddgi_u)
    z0, z1 = tiling.probe_slabs(cz, world, rank)
    t_gi.set_shard(z0, z1, b0, b1)
    rng = np.random.RandomState(3)
    tk = dict(sh=0, ao=0, gi=0, rf=0)
    bad = []
    for f in range(n_frames):
        fi = hr.frame_inputs(gbs[f], gbs[f - 1 if f else 0], ubos[f], f, f & 1, sob_d, sr_d, cur_full=gbs[f], z_buffer_params=zbp)
        orient = synth_env.random_orientation(rng)      # the same on every rank (same seed)
        w_sh.render(scene, fi); w_ao.render(scene, fi); w_gi.render(scene, fi, env, orient); w_rf.render(scene, fi, env, w_gi)
        nc.wait(ticket=tk["sh"]); t_sh.render(scene, fi); tk["sh"] = nc.exchange_shadows(t_sh, bounds, f & 1, tiling.HISTORY_HALO)
        nc.wait(ticket=tk["ao"]); t_ao.render(scene, fi); tk["ao"] = nc.exchange_ao(t_ao, bounds, f & 1, tiling.HALO)
        t_gi.set_orientation(orient)
        t_gi.ray_trace(scene, fi, env); t_gi.probe_update()
        tk["gi"] = nc.allgather_ddgi(t_gi)
        nc.wait(ticket=tk["gi"])
        t_gi.sample_probe_grid(fi); t_gi.end_frame()
        nc.wait(ticket=tk["rf"]); t_rf.render(scene, fi, env, t_gi); tk["rf"] = nc.exchange_reflections(t_rf, bounds, f & 1, tiling.HALO)
        torch.cuda.synchronize()
        gi_r, gd_r = t_gi.current_read()
        wi, wd = w_gi.current_read()
        eq = dict(shadows=bool(torch.equal(t_sh.output(hr.OUTPUT_ATROUS)[b0:b1], w_sh.output(hr.OUTPUT_ATROUS)[b0:b1])),
                  ao=bool(torch.equal(t_ao.output(hr.OUTPUT_UPSAMPLE)[b0:b1], w_ao.output(hr.OUTPUT_UPSAMPLE)[b0:b1])),
                  atlases=bool(torch.equal(gi_r, wi) and torch.equal(gd_r, wd)),
                  ddgi_sample=bool(torch.equal(t_gi.output()[b0:b1], w_gi.output()[b0:b1])),
                  reflections=bool(torch.equal(t_rf.output(hr.OUTPUT_UPSAMPLE)[b0:b1], w_rf.output(hr.OUTPUT_UPSAMPLE)[b0:b1])))
        if not all(eq.values()):
            bad.append((f, eq))
            print(f"[rank {rank}] frame {f}: band rows {b0}-{b1} differ from the un-tiled render: {eq}", file=sys.stderr)
    nc.wait()
    torch.cuda.synchronize()
    t = torch.tensor([0.0 if bad else 1.0, 1.0], device="cuda")
    tmin = t.clone()
    dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps(dict(transport="RCCL (libhr_comm.so)", ranks_seen=int(t[1].item()), frames=n_frames, bands=bounds,
                              bit_identical_to_untiled=bool(tmin[0].item() > 0.5), last_ticket=tk["rf"])))
    nc.close()
    dist.destroy_process_group()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
