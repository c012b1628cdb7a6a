"""Whole hybrid frame (BASELINE.json configs[4]: shadows + AO + DDGI 16x8x16x256 + half-res reflections) on 1..N GPUs.

    python tools/frame_bench.py [--width 3840 --height 2160 --frames 30]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/frame_bench.py --gpus N

Every rank renders its cost-balanced row band of the SAME frame with tiling.TiledShadows / TiledAO / TiledReflections and
its probe slab with tiling.ShardedDDGI (RCCL: one neighbour exchange per tiled pass + one all-gather per DDGI atlas per
frame).  Not the headline benchmark (that is bench.py): a functional + timing driver for the multi-pass, multi-GPU path.
HR_DIST_BACKEND=gloo HR_FORCE_DEVICE=0 runs several ranks on one GPU (functional check)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--ao-spp", type=int, default=4)
    ap.add_argument("--probes", default="16,8,16")
    ap.add_argument("--rays-per-probe", type=int, default=256)
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--exact", type=int, default=0, help="1 = bit-for-bit parity arithmetic, 0 = tolerance mode (the shipping mode)")
    ap.add_argument("--check", action="store_true", help="compare every band with an un-tiled render on this rank (slow)")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("HR_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("HR_DIST_BACKEND", "nccl"))
    from hybrid_rendering_amd import api as hr, api_gi, api_reflections, synth, synth_env, tiling
    W, H = args.width, args.height
    sd = synth.sponza_like(args.detail)
    ctx = hr.Context(local)
    scene = hr.Scene(ctx, sd)
    light = synth.sponza_light()
    sob, sr = synth.blue_noise_tables()
    sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
    cams = [synth.sponza_camera(W / H, frame=f, dolly=0.5) for f in range(3)]
    ubos = [synth.make_ubo(cams[i + 1], cams[i], light) for i in range(2)]
    gbs = [scene.gbuffer(u, W, H) for u in ubos]
    half = [hr.gbuffer_mip(g, 1) for g in gbs]
    zbp = synth.z_buffer_params()
    lo, hi = sd.bounds()
    ddgi_u = synth_env.ddgi_uniforms(lo, hi, probe_counts=tuple(int(v) for v in args.probes.split(",")), rays_per_probe=args.rays_per_probe, normal_bias=0.1)
    sky = synth_env.sky_cubemap(32)
    f16 = lambda a: torch.from_numpy(a).cuda().view(torch.float16)
    env = api_gi.environment(f16(sky), f16(synth_env.prefiltered_chain(sky, 5)), 32, 5, f16(synth_env.brdf_lut(32)))

    # cost-balanced bands from a calibration trace of the shadow pass (full-resolution rows; the half-resolution
    # reflections band is the same band in half-resolution rows, which needs 16-row alignment of the full-res cuts)
    bounds = None
    if world > 1:
        cal = hr.RayTracedShadows(ctx, W, H)
        cal.ray_trace(scene, hr.frame_inputs(gbs[0], gbs[0], ubos[0], 0, 0, sob_d, sr_d))
        cost = tiling.shadow_cost_per_tile_row(gbs[0]["depth"], cal.tile_ray_counts())
        cal.close()
        cost16 = np.add.reduceat(cost, np.arange(0, len(cost), 2))
        bounds = tiling.balanced_bounds(cost16, world, H, min_tiles=4, align=16)
        tb = torch.tensor(bounds, dtype=torch.int64, device="cuda")
        dist.broadcast(tb, src=0)
        bounds = [int(v) for v in tb.cpu()]
    hb = [b // 2 for b in bounds] if bounds else None
    if hb:
        hb[-1] = H // 2
    shadows = tiling.TiledShadows(ctx, W, H, rank, world, bounds=bounds)
    ao = tiling.TiledAO(ctx, W, H, rank, world, scale=0, bounds=bounds)
    ao.params.spp = args.ao_spp
    gi = tiling.ShardedDDGI(ctx, W, H, ddgi_u, rank, world)
    if world > 1 and bounds:
        gi.pass_.set_shard(gi.z0, gi.z1, bounds[rank], bounds[rank + 1])
        gi.b0, gi.b1 = bounds[rank], bounds[rank + 1]
    refl = tiling.TiledReflections(ctx, W, H, rank, world, scale=1, bounds=hb)
    for p_ in (shadows, ao, gi, refl):
        p_.params.exact = 1 if args.check else args.exact       # --check compares band rows bit for bit: parity arithmetic
    rng = np.random.RandomState(1)
    orients = [synth_env.random_orientation(rng) for _ in range(args.warmup + args.frames)]

    def frame(k):
        fi = hr.frame_inputs(gbs[k & 1], gbs[(k + 1) & 1], ubos[k & 1], k, k & 1, sob_d, sr_d, cur_full=gbs[k & 1], z_buffer_params=zbp)
        fh = hr.frame_inputs(half[k & 1], half[(k + 1) & 1], ubos[k & 1], k, k & 1, sob_d, sr_d, cur_full=gbs[k & 1], z_buffer_params=zbp)
        shadows.render(scene, fi)
        ao.render(scene, fi)
        gi.render(scene, fi, env, orients[k])
        refl.render(scene, fh, env, gi.pass_)

    for k in range(args.warmup):
        frame(k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.warmup, args.warmup + args.frames):
        frame(k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ms = (time.perf_counter() - t0) / args.frames * 1e3
    rays = shadows.pass_.ray_count() + ao.pass_.ray_count() + gi.pass_.ray_count() + refl.pass_.ray_count()
    ok = True
    if args.check:
        # every band / shard against an un-tiled instance fed with the same frames (first frames only: cheap)
        ws, wa, wg = hr.RayTracedShadows(ctx, W, H), hr.RayTracedAO(ctx, W, H, 0), api_gi.DDGI(ctx, W, H, ddgi_u)
        wr = api_reflections.RayTracedReflections(ctx, W, H, 1)
        wa.params.spp = args.ao_spp
        ts, ta = tiling.TiledShadows(ctx, W, H, rank, world, bounds=bounds), tiling.TiledAO(ctx, W, H, rank, world, scale=0, bounds=bounds)
        ta.params.spp = args.ao_spp
        tg = tiling.ShardedDDGI(ctx, W, H, ddgi_u, rank, world)
        if world > 1 and bounds:
            tg.pass_.set_shard(tg.z0, tg.z1, bounds[rank], bounds[rank + 1])
        tr = tiling.TiledReflections(ctx, W, H, rank, world, scale=1, bounds=hb)
        for k in range(3):
            fi = hr.frame_inputs(gbs[k & 1], gbs[(k + 1) & 1], ubos[k & 1], k, k & 1, sob_d, sr_d, cur_full=gbs[k & 1], z_buffer_params=zbp)
            fh = hr.frame_inputs(half[k & 1], half[(k + 1) & 1], ubos[k & 1], k, k & 1, sob_d, sr_d, cur_full=gbs[k & 1], z_buffer_params=zbp)
            ws.render(scene, fi); wa.render(scene, fi); wg.render(scene, fi, env, orients[k]); wr.render(scene, fh, env, wg)
            ts.render(scene, fi); ta.render(scene, fi); tg.render(scene, fi, env, orients[k]); tr.render(scene, fh, env, tg.pass_)
            for t in (ts, ta, tr):
                t.wait_exchange()
            torch.cuda.synchronize()
            b0, b1 = ts.b0, ts.b1
            eq = dict(shadows=bool(torch.equal(ts.pass_.output(hr.OUTPUT_ATROUS)[b0:b1], ws.output(hr.OUTPUT_ATROUS)[b0:b1])),
                      ao=bool(torch.equal(ta.pass_.output(hr.OUTPUT_UPSAMPLE)[b0:b1], wa.output(hr.OUTPUT_UPSAMPLE)[b0:b1])),
                      ddgi=bool(torch.equal(tg.pass_.output()[b0:b1], wg.output()[b0:b1])),
                      reflections=bool(torch.equal(tr.pass_.output(hr.OUTPUT_UPSAMPLE)[b0:b1], wr.output(hr.OUTPUT_UPSAMPLE)[b0:b1])))
            if not all(eq.values()):
                print(f"[rank {rank}] frame {k}: band rows {b0}-{b1} differ from the un-tiled render: {eq}", file=sys.stderr)
            ok &= all(eq.values())
        if world > 1:
            t = torch.tensor([1.0 if ok else 0.0], device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = bool(t.item() > 0.5)
    if world > 1:
        t = torch.tensor([ms, float(rays)], dtype=torch.float64, device="cuda")
        tm = t.clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        ms, rays = float(tm[0]), float(t[1])
    if rank == 0:
        print(json.dumps(dict(workload=f"{W}x{H} hybrid frame: shadows + AO {args.ao_spp}spp + DDGI {args.probes}x{args.rays_per_probe} + half-res reflections",
                              n_gpus=world, ms_per_frame=round(ms, 4), frames_per_s=round(1e3 / ms, 1), rays_per_frame=int(rays),
                              Mrays_per_s=round(rays / ms / 1e3, 1), bands=bounds, bit_identical_to_untiled=(ok if args.check else None))))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
