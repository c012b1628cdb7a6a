#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native ray-trace + denoise hot path.

One "step" = one frame of RayTracedShadows::render (1 spp soft-shadow trace + SVGF temporal +
4 x a-trous) on synthetic 1080p G-buffers of the procedural Sponza-like scene (~278k triangles),
inputs resident in HBM (BASELINE.json configs[1]).  Prints ONE JSON line (see the driver contract).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

N > 1 (weak scaling): the SAME view is rendered with N times the pixels — 1920*sqrt(N) x 1080*sqrt(N), rounded to
multiples of 8 — so rays and pixels per frame grow with N while the content statistics stay those of the N = 1 frame.
The frame is row-tiled, one band per GPU; band boundaries are chosen from a calibration frame so that every band
carries the same share of the cost model  pixels + 1.25 * rays  (tiling.balanced_bounds: sky rows fire no ray, the
floor fires one per pixel — equal-height bands would leave most GPUs idle).  Every band re-traces / re-filters 24 halo
rows locally and, once per frame, exchanges the 40 history rows next to each band boundary with its neighbours over
RCCL (hybrid_rendering_amd/tiling.py), overlapped with the next frame's trace; band rows are bit-identical to the
single-GPU result (tests/test_gpu_tiling.py).  `value` counts only the rays of band rows (halo work is overhead).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
NODE_BYTES, TRI_BYTES = 80, 48


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--detail", type=float, default=1.0, help="scene tessellation (1.0 = ~278k triangles)")
    ap.add_argument("--ring", type=int, default=8, help="distinct camera positions cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=10)
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    if os.environ.get("HR_FORCE_DEVICE") is not None:  # developer switch: several ranks on ONE GPU (functional test of the N>1 path)
        local_rank = int(os.environ["HR_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("HR_DIST_BACKEND", "nccl"))  # "nccl" is RCCL on ROCm

    from hybrid_rendering_amd import api as hr
    from hybrid_rendering_amd import synth, tiling

    # N x the pixels of the N = 1 frame, same aspect and view (multiples of 8: tile / band alignment)
    sc = math.sqrt(world)
    W, H = (int(round(args.width * sc / 8)) * 8, int(round(args.height * sc / 8)) * 8) if world > 1 else (args.width, args.height)
    sd = synth.sponza_like(args.detail)
    ctx = hr.Context(local_rank)
    scene = hr.Scene(ctx, sd)
    light = synth.sponza_light()
    sob, sr = synth.blue_noise_tables()
    sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()

    # ring of camera positions (dolly 0.5 units/frame)
    R = max(2, args.ring)
    cams = [synth.sponza_camera(W / H, frame=f, dolly=0.5) for f in range(R + 1)]
    # G-buffers: ring position i rendered with prev = i-1 (forward sweep) and with prev = i+1 (backward sweep)
    gbs = {}
    ubos = {}
    for i in range(R):
        for direction, j in (("f", i - 1 if i > 0 else 0), ("b", i + 1)):
            ubo = synth.make_ubo(cams[i], cams[j], light)
            ubos[(i, direction)] = ubo
            gbs[(i, direction)] = scene.gbuffer(ubo, W, H)
    torch.cuda.synchronize()

    # sequence of (cur, prev) keys: 0f,1f,...,R-1f, R-2b, ..., 0b, 1f, ...
    seq = [(i, "f") for i in range(R)] + [(i, "b") for i in range(R - 2, -1, -1)]
    seq = seq[1:] if len(seq) > 1 else seq

    def inputs_for(k):
        key = seq[k % len(seq)]
        pk = seq[(k - 1) % len(seq)]
        return hr.frame_inputs(gbs[key], gbs[pk], ubos[key], k, k & 1, sob_d, sr_d)

    cycle = [inputs_for(k) for k in range(len(seq) * 2)]  # even length: ping_pong parity preserved when cycling
    bounds = None
    if world > 1:
        # calibration (outside the timed region, identical on every rank): rays per tile row of the whole frame + geometry
        # pixels per tile row -> band boundaries of equal modelled cost (tiling.shadow_cost_per_tile_row)
        cal = hr.RayTracedShadows(ctx, W, H)
        cal.ray_trace(scene, cycle[0])
        cost = tiling.shadow_cost_per_tile_row(gbs[seq[0]]["depth"], cal.tile_ray_counts())
        cal.close()
        bounds = tiling.balanced_bounds(cost, world, H)
        if world > 1:
            tb = torch.tensor(bounds, dtype=torch.int64, device="cuda")
            dist.broadcast(tb, src=0)      # belt and braces: every rank uses rank 0's partition
            bounds = [int(v) for v in tb.cpu()]
    tiled = tiling.TiledShadows(ctx, W, H, rank, world, bounds=bounds)
    shadows = tiled.pass_
    b0, b1 = tiled.b0, tiled.b1

    def step(k):
        fi = cycle[k % len(cycle)]
        fi.num_frames = k
        tiled.render(scene, fi)

    def barrier():
        if world > 1:
            dist.barrier()

    for k in range(args.warmup):
        step(k)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    # Per-kernel HIP events on the launch stream: with HR_BENCH_INLINE_EVENTS=1 they are recorded INSIDE the timed region (a
    # ring of event pairs per stage in the library, read after the region).  Default: right AFTER it, same stream, same
    # frames cycle — 12 event records per 0.26 ms frame cost 18% of the throughput (measured: 0.262 -> 0.309 ms/frame), while
    # the per-kernel averages agree within 1.5% either way (trace 115.7 vs 117.2 us), and rocprofv3 agrees with both.
    profile_inside = os.environ.get("HR_BENCH_INLINE_EVENTS") is not None
    if profile_inside:
        shadows.set_profiling(True)
        shadows.stage_times()          # start a fresh averaging window
    t0 = time.perf_counter()
    for k in range(args.warmup, args.warmup + args.steps):
        step(k)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    acc = {}
    if profile_inside:
        tiled.wait_exchange()
        for name, ms, nbytes in shadows.stage_times():      # averages over the (last <= 512) timed frames
            acc[name] = [ms, nbytes]
        shadows.set_profiling(False)

    # ---- ray counts (and, if the inline events were switched off, per-stage timing) outside the timed region ----------
    rays_total, n_prof = 0, min(args.steps, 60)
    k0 = args.warmup + args.steps
    if not profile_inside:
        shadows.set_profiling(True)
    for k in range(k0, k0 + n_prof):
        step(k)
        rays_total += shadows.ray_count()
        if not profile_inside:
            for name, ms, nbytes in shadows.stage_times():
                a = acc.setdefault(name, [0.0, nbytes])
                a[0] += ms / n_prof
    shadows.set_profiling(False)
    rays_per_frame = rays_total / n_prof
    if world > 1:
        # useful rays = rays of the band rows only: count them with a halo-free pass on the same inputs
        counter = hr.RayTracedShadows(ctx, W, H, hr.SCALE_FULL_RES, band=(b0, b1, 0, 0))
        tot = 0
        for k in range(k0, k0 + 8):
            fi = cycle[k % len(cycle)]
            fi.num_frames = k
            counter.ray_trace(scene, fi)
            tot += counter.ray_count()
        traced_per_frame, rays_per_frame = rays_per_frame, tot / 8
        counter.close()
    stages = {n: dict(ms=v[0], bytes=v[1]) for n, v in acc.items()}
    # instrumented trace (node visits / triangle tests) on a few frames of the cycle
    nn = nt = nr = 0
    for k in range(4):
        r, a, b = shadows.trace_stats(scene, cycle[(k0 + k) % len(cycle)])
        nr, nn, nt = nr + r, nn + a, nt + b
    nodes_per_ray, tris_per_ray = nn / max(nr, 1), nt / max(nr, 1)
    px = W * (b1 - b0)
    trace_bytes = px * 12.125 + rays_per_frame * (nodes_per_ray * NODE_BYTES + tris_per_ray * TRI_BYTES)
    if "ray_trace" in stages:
        stages["ray_trace"]["bytes"] = int(trace_bytes)
    for s in stages.values():
        s["GBps"] = s["bytes"] / (s["ms"] * 1e-3) / 1e9 if s["ms"] > 0 else 0.0
        s["frac"] = s["GBps"] / HBM_PEAK_GBS

    total_rays = rays_per_frame * args.steps * world
    if world > 1:
        t = torch.tensor([rays_per_frame], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total_rays = float(t.item()) * args.steps
    ms_per_step = elapsed / args.steps * 1e3
    value = total_rays / elapsed / 1e6
    dom = max(stages.items(), key=lambda kv: kv[1]["ms"])
    # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes of this build (separate --pmc runs;
    # FETCH_SIZE is doubled per the gfx950 note in MI355X_MICROARCH.md §HBM) — PMC cannot be sampled from inside this process.
    traffic, traffic_src = None, None
    try:
        pdirs = sorted(d for d in os.listdir(os.path.join(ROOT, "profiles")) if os.path.exists(os.path.join(ROOT, "profiles", d, "pmc_summary.json")))
        if pdirs and world == 1 and (W, args.height) == (1920, 1080):
            pm = json.load(open(os.path.join(ROOT, "profiles", pdirs[-1], "pmc_summary.json")))
            kname = {"ray_trace": "k_shadows_trace<false>", "temporal_accumulation": "k_shadows_temporal"}.get(dom[0], "k_shadows_atrous")
            for k, v in pm.items():
                if kname in k:
                    traffic = int((2 * v["FETCH_SIZE_KB_avg_per_launch"] + v["WRITE_SIZE_KB_avg_per_launch"]) * 1024)
                    traffic_src = f"profiles/{pdirs[-1]}/pmc_summary.json (2*FETCH_SIZE + WRITE_SIZE)"
    except Exception:
        pass
    out = {
        "metric": "shadow Mrays/s over the fully denoised frame (1 spp trace + SVGF temporal + 4x a-trous)",
        "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{W}x{H} procedural Sponza-like ({sd.n_tris} tris) ray-traced shadows 1spp + SVGF denoise"
                               + (f", same view as 1920x1080 with {world}x the pixels, row-tiled into {world} cost-balanced bands (rows {bounds})" if world > 1 else ""),
                   "rays_per_frame_per_gpu": int(rays_per_frame), "pixels_per_gpu": px, "bvh_nodes": int(scene.info.n_nodes),
                   "nodes_per_ray": round(nodes_per_ray, 2), "tris_per_ray": round(tris_per_ray, 2)},
        "denoised_frames_per_s": round(args.steps / elapsed, 2),                  # frames of W x H (the whole tiled frame)
        "denoised_1080p_equiv_per_s": round(W * H / (args.width * args.height) * args.steps / elapsed, 2),
        "trace_only_Mrays_per_s": round(rays_per_frame / (stages["ray_trace"]["ms"] * 1e-3) / 1e6, 2) if "ray_trace" in stages else None,
        "roofline": {"kernel": dom[0], "bound": "hbm", "achieved": round(dom[1]["GBps"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(dom[1]["frac"], 4), "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes": int(dom[1]["bytes"])},
        "stages": {n: {"ms": round(s["ms"], 4), "GBps": round(s["GBps"], 1), "frac": round(s["frac"], 4), "bytes": s["bytes"]} for n, s in stages.items()},
    }

    # ---- CPU baseline: the oracle (a port) on the host cores, bounded sample, rank 0 at N=1 only --------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle as po
        osc = po.Scene(sd)
        op = po.ShadowsPass(W, H)
        host = {k: {n: (t.cpu().numpy().view(np.uint16) if t.dtype == torch.float16 else t.cpu().numpy()) for n, t in gbs[k].items()} for k in seq[:3]}
        op.render(osc, ubos[seq[0]], host[seq[0]], host[seq[0]], sob, sr, 0)  # warm (page-in, history)
        nrays, t0c = 0, time.perf_counter()
        nf = max(1, args.cpu_frames)
        for f in range(nf):
            key, pk = seq[(f + 1) % 3], seq[f % 3]
            op.render(osc, ubos[key], host[key], host[pk], sob, sr, f + 1)
            nrays += op.stages["rays"]
        dt = time.perf_counter() - t0c
        out["cpu_baseline"] = {"value": round(nrays / dt / 1e6, 3), "unit": "Mrays/s", "cores": os.cpu_count(), "kind": "port",
                               "sample": f"{nf} full {W}x{H} frames (trace + SVGF denoise) of the same workload through oracle/ (OpenMP, {os.cpu_count()} threads)",
                               "frames_per_s": round(nf / dt, 3)}
        # the reference's OWN shaders (oracle/_ref, one host thread): a small frame of the same view, trace + denoise
        try:
            from oracle import pyref, ref_harness as rh
            if pyref.available():
                rw, rhh = 240, 136
                rcams = [synth.sponza_camera(rw / rhh, frame=f, dolly=0.5) for f in range(3)]
                rubos = [synth.make_ubo(rcams[i + 1], rcams[i], light) for i in range(2)]
                rgb = [{n: (t.cpu().numpy().view(np.uint16) if t.dtype == torch.float16 else t.cpu().numpy()) for n, t in scene.gbuffer(u, rw, rhh).items()}
                       for u in rubos]
                rp, orp = rh.RefShadowsPass(rw, rhh), po.ShadowsPass(rw, rhh)
                rp.render(osc, rubos[0], rgb[0], rgb[0], sob, sr, 0)
                orp.render(osc, rubos[0], rgb[0], rgb[0], sob, sr, 0)
                t0r = time.perf_counter()
                rp.render(osc, rubos[1], rgb[1], rgb[0], sob, sr, 1)
                dtr = time.perf_counter() - t0r
                orp.render(osc, rubos[1], rgb[1], rgb[0], sob, sr, 1)
                out["cpu_baseline"]["reference_shaders"] = {
                    "value": round(orp.stages["rays"] / dtr / 1e6, 4), "unit": "Mrays/s", "cores": 1,
                    "sample": f"one {rw}x{rhh} frame of the same view through the reference's shaders compiled for the CPU (oracle/_ref)",
                    "bit_identical_to_port": bool(np.array_equal(rp.stages["output"], orp.stages["output"]))}
        except Exception as e:  # the baseline is a report, never a reason to lose the bench line
            out["cpu_baseline"]["reference_shaders"] = {"error": repr(e)[:200]}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
