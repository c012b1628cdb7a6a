"""The whole hybrid frame of the reference's render loop (main.cpp:80-83: shadows, AO, DDGI, reflections) on 1..N GPUs —
BASELINE.json configs[4].  Shared by bench.py (the `passes` block and the N > 1 `hybrid_4k` figure) and tools/frame_bench.py.

Every rank renders its cost-balanced row band of the SAME frame with tiling.TiledShadows / TiledAO / TiledReflections and its
probe slab with tiling.ShardedDDGI (RCCL: one neighbour exchange per tiled pass + one all-gather per DDGI atlas per frame)."""
from __future__ import annotations

import time

import numpy as np


class HybridFrame:
    def __init__(self, ctx, scene, sd, W, H, rank=0, world=1, exact=0, ao_spp=4, probes=(16, 8, 16), rays_per_probe=256, refl_scale=1, group=None,
                 concurrent=False, bounds=None):
        import torch
        import torch.distributed as dist
        from . import api as hr, api_gi, synth, synth_env, tiling
        self.ctx, self.scene, self.W, self.H, self.rank, self.world = ctx, scene, W, H, rank, world
        self.hr = hr
        light = synth.sponza_light()
        sob, sr = synth.blue_noise_tables()
        self.sob_d, self.sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
        cams = [synth.sponza_camera(W / H, frame=f, dolly=0.5) for f in range(3)]
        self.ubos = [synth.make_ubo(cams[i + 1], cams[i], light) for i in range(2)]
        self.gbs = [scene.gbuffer(u, W, H) for u in self.ubos]
        self.low = [hr.gbuffer_mip(g, refl_scale) for g in self.gbs] if refl_scale else self.gbs
        self.zbp = synth.z_buffer_params()
        lo, hi = sd.bounds()
        self.ddgi_u = synth_env.ddgi_uniforms(lo, hi, probe_counts=tuple(probes), rays_per_probe=rays_per_probe, normal_bias=0.1)
        sky = synth_env.sky_cubemap(32)
        f16 = lambda a: torch.from_numpy(a).cuda().view(torch.float16)
        self.env = api_gi.environment(f16(sky), f16(synth_env.prefiltered_chain(sky, 5)), 32, 5, f16(synth_env.brdf_lut(32)))
        # cost-balanced bands from a calibration trace of the shadow pass (full-resolution rows; the low-resolution reflections
        # band is the same band in low-resolution rows, which needs 16-row alignment of the full-res cuts)
        if world > 1 and bounds is None:
            cal = hr.RayTracedShadows(ctx, W, H)
            cal.ray_trace(scene, hr.frame_inputs(self.gbs[0], self.gbs[0], self.ubos[0], 0, 0, self.sob_d, self.sr_d))
            cost = tiling.shadow_cost_per_tile_row(self.gbs[0]["depth"], cal.tile_ray_counts())
            cal.close()
            cost16 = np.add.reduceat(cost, np.arange(0, len(cost), 2))
            bounds = tiling.balanced_bounds(cost16, world, H, min_tiles=4, align=16)
            tb = torch.tensor(bounds, dtype=torch.int64, device="cuda")
            dist.broadcast(tb, src=0, group=group)
            bounds = [int(v) for v in tb.cpu()]
        self.group, self.exact, self.ao_spp, self.refl_scale = group, int(exact), ao_spp, refl_scale
        self._build_passes(bounds)
        self.rng = np.random.RandomState(1)
        # host-side cost matters once the GPU frame is ~1.5 ms: the two parities' input blocks and a ring of probe rotations are built once
        self._inputs = [self._make_inputs(0), self._make_inputs(1)]
        self._orients = [synth_env.random_orientation(self.rng) for _ in range(16)]
        self.ao_spp, self.probes, self.rays_per_probe, self.refl_scale = ao_spp, tuple(probes), rays_per_probe, refl_scale
        # concurrent: shadows, AO and DDGI -> reflections are independent chains inside a frame (the reference records them into one
        # command buffer with per-resource barriers only); on separate HIP streams (+ one for the DDGI probe-grid sample, which only the composite reads) the latency-bound denoise kernels of one chain fill
        # the SIMD slots the VALU-bound trace kernels of another leave idle.  Joined on the caller's stream at the end of the frame.
        self.concurrent = False
        self.concurrent_streams(concurrent)
        import os
        self.forked = world > 1 and os.environ.get("HR_FRAME_FORKED", "0") == "1"   # opt-in (never exercised on a real multi-GPU node yet)
        self.forked_error = None

    def _build_passes(self, bounds):
        """(re)creates the tiled passes of this rank for the band boundaries `bounds` (full-resolution rows; None: one GPU)"""
        from . import tiling
        ctx, W, H, rank, world, group, refl_scale = self.ctx, self.W, self.H, self.rank, self.world, self.group, self.refl_scale
        for p in getattr(self, "_passes", []):
            if hasattr(p, "wait_exchange"):
                p.wait_exchange()      # no history rows in flight into images that are about to be freed
            p.pass_.close()
        self.bounds = bounds
        lb = [b >> refl_scale for b in bounds] if bounds else None
        if lb:
            lb[-1] = H >> refl_scale
        self.shadows = tiling.TiledShadows(ctx, W, H, rank, world, bounds=bounds, group=group)
        self.ao = tiling.TiledAO(ctx, W, H, rank, world, scale=0, bounds=bounds, group=group)
        self.ao.params.spp = self.ao_spp
        self.gi = tiling.ShardedDDGI(ctx, W, H, self.ddgi_u, rank, world, group=group)
        if world > 1 and bounds:
            self.gi.pass_.set_shard(self.gi.z0, self.gi.z1, bounds[rank], bounds[rank + 1])
            self.gi.b0, self.gi.b1 = bounds[rank], bounds[rank + 1]
        self.refl = tiling.TiledReflections(ctx, W, H, rank, world, scale=refl_scale, bounds=lb, group=group)
        self._passes = [self.shadows, self.ao, self.gi, self.refl]
        for p in self._passes:
            p.params.exact = self.exact

    def rebalance(self, rounds=1, barrier=None):
        """N > 1: re-cut the bands from MEASURED per-rank frame times (tiling.rebalanced_bounds).  The first cut balances the
        shadow pass's cost model only; the hybrid frame's cost (AO rays, glossy pixels, each rank's fixed launch floors) shifts it."""
        import torch
        import torch.distributed as dist
        from . import tiling
        if self.world == 1:
            return self.bounds
        for _ in range(rounds):
            ms = self.time(6, 3, barrier=barrier)
            t = torch.zeros(self.world, dtype=torch.float64, device="cuda")
            t[self.rank] = ms
            dist.all_reduce(t, group=self.group)
            new = tiling.rebalanced_bounds(self.bounds, [float(v) for v in t.cpu()], self.H)
            if new == list(self.bounds):
                break
            torch.cuda.synchronize()
            self._build_passes(new)
        return self.bounds

    def concurrent_streams(self, on, mode="streams"):
        """on: enqueue the frame as its dependency graph through the native hr_hybrid_frame (include/hr_api.h) instead of serially —
        mode "streams" (fork / join over internal streams) or "graph" (one hipGraph per frame, updated in place).  One GPU only here:
        the tiled (N > 1) frame keeps its exchanges on one compute stream."""
        import torch
        from . import api_frame
        torch.cuda.synchronize()
        self.concurrent = bool(on) and self.world == 1
        self.frame_mode = {"streams": api_frame.FRAME_STREAMS, "graph": api_frame.FRAME_GRAPH, "serial": api_frame.FRAME_SERIAL}[mode]
        if self.concurrent and getattr(self, "_native", None) is None:
            self._native = api_frame.HybridFrame(self.ctx, self.shadows.pass_, self.ao.pass_, self.gi.pass_, self.refl.pass_)

    def passes(self):
        return dict(shadows=self.shadows.pass_, ao=self.ao.pass_, ddgi=self.gi.pass_, reflections=self.refl.pass_)

    def _make_inputs(self, k):
        hr = self.hr
        a, b = k & 1, (k + 1) & 1
        fi = hr.frame_inputs(self.gbs[a], self.gbs[b], self.ubos[a], k, a, self.sob_d, self.sr_d, cur_full=self.gbs[a], z_buffer_params=self.zbp)
        fl = hr.frame_inputs(self.low[a], self.low[b], self.ubos[a], k, a, self.sob_d, self.sr_d, cur_full=self.gbs[a], z_buffer_params=self.zbp)
        return fi, fl

    def inputs(self, k):
        fi, fl = self._inputs[k & 1]
        fi.num_frames = fl.num_frames = k
        return fi, fl

    def _render_forked(self, fi, fl, k):
        """N > 1, HR_FRAME_FORKED=1: the three independent chains of the frame — shadows | AO | DDGI -> reflections — on three torch streams
        (torch.distributed orders its RCCL operations against the CURRENT stream, so each chain's exchange stays in its own ordering
        domain; every rank issues the collectives in the same host order), joined on the caller's stream."""
        import torch
        main = torch.cuda.current_stream()
        if getattr(self, "_side", None) is None:
            self._side = [torch.cuda.Stream(), torch.cuda.Stream()]
            self._ev = [torch.cuda.Event() for _ in range(3)]
        self._ev[0].record(main)
        for s in self._side:
            s.wait_event(self._ev[0])
        self.shadows.render(self.scene, fi, stream=self._side[0])
        self.ao.render(self.scene, fi, stream=self._side[1])
        self.gi.render(self.scene, fi, self.env, self._orients[k & 15])
        self.refl.render(self.scene, fl, self.env, self.gi.pass_)
        for i, s in enumerate(self._side):
            self._ev[1 + i].record(s)
            main.wait_event(self._ev[1 + i])

    def render(self, k, only=None):
        """frame k in the reference's order (main.cpp:80-83); `only`: one pass name (DDGI still runs before reflections once)"""
        fi, fl = self.inputs(k)
        if self.world > 1 and only is None and getattr(self, "forked", False):
            try:
                return self._render_forked(fi, fl, k)
            except Exception as e:   # automatic fall-back to the serial frame (bench.py reports `forked_error`)
                import torch
                self.forked, self.forked_error = False, repr(e)[:200]
                torch.cuda.synchronize()
                # The forked attempt may have enqueued SOME passes of this frame already (shadows on its side stream, say, before AO raised): their
                # ping-pong, first-frame and launch-order state has advanced once.  Rendering the same frame again serially would advance it twice
                # and blend the frame into its own history, so every pass starts over — a disocclusion-like frame, never a corrupt one.
                for t in (self.shadows, self.ao, self.refl):
                    t.pass_.reset_history()
                self.gi.pass_.restart_accumulation()
        if self.concurrent and only is None:
            self.gi.pass_.set_orientation(self._orients[k & 15])
            self._native.render(self.scene, self.env, fi, fi, fi, fl, mode=self.frame_mode)
            return
        if only in (None, "shadows"):
            self.shadows.render(self.scene, fi)
        if only in (None, "ao"):
            self.ao.render(self.scene, fi)
        if only in (None, "ddgi"):
            self.gi.render(self.scene, fi, self.env, self._orients[k & 15])
        if only in (None, "reflections"):
            self.refl.render(self.scene, fl, self.env, self.gi.pass_)

    def ray_counts(self):
        return {n: int(p.ray_count()) for n, p in self.passes().items()}

    def trace_stats(self, k=200):
        """{pass: (rays, BVH node steps, triangle tests)} of the four ray-trace kernels on frame k, from their instrumented builds
        (hr_*_trace_stats): the BVH term of the trace passes' algorithmic bytes"""
        fi, fl = self.inputs(k)
        self.gi.pass_.set_orientation(self._orients[k & 15])
        return dict(shadows=self.shadows.pass_.trace_stats(self.scene, fi), ao=self.ao.pass_.trace_stats(self.scene, fi),
                    ddgi=self.gi.pass_.trace_stats(self.scene, fi, self.env), reflections=self.refl.pass_.trace_stats(self.scene, fl, self.env, self.gi.pass_))

    def trace_bytes(self, k=200):
        """algorithmic bytes of each ray-trace kernel per launch: pixels (or rays) x (inputs + output) + node steps x 80 B + triangle tests x 48 B"""
        st = self.trace_stats(k)
        px = self.W * self.H
        lpx = (self.W >> self.refl_scale) * (self.H >> self.refl_scale)
        n_probe_rays = int(np.prod(self.probes)) * self.rays_per_probe
        fixed = dict(shadows=px * 12.125, ao=px * (12.0 + self.ao_spp / 8.0), reflections=lpx * 28.0, ddgi=n_probe_rays * 16.0)
        return {n: dict(bytes=int(fixed[n] + st[n][1] * 80 + st[n][2] * 48), rays=st[n][0], nodes_per_ray=round(st[n][1] / max(st[n][0], 1), 2),
                        tris_per_ray=round(st[n][2] / max(st[n][0], 1), 2)) for n in st}

    def time(self, frames, warmup=4, only=None, barrier=None, repeats=1):
        """wall-clock ms per frame between device synchronisations (+ `barrier()` across ranks); the best of `repeats` runs of
        `frames` frames (a single 30 ms hiccup of the host — allocator, collector, driver housekeeping — is 1 ms per frame of a 30-frame run)"""
        import torch
        for k in range(warmup):
            self.render(k, only)
        best, k0 = None, warmup
        fences = [torch.cuda.Event() for _ in range(4)]
        for _ in range(max(1, repeats)):
            torch.cuda.synchronize()
            if barrier:
                barrier()
            t0 = time.perf_counter()
            for k in range(k0, k0 + frames):
                if k - k0 >= len(fences):          # at most 4 frames in flight (a deeper run-ahead stalls the HIP queue sporadically)
                    fences[k % len(fences)].synchronize()
                self.render(k, only)
                fences[k % len(fences)].record()
            torch.cuda.synchronize()
            if barrier:
                barrier()
            ms = (time.perf_counter() - t0) / frames * 1e3
            best = ms if best is None else min(best, ms)
            k0 += frames
        return best

    def stage_times(self, frames=10):
        """per-kernel HIP-event averages of every pass: {pass: {stage: (ms, algorithmic bytes)}}"""
        import torch
        ps = self.passes()
        for p in ps.values():
            p.set_profiling(True)
            p.stage_times()
        for k in range(100, 100 + frames):
            self.render(k)
        torch.cuda.synchronize()
        for t in (self.shadows, self.ao, self.refl):
            t.wait_exchange()
        out = {}
        for n, p in ps.items():
            out[n] = {s: (ms, b) for s, ms, b in p.stage_times()}
            p.set_profiling(False)
        return out

    def close(self):
        for t in getattr(self, "_passes", []):
            if hasattr(t, "wait_exchange"):
                t.wait_exchange()
        if getattr(self, "_native", None) is not None:
            self._native.close()
            self._native = None
        for p in self.passes().values():
            p.close()
