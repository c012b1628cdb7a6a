"""GPU developer tool: the A/B SURVEY.md §8e asks for — a-trous halos by REDUNDANT COMPUTE (what ships: every band filters 24
extra rows per side, no communication inside a frame) against a NEIGHBOUR EXCHANGE of 1, 2, 4, 8 rows before the four a-trous
iterations — on the band geometry of BASELINE configs[4] (3840 wide, 270-row bands), shadows pass, tolerance mode.

Only one GPU is available to the builder, so the exchange side is measured as a LOWER BOUND: three bands (the middle one has two
neighbours, like an interior band of the 8-GPU frame) run as three ranks of the native transport's loopback back end on one device
(include/hr_comm.h: the same plan, events and stream ordering as the RCCL back end, device-to-device copies instead of
ncclSend / ncclRecv).  The exchange emulation computes with an 8-row halo (the mask rows the temporal 17x17 needs; a real exchange
variant computes even less) and posts + waits four row exchanges between the a-trous iterations.  What is reported per band and frame:
  redundant      halo 24, no exchange inside the frame
  exchange >=    halo 8 + four dependent exchanges through the loopback (RCCL adds its launch + synchronisation latency on top)
usage (GPU box): python tools/halo_ab.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from hybrid_rendering_amd import api as hr, comm, synth
    W, rows, world = 3840, 272, 3   # 272: bands are cut on 8-row tile boundaries (4K / 8 = 270)
    H = rows * world
    sd = synth.sponza_like(1.0)
    ctx = hr.Context(0)
    scene = hr.Scene(ctx, sd)
    light = synth.sponza_light()
    cams = [synth.sponza_camera(W / H, frame=f, dolly=0.5) for f in range(3)]
    ubos = [synth.make_ubo(cams[i + 1], cams[i], light) for i in range(2)]
    gbs = [scene.gbuffer(u, W, H) for u in ubos]
    sob, sr = synth.blue_noise_tables()
    sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
    fis = [hr.frame_inputs(gbs[k & 1], gbs[(k + 1) & 1], ubos[k & 1], k, k & 1, sob_d, sr_d) for k in range(2)]
    bounds = [0, rows, 2 * rows, H]

    def run(halo, exchange, frames=40, warmup=6):
        comms = [comm.NativeComm(ctx, world, r, loopback_name=f"ab{halo}{int(exchange)}") for r in range(world)]
        bands = [hr.RayTracedShadows(ctx, W, H, 0, band=(bounds[r], bounds[r + 1], halo, 40)) for r in range(world)]
        for b in bands:
            b.params.exact = 0
        t0 = None
        for k in range(warmup + frames):
            if k == warmup:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            fi = fis[k & 1]
            fi.num_frames = k
            for r, b in enumerate(bands):
                comms[r].wait()
                b.ray_trace(scene, fi)
                b.temporal(fi)
            for i in range(4):
                if exchange:
                    # rows of the image iteration i reads, from both neighbours, before anyone may start the iteration
                    for r, b in enumerate(bands):
                        src = b.image(b.IMG_TEMPORAL) if i == 0 else b.image(b.IMG_ATROUS0 + ((i - 1) & 1))
                        comms[r].exchange_rows([src], bounds, 1 << i)
                    for r in range(world):
                        comms[r].wait()
                for b in bands:
                    b.atrous_iteration(fi, i)
            for r, b in enumerate(bands):
                comms[r].exchange_shadows(b, bounds, k & 1, 40)   # the history rows (both variants need them)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / frames * 1e3
        for c in comms:
            c.wait()
        torch.cuda.synchronize()
        for b in bands:
            b.close()
        for c in comms:
            c.close()
        return ms / world

    res = {}
    for name, halo, ex in (("redundant (halo 24)", 24, False), ("no halo work at all (halo 8, no exchange: what the exchange could at best save)", 8, False),
                           ("exchange >= (halo 8 + 4 exchanges, loopback)", 8, True), ("redundant (halo 24)", 24, False)):
        ms = run(halo, ex)
        res.setdefault(name, []).append(ms)
        print(f"{name:90s} {ms * 1e3:8.1f} us per band and frame")
    scene.close()


if __name__ == "__main__":
    main()
