"""Per-wave timeline of one launch of the shadow trace kernel (HR_DEBUG_TIMELINE): how many waves are resident over
time, how long waves live, how the work spreads over XCDs / CUs.  python tools/timeline.py [W H]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hybrid_rendering_amd import api as hr, synth

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
path = '/tmp/hr_timeline.bin'
sd = synth.sponza_like(1.0); ctx = hr.Context(0); sc = hr.Scene(ctx, sd)
cam = synth.sponza_camera(W / H); ubo = synth.make_ubo(cam, None, synth.sponza_light())
gb = sc.gbuffer(ubo, W, H); sob, sr = synth.blue_noise_tables()
sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
p = hr.RayTracedShadows(ctx, W, H)
fi = hr.frame_inputs(gb, gb, ubo, 0, 0, sob_d, sr_d)
for _ in range(3):
    p.ray_trace(sc, fi)
torch.cuda.synchronize()
# the library reads its developer switches when a pass object is created (never in render): a second pass carries them
os.environ['HR_DEBUG_TIMELINE'] = path
pt = hr.RayTracedShadows(ctx, W, H)
del os.environ['HR_DEBUG_TIMELINE']
# every call rewrites the file: from the second call on the launch runs in last frame's heaviest-first order (csrc/tile_order.h; HR_TILE_ORDER=0: never)
for _ in range(int(os.environ.get('TIMELINE_CALLS', '3'))):
    pt.ray_trace(sc, fi)
torch.cuda.synchronize()
pt.close()
t = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)
t0 = t[:, 0].min()
start = (t[:, 0] - t0).astype(np.float64) * 0.01   # us (100 MHz)
end = (t[:, 1] - t0).astype(np.float64) * 0.01
dur = end - start
hw = (t[:, 2] & 0xffffffff).astype(np.uint32); xcc = (t[:, 2] >> 32).astype(np.uint32) & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
print('waves %d  span %.1f us  last start %.1f us' % (len(t), end.max(), start.max()))
print('wave duration us: mean %.2f  p50 %.2f  p90 %.2f  p99 %.2f  max %.2f   sum %.0f us' % (
    dur.mean(), np.percentile(dur, 50), np.percentile(dur, 90), np.percentile(dur, 99), dur.max(), dur.sum()))
edges = np.linspace(0, end.max(), 14)
print('time us   resident waves (mean)   started')
for a, b in zip(edges[:-1], edges[1:]):
    mid = np.linspace(a, b, 8)
    res = np.mean([((start <= m) & (end > m)).sum() for m in mid])
    print('%6.1f-%6.1f   %8.0f   %6d' % (a, b, res, ((start >= a) & (start < b)).sum()))
print('per XCC waves', np.bincount(xcc, minlength=8), ' busy-sum us', np.round(np.bincount(xcc, weights=dur, minlength=8)))
key = xcc * 1000 + se * 100 + sh * 50 + cu
u, cnt = np.unique(key, return_counts=True)
busy = np.array([dur[key == k].sum() for k in u])
print('distinct CU keys %d; waves per CU min/mean/max %d/%.0f/%d; busy-sum per CU us min/mean/max %.0f/%.0f/%.0f' % (
    len(u), cnt.min(), cnt.mean(), cnt.max(), busy.min(), busy.mean(), busy.max()))
# slots in launch order: when does slot i start?
idx = np.arange(len(t))
for q in (0.1, 0.25, 0.5, 0.75, 0.9, 1.0):
    i = min(len(t) - 1, int(q * len(t)) - 1)
    print('slot %6d (%.0f%%) starts at %.1f us' % (i, q * 100, start[i]))
tiles_x = W // 8 + (1 if W % 8 else 0)
rows = idx // tiles_x
nrow = rows.max() + 1
print('tile-row band: mean / max wave duration (us), share of total busy time')
for b in range(10):
    m = (rows >= b * nrow // 10) & (rows < (b + 1) * nrow // 10)
    print('  rows %3d-%3d  mean %6.2f  max %6.2f  share %.3f' % (b * nrow // 10, (b + 1) * nrow // 10 - 1, dur[m].mean(), dur[m].max(), dur[m].sum() / dur.sum()))
heavy = np.argsort(-dur)[:20]
print('20 longest waves: (tile x, tile y, start us, dur us)', [(int(i % tiles_x), int(i // tiles_x), round(float(start[i]), 1), round(float(dur[i]), 1)) for i in heavy])
if os.environ.get('HR_DEBUG_TIMELINE_STATS'):
    wmax = (t[:, 3] & 0xffffffff).astype(np.int64); tot = (t[:, 3] >> 32).astype(np.int64)
    print('per-wave max lane steps: mean %.1f p50 %d p90 %d p99 %d max %d' % (wmax.mean(), np.percentile(wmax, 50), np.percentile(wmax, 90), np.percentile(wmax, 99), wmax.max()))
    print('20 longest waves: max lane steps / sum lane steps / us per step', [(int(wmax[i]), int(tot[i]), round(float(dur[i] / max(1, wmax[i])), 2)) for i in heavy])
    m = wmax > 0
    print('us per (max-lane) step over waves with rays: mean %.3f; corr(dur, wmax) %.3f' % ((dur[m] / wmax[m]).mean(), np.corrcoef(dur[m], wmax[m])[0, 1]))
# the critical path of the slowest tiles with the machine to themselves
for i in heavy[:4]:
    os.environ['HR_DEBUG_ONLY_TILE'] = '%d,%d' % (i % tiles_x, i // tiles_x)
    os.environ['HR_DEBUG_TIMELINE'] = path
    pt = hr.RayTracedShadows(ctx, W, H)
    del os.environ['HR_DEBUG_TIMELINE'], os.environ['HR_DEBUG_ONLY_TILE']
    pt.ray_trace(sc, fi); torch.cuda.synchronize()
    pt.close()
    t1 = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)
    print('tile (%d,%d): %.1f us in the full launch, %.1f us alone' % (i % tiles_x, i // tiles_x, dur[i], (float(t1[i, 1]) - float(t1[i, 0])) * 0.01))
