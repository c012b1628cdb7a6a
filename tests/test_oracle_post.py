"""Oracle checks for the SURVEY §8f rows 3-4: ground-truth accumulator and TAA (known answers and invariants —
the reference ships no vectors for them)."""
import os
import sys

import numpy as np

import helpers
from hybrid_rendering_amd import synth, synth_env

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def _f(a):
    return a.view(np.float16).astype(np.float32)


def test_halton_known_answers():
    from oracle import pyoracle_post as op
    assert [op.halton(2, i) for i in range(1, 9)] == [0.5, 0.25, 0.75, 0.125, 0.625, 0.375, 0.875, 0.0625]   # van der Corput, exact in fp32
    h3 = [op.halton(3, i) for i in range(1, 5)]
    assert np.allclose(h3, [1 / 3, 2 / 3, 1 / 9, 4 / 9], atol=1e-6)
    t = op.TAAPass(64, 32)
    j0, j1 = t.update(0).copy(), t.update(1).copy()
    assert np.allclose(j0[:2], [(2 * 0.5 - 1) / 64, (2 * (1 / 3) - 1) / 32], atol=1e-7) and np.all(j0[2:] == 0)
    assert np.array_equal(j1[2:], j0[:2])                                       # prev jitter = last frame's current
    assert abs(j1[0] - (2 * 0.25 - 1) / 64) < 1e-7
    t.enabled = False
    assert np.all(t.update(5) == 0)


def test_taa_flat_image_is_a_fixed_point():
    """constant colour, no motion: neighbourhood box = the colour, sharpen = identity, tonemap round trip"""
    from oracle import pyoracle_post as op
    w, h = 40, 24
    for val in (0.25, 0.6, 0.9):
        color = np.zeros((h, w, 4), np.float16); color[..., :3] = val; color[..., 3] = 1
        color = color.view(np.uint16)
        gb = dict(gb2=np.zeros((h, w, 4), np.uint16), depth=np.full((h, w), 0.5, np.float32))
        t = op.TAAPass(w, h)
        t.update(3)
        t.render(color, gb, 1)
        out = _f(t.output(1))
        assert np.allclose(out[..., :3], val, atol=2e-3) and np.all(out[..., 3] == 1.0)


def test_taa_history_blend_and_clipping():
    from oracle import pyoracle_post as op
    w, h = 32, 16
    mk = lambda v: np.ascontiguousarray(np.concatenate([np.full((h, w, 3), v, np.float16), np.ones((h, w, 1), np.float16)], -1)).view(np.uint16)
    gb = dict(gb2=np.zeros((h, w, 4), np.uint16), depth=np.full((h, w), 0.5, np.float32))
    t = op.TAAPass(w, h, reset=False, sharpen=False)
    t.enabled = True
    t.jitter[:] = 0
    t.render(mk(0.2), gb, 0)          # history (zeros) is clipped to the neighbourhood box of the current frame
    assert np.allclose(_f(t.output(0))[..., :3], 0.2, atol=2e-3)
    t.render(mk(0.8), gb, 1)          # history 0.2 lies outside the box [0.8, 0.8] -> clipped to 0.8
    assert np.allclose(_f(t.output(1))[..., :3], 0.8, atol=3e-3)
    # a vertical edge: the output stays inside the local min/max of the input
    img = np.zeros((h, w, 4), np.float16); img[:, : w // 2, :3] = 0.1; img[:, w // 2:, :3] = 0.7; img[..., 3] = 1
    t2 = op.TAAPass(w, h, sharpen=False)
    t2.update(0)
    t2.render(img.view(np.uint16), gb, 0)
    o = _f(t2.output(0))[..., 0]
    assert o.min() >= 0.1 - 2e-3 and o.max() <= 0.7 + 2e-3 and abs(o[:, 2].mean() - 0.1) < 2e-3 and abs(o[:, -3].mean() - 0.7) < 3e-3


def test_ground_truth_running_mean_and_restart(oracle):
    from oracle import pyoracle_post as op
    sd = helpers.scene_data("cornell")
    sc = oracle.Scene(sd)
    w = h = 48
    fr = helpers.make_frames(oracle, sc, "cornell", w, h, 1, 0.0, "soft")[0]
    sky = synth_env.sky_cubemap(8)
    gt = op.GroundTruthPass(w, h)
    outs = [gt.render(sc, fr["ubo"], sky).copy() for _ in range(6)]
    assert gt.rays >= w * h and gt.rays <= 3 * w * h
    f = [_f(o)[..., :3] for o in outs]
    assert all(np.isfinite(x).all() and x.min() >= 0 and x.max() <= 1.0 for x in f) and all(np.all(_f(o)[..., 3] == 1) for o in outs)
    # upstream quirk (rgen:104): frame k blends with weight 1/k, so frame 1 REPLACES frame 0
    single = op.GroundTruthPass(w, h)
    single.frame_idx = 1
    single.ping_pong = True
    s1 = _f(single.render(sc, fr["ubo"], sky))[..., :3]      # sample of frame index 1 over a zero history = that sample alone
    assert np.array_equal(f[1], s1)
    # the mean converges: later frames move less
    d = [np.abs(f[i + 1] - f[i]).mean() for i in range(1, 5)]
    assert d[-1] < d[0]
    # restart: the next frame equals a fresh first frame
    gt.restart_accumulation()
    again = gt.render(sc, fr["ubo"], sky)
    assert np.array_equal(again, outs[0])
    # a band renders exactly its rows
    band = op.GroundTruthPass(w, h, band=(16, 32))
    b0 = band.render(sc, fr["ubo"], sky)
    assert np.array_equal(b0[16:32], outs[0][16:32]) and not b0[:16].any() and not b0[32:].any()


def test_ground_truth_sees_the_sky(oracle):
    """a camera looking at the open side of an empty scene gets the (clamped) sky cubemap"""
    from oracle import pyoracle_post as op
    sd = helpers.scene_data("cornell")
    sc = oracle.Scene(sd)
    w, h = 32, 24
    fr = helpers.make_frames(oracle, sc, "cornell", w, h, 1, 0.0)[0]
    sky = synth_env.sky_cubemap(8, intensity=0.3)
    gt = op.GroundTruthPass(w, h)
    out = _f(gt.render(sc, fr["ubo"], sky))
    sky_px = fr["gb"]["depth"] == 1.0
    if sky_px.any():
        vals = _f(sky)[..., :3].reshape(-1, 3)
        px = out[sky_px][:, :3]
        # the primary ray is jittered by up to 1.5 px (rgen:66-68), so pixels next to geometry may hit it
        hits = [bool(np.any(np.all(np.isclose(vals, p, atol=1e-3), axis=1))) for p in px[:60]]
        assert np.mean(hits) > 0.7
