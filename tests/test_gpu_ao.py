"""GPU parity: RayTracedAO (HIP, through the C ABI) vs the CPU oracle, stage by stage, bit for bit."""
import numpy as np
import pytest

import helpers
from hybrid_rendering_amd import synth

pytestmark = pytest.mark.gpu


def _run_case(oracle, hr, ctx, name, W, H, scale, n_frames, dolly, spp=1, params=None):
    import torch
    sd = helpers.scene_data(name)
    osc, gsc = oracle.Scene(sd), hr.Scene(ctx, sd)
    frames = helpers.make_frames(oracle, osc, name, W, H, n_frames, dolly, scale_mips=scale)
    sob, sr = synth.blue_noise_tables()
    sob_d, sr_d = torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda()
    zbp = synth.z_buffer_params()
    w, h = W >> scale, H >> scale
    gp = hr.RayTracedAO(ctx, W, H, scale)
    gp.params.spp = spp
    kw = dict(params or {})
    for k, v in kw.items():
        setattr(gp.params, k, v)
    op = oracle.AOPass(w, h, spp=spp, zbp=zbp, **kw)
    ping = False
    for f in range(n_frames):
        lvl = (lambda fr: fr["mips"][scale] if scale else fr["gb"])
        cur, prev = lvl(frames[f]), lvl(frames[f - 1] if f > 0 else frames[f])
        full = frames[f]["gb"] if scale else None
        op.render(osc, frames[f]["ubo"], cur, prev, sob, sr, f, full=full)
        fi = hr.frame_inputs(helpers.to_cuda(cur), helpers.to_cuda(prev), frames[f]["ubo"], f, ping, sob_d, sr_d,
                             cur_full=helpers.to_cuda(full) if full is not None else None, z_buffer_params=zbp)
        gp.render(gsc, fi)
        torch.cuda.synchronize()
        st = op.stages
        mh = (h + 3) // 4
        mask = gp.image(gp.IMG_MASK).cpu().numpy().view(np.uint32)[:spp * mh].reshape(spp, mh, -1)
        assert int((mask != st["mask"]).sum()) == 0, f"frame {f}: AO mask differs"
        assert gp.ray_count() == st["rays"], f"frame {f}: ray count"
        assert np.array_equal(gp.image(gp.IMG_TILES).cpu().numpy(), st["tiles"]), f"frame {f}: tiles"
        assert np.array_equal(helpers.bits16(gp.image(gp.IMG_AO1 if ping else gp.IMG_AO0)), st["temporal"]), f"frame {f}: temporal AO"
        assert np.array_equal(helpers.bits16(gp.image(gp.IMG_LEN1 if ping else gp.IMG_LEN0)), st["length"]), f"frame {f}: history length"
        assert np.array_equal(helpers.bits16(gp.image(gp.IMG_BLUR0)), st["blur0"]), f"frame {f}: blur x"
        assert np.array_equal(helpers.bits16(gp.image(gp.IMG_BLUR1)), st["blur1"]), f"frame {f}: blur y"
        out = helpers.bits16(gp.output(hr.OUTPUT_UPSAMPLE))
        ref = st["output"] if st["output"].ndim == 2 else st["output"][..., 0]
        assert np.array_equal(out, ref), f"frame {f}: final output differs in {(out != ref).sum()} halfs"
        ping = not ping
    gp.close()
    gsc.close()
    return op


def test_ao_cornell_full_res(oracle, hr, ctx):
    op = _run_case(oracle, hr, ctx, "cornell", 192, 192, 0, 3, 1.0)
    assert 0.3 < helpers.unpack_mask(op.stages["mask"][0], 192, 192).mean() < 0.99


def test_ao_sponza_half_res_upsample(oracle, hr, ctx):
    """Reference default: AO runs at half resolution and is bilaterally upsampled (ray_traced_ao.h:23)."""
    _run_case(oracle, hr, ctx, "sponza_small", 320, 176, 1, 3, 2.0)


def test_ao_ragged_half_res(oracle, hr, ctx):
    """960x540-style odd tile counts: 270/8 is not an integer."""
    _run_case(oracle, hr, ctx, "sponza_small", 240, 140, 1, 2, 1.0)


def test_ao_ragged_width_and_height_edge_threads(oracle, hr, ctx):
    """122x70 AO image: edge threads right of and below the image trace rays too (device_math.h trace_lane_kind), and the
    denoiser's neighbourhood statistics read their bits — as the reference shader does (tests/test_ref_shaders.py)"""
    _run_case(oracle, hr, ctx, "sponza_small", 244, 140, 1, 3, 1.0)
    _run_case(oracle, hr, ctx, "sponza_small", 61, 45, 0, 3, 1.0)


def test_ao_4spp_extension(oracle, hr, ctx):
    _run_case(oracle, hr, ctx, "sponza_small", 256, 144, 0, 3, 1.0, spp=4)


def test_ao_params(oracle, hr, ctx):
    _run_case(oracle, hr, ctx, "cornell", 128, 96, 0, 2, 1.0, params=dict(ray_length=25.0, blur_radius=7, alpha=0.1, bias=0.1))
