"""The A/B paths that lost and were fenced out of the product build (numbers: docs/EXPERIMENTS.md 4.2 / 4.3): the persistent-wave
shadow trace (HR_TRACE_KERNEL=queue) and the wavefront form of the DDGI probe trace (HR_DDGI_WAVEFRONT=1).  They exist in a library built with
HR_CFLAGS=-DHR_DEV_PATHS (hybrid_rendering_amd/variants/libhybrid_rendering_amd.dev.so, built by __graft_entry__.build()); this module runs against
THAT library — test_dev_paths_in_their_own_build spawns it with HR_LIBRARY set — and skips itself on the product library.  What they prove is
worth keeping: masks and DDGI images do not depend on which lane, wave or kernel traces a ray."""
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV_LIB = os.path.join(ROOT, "hybrid_rendering_amd", "variants", "libhybrid_rendering_amd.dev.so")


def _is_dev(hr):
    return b"+dev" in hr.lib().hr_version()


def test_dev_paths_in_their_own_build(hr):
    """the product library has no dev paths; the variant library runs this module's other tests in a process of its own"""
    if _is_dev(hr):
        pytest.skip("already running on the -DHR_DEV_PATHS build")
    assert b"+dev" not in hr.lib().hr_version()
    if not os.path.exists(DEV_LIB):
        pytest.skip("no hybrid_rendering_amd/variants/libhybrid_rendering_amd.dev.so (HR_CFLAGS=-DHR_DEV_PATHS python -m hybrid_rendering_amd.build --variant dev)")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k", "persistent or wavefront"],
                         capture_output=True, text=True, env=dict(os.environ, HR_LIBRARY=DEV_LIB), cwd=ROOT, timeout=900)
    assert out.returncode == 0 and "2 passed" in out.stdout, out.stdout[-1500:] + out.stderr[-500:]


def test_persistent_wave_shadow_trace_gives_the_same_masks(hr, ctx, monkeypatch):
    import torch
    from hybrid_rendering_amd import synth
    if not _is_dev(hr):
        pytest.skip("product build: no dev paths")
    W, H = 1920, 1080
    sd = helpers.scene_data("sponza")
    scene = hr.Scene(ctx, sd)
    cams = [synth.sponza_camera(W / H, frame=f, dolly=0.5) for f in range(2)]
    ubo = synth.make_ubo(cams[1], cams[0], synth.sponza_light())
    gb = scene.gbuffer(ubo, W, H)
    sob, sr = synth.blue_noise_tables()
    fi = hr.frame_inputs(gb, gb, ubo, 1, 1, torch.from_numpy(sob).cuda(), torch.from_numpy(sr).cuda())
    ref = hr.RayTracedShadows(ctx, W, H)
    ref.ray_trace(scene, fi)
    torch.cuda.synchronize()
    base, rays = ref.image(ref.IMG_MASK).clone(), ref.ray_count()
    monkeypatch.setenv("HR_TRACE_KERNEL", "queue")
    q = hr.RayTracedShadows(ctx, W, H)
    monkeypatch.delenv("HR_TRACE_KERNEL")
    q.ray_trace(scene, fi)
    torch.cuda.synchronize()
    assert torch.equal(q.image(q.IMG_MASK), base) and q.ray_count() == rays and rays > 100000


def test_ddgi_wavefront_variant(oracle, hr, ctx, monkeypatch):
    """HR_DDGI_WAVEFRONT=1 (csrc/trace_queue.h): ray generation -> persistent closest-hit queue -> dense hit shading with a secondary ray
    queue -> persistent any-hit queue -> combine.  Same images, bit for bit, as the single-kernel path and the oracle."""
    if not _is_dev(hr):
        pytest.skip("product build: no dev paths")
    import test_gpu_ddgi as T
    monkeypatch.setenv("HR_DDGI_WAVEFRONT", "1")
    T._run(oracle, hr, ctx, "sponza_small", 160, 96, (6, 3, 5), 128, 3)
    T._run(oracle, hr, ctx, "sponza_small", 96, 64, (4, 3, 4), 96, 2, light_kind="point", params=dict(infinite_bounce_intensity=0.8, gi_intensity=2.0))
