"""The oracle reproduces the committed golden fixtures bit for bit (tests/golden/*.npz, made by make_golden.py).
The reference ships no golden vectors of its own; tests/test_ref_shaders.py proves these fixtures equal the outputs of the
reference's shaders run on the CPU (tests/golden/make_ref_golden.py), so oracle == fixtures == reference shaders."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_oracle_matches_committed_golden(oracle):
    import make_golden
    cases = make_golden.build_cases()
    assert set(cases) == {"shadows_cornell64", "ao_cornell64_half", "ddgi_sponza", "reflections_sponza", "ground_truth_sponza", "taa_sponza"}
    for name, arrs in cases.items():
        gold = np.load(os.path.join(HERE, "golden", name + ".npz"))
        assert set(gold.files) == set(arrs)
        for k, v in arrs.items():
            assert gold[k].dtype == v.dtype and gold[k].shape == v.shape, (name, k)
            assert np.array_equal(gold[k].view(np.uint8), np.ascontiguousarray(v).view(np.uint8)), f"{name}/{k} differs from the committed golden"


def test_golden_content_is_meaningful():
    g = np.load(os.path.join(HERE, "golden", "shadows_cornell64.npz"))
    bits = np.unpackbits(g["mask"].view(np.uint8))
    assert 0.1 < bits.mean() < 0.95                       # partly lit, partly shadowed
    vis = g["output"][..., 0].view(np.float16).astype(np.float32)
    assert 0.0 <= vis.min() and vis.max() <= 1.01 and 0.1 < vis.mean() < 0.95
    r = np.load(os.path.join(HERE, "golden", "reflections_sponza.npz"))
    col = r["output"][..., :3].view(np.float16).astype(np.float32)
    assert np.isfinite(col).all() and col.max() <= 0.7003 and col.mean() > 0.01   # min(colour, 0.7) clamp of the rgen
    d = np.load(os.path.join(HERE, "golden", "ddgi_sponza.npz"))
    dd = d["direction_distance"].view(np.float16).astype(np.float32)
    assert np.allclose(np.linalg.norm(dd[..., :3], axis=-1), 1.0, atol=2e-3)       # unit probe-ray directions
