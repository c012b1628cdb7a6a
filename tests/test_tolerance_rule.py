"""The tolerance-mode image rule itself (tests/test_gpu_tolerance.py compare16) on synthetic images, no GPU: what it accepts and
what it refuses — in particular the counted allowance for pixels beyond the hard cap and its interplay with the all-texel L2 bound
on small images (one allowed outlier pixel must not fail a 70x32 image through the L2 figure; nine must fail it)."""
import numpy as np
import pytest

import test_gpu_tolerance as T


def _image(h=32, w=70, c=3, seed=0):
    rng = np.random.RandomState(seed)
    return (rng.rand(h, w, c) * 0.5 + 0.1).astype(np.float16)


def _bits(a):
    return a.view(np.uint16)


def test_identical_and_one_ulp_images_pass_a_uniform_bias_does_not():
    ref = _image()
    T.compare16(_bits(ref.copy()), _bits(ref), "identical")
    step = np.random.RandomState(1).choice([-1, 0, 1], size=ref.shape)
    T.compare16((_bits(ref).astype(np.int32) + step).astype(np.uint16), _bits(ref), "one ulp either way")
    with pytest.raises(AssertionError, match="relative L2"):    # every texel 2 ulp UP: inside the per-texel bound, 1.5e-3 of the L2 norm
        T.compare16((_bits(ref).astype(np.int32) + 2).astype(np.uint16), _bits(ref), "two ulp bias")


def test_one_counted_outlier_in_a_small_image_passes_and_is_range_bounded():
    ref = _image()
    got = ref.copy()
    got[5, 7, 1] = np.float16(float(ref[5, 7, 1]) + 0.3)        # inside the channel's value range
    T.compare16(_bits(got), _bits(ref), "one outlier", outlier_pixels=T.REFL_OUTLIERS)
    got[5, 7, 1] = np.float16(5.0)                              # far outside it
    with pytest.raises(AssertionError, match="value range"):
        T.compare16(_bits(got), _bits(ref), "one wild outlier", outlier_pixels=T.REFL_OUTLIERS)


def test_more_outliers_than_the_allowance_fail():
    ref = _image()
    got = ref.copy()
    for k in range(9):
        got[3 + k, 9, 0] = np.float16(0.9)
    with pytest.raises(AssertionError, match="beyond the hard cap"):
        T.compare16(_bits(got), _bits(ref), "nine outliers", outlier_pixels=T.REFL_OUTLIERS)


def test_many_small_errors_fail_the_share_or_the_l2_bound():
    ref = _image()
    got = (_bits(ref).astype(np.int32) + 8).astype(np.uint16)   # every texel 8 ulp up: inside the cap, outside the 2-ulp share
    with pytest.raises(AssertionError):
        T.compare16(got, _bits(ref), "eight ulp everywhere")
