"""The tolerance-mode image rule itself (tests/test_gpu_tolerance.py compare16) on synthetic images, no GPU: what it accepts and
what it refuses — in particular that there is no counted allowance by default (round 5), and how the reflections' bounded allowance
interacts with the all-texel L2 bound."""
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


def test_no_outlier_is_allowed_by_default_and_the_reflections_allowance_is_tightly_bounded():
    """round 5: shadows / AO / DDGI images have NO counted allowance; the reflections' denoised images keep max(4, 2e-5 px) pixels, each within
    OUTLIER_ULPS fp16 ulp or OUTLIER_ABS (not the channel's value range), and the all-texel L2 bound covers those pixels again"""
    ref = _image()
    got = ref.copy()
    got[5, 7, 1] = np.float16(float(ref[5, 7, 1]) + 0.02)       # beyond the hard cap (32 ulp / 2^-10), inside the outlier bound (2^-5)
    with pytest.raises(AssertionError, match="beyond the hard cap"):
        T.compare16(_bits(got), _bits(ref), "one outlier, default rule")
    T.compare16(_bits(got), _bits(ref), "one outlier, reflections rule", outlier_pixels=T.REFL_OUTLIERS)
    got[5, 7, 1] = np.float16(float(ref[5, 7, 1]) + 0.3)        # inside the channel's value range — round 4 accepted it — but beyond the outlier bound
    with pytest.raises(AssertionError, match="outlier pixel differs by more than"):
        T.compare16(_bits(got), _bits(ref), "one wild outlier", outlier_pixels=T.REFL_OUTLIERS)
    assert T.OUTLIER_PIXELS == 0.0 and T.DDGI_OUTLIERS == 0.0 and 0 < T.REFL_OUTLIERS <= 2e-5 and T.OUTLIER_ABS <= 2.0 ** -5 and T.OUTLIER_ULPS <= 512


def test_more_outliers_than_the_allowance_fail():
    ref = _image()
    got = ref.copy()
    for k in range(9):
        got[3 + k, 9, 0] = np.float16(float(ref[3 + k, 9, 0]) + 0.02)
    with pytest.raises(AssertionError, match="beyond the hard cap"):
        T.compare16(_bits(got), _bits(ref), "nine outliers", outlier_pixels=T.REFL_OUTLIERS)


def test_allowed_outliers_count_in_the_all_texel_l2_bound_again():
    ref = (_image(8, 12) * 0.02).astype(np.float16)            # a small, dark image: four allowed outliers of 0.03 outweigh everything else
    got = ref.copy()
    for k in range(4):
        got[1 + k, 3, 0] = np.float16(float(ref[1 + k, 3, 0]) + 0.03)
    with pytest.raises(AssertionError, match="over all texels"):
        T.compare16(_bits(got), _bits(ref), "outliers that dominate the image", outlier_pixels=T.REFL_OUTLIERS)


def test_many_small_errors_fail_the_share_or_the_l2_bound():
    ref = _image()
    got = (_bits(ref).astype(np.int32) + 8).astype(np.uint16)   # every texel 8 ulp up: inside the cap, outside the 2-ulp share
    with pytest.raises(AssertionError):
        T.compare16(got, _bits(ref), "eight ulp everywhere")


def test_absolute_floor_binds_only_below_a_thousandth():
    """OUTPUT_FLOOR = 2^-20: a patch five ulp apart at 1e-4 (3e-7) counts as equal, the same patch five ulp apart at 4e-3 (2e-5) does not"""
    def patch(v):
        ref = np.ones((32, 32), np.float16)
        ref[8:16, 8:16] = v
        got = _bits(ref).copy()
        got[8:16, 8:16] = (got[8:16, 8:16].astype(np.int32) + 5).astype(np.uint16)
        return got, _bits(ref)
    T.compare16(*patch(1.0e-4), "five ulp at 1e-4")
    with pytest.raises(AssertionError):
        T.compare16(*patch(4.0e-3), "five ulp at 4e-3")


def test_thresholds_are_constants():
    """every threshold of the tolerance rule is a constant of tests/test_gpu_tolerance.py: no environment variable can loosen it (round 5 had
    HR_TEST_* overrides for strict fuzz campaigns); the one switch left prints a report"""
    import os, re
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_gpu_tolerance.py")).read()
    assert re.findall(r"environ[^\n]*?(HR_TEST_\w+)", src) == ["HR_TEST_TOLERANCE_REPORT"]
    assert not [k for k in os.environ if k.startswith("HR_TEST_") and k != "HR_TEST_TOLERANCE_REPORT"]
    assert (T.CAP_ULPS, T.CAP_ABS, T.INTERMEDIATE_FLOOR, T.VARIANCE_FLOOR, T.OUTPUT_FLOOR) == (32, 2.0 ** -10, 2e-4, 1e-4, 2.0 ** -20)
    assert (T.OUTLIER_PIXELS, T.DDGI_OUTLIERS, T.REFL_OUTLIERS, T.ATROUS_OUTLIERS, T.OUTLIER_ULPS, T.OUTLIER_ABS) == (0.0, 0.0, 2e-5, 2e-5, 512, 2.0 ** -5)


def test_fuzz_sequences_name_the_recorded_draws():
    """the six sequences of test_reflections_fuzz_sequences_that_missed_the_population_bound are the configurations the round-5 campaigns printed
    (profiles/r5_f/fuzz_tolerance_8088x160.txt, profiles/r5_g/fuzz_one_six_v2.txt)"""
    import helpers
    got = {(s, t): helpers.fuzz_config(s, t) for s, t in T.FUZZ_SEQUENCES}
    want = {(8088, 21): ("sponza_small", 283, 147, "point", 2), (8088, 61): ("sponza_small", 160, 124, "default", 1), (8088, 84): ("sponza_small", 339, 136, "default", 2),
            (8088, 159): ("sponza_small", 177, 136, "default", 2), (555, 66): ("sponza_small", 336, 138, "spot", 1), (31337, 206): ("sponza_small", 181, 159, "spot", 1)}
    for k, c in got.items():
        assert (c["name"], c["W"], c["H"], c["light"], c["scale"]) == want[k], (k, c)
