"""Known-answer tests pinning the oracle itself (the reference ships no tests or golden vectors —
SURVEY.md §4/§8c — so these analytic KATs are what the oracle is anchored on)."""
import ctypes as C

import numpy as np

from hybrid_rendering_amd import synth


def test_f16_conversion_matches_ieee(oracle):
    L = oracle.lib()
    rng = np.random.RandomState(0)
    vals = np.concatenate([rng.uniform(-70000, 70000, 20000), rng.uniform(-1e-4, 1e-4, 20000), rng.normal(size=20000),
                           [0.0, -0.0, 65504.0, 65519.99, 65520.0, 5.96e-8, 2.9802322e-8, 2.98e-8, 3.1e-8, np.inf, -np.inf]]).astype(np.float32)
    got = np.array([L.orc_f32_to_f16(float(v)) for v in vals], np.uint16)
    assert np.array_equal(got, vals.astype(np.float16).view(np.uint16))
    allh = np.arange(0, 0x7c00, 7, dtype=np.uint16)
    back = np.array([L.orc_f16_to_f32(int(h)) for h in allh], np.float32)
    assert np.array_equal(back, allh.view(np.float16).astype(np.float32))


def test_unorm8_identity(oracle):
    """bnd_sampler.glsl: int(clamp(texel*256, 0, 255)) of an UNORM8 texel b/255 is b itself."""
    b = np.arange(256, dtype=np.float32)
    v = np.clip(np.float32(b / np.float32(255.0)) * np.float32(256.0), 0, 255).astype(np.int32)
    assert np.array_equal(v, np.arange(256))


def test_blue_noise_integer_path(oracle):
    L = oracle.lib()
    sob, sr = synth.blue_noise_tables()
    rng = np.random.RandomState(1)
    for _ in range(500):
        x, y, idx, dim = int(rng.randint(0, 4000)), int(rng.randint(0, 4000)), int(rng.randint(0, 100000)), int(rng.randint(0, 2))
        got = L.orc_sample_blue_noise(x, y, idx, dim, sob.ctypes.data_as(C.c_void_p), sr.ctypes.data_as(C.c_void_p))
        t = sr[y % 128, x % 128]
        ranked = (idx % 256) ^ int(t[2])
        value = int(sob[ranked, dim]) ^ int(t[dim % 2])
        assert got == np.float32((0.5 + value) / 256.0)
    # stratification: over 256 consecutive sample indices every value 0..255 appears once per dimension
    vals = sorted(int(L.orc_sample_blue_noise(5, 9, i, 0, sob.ctypes.data_as(C.c_void_p), sr.ctypes.data_as(C.c_void_p)) * 256) for i in range(256))
    assert vals == list(range(256))


def test_octahedral_round_trip(oracle):
    L = oracle.lib()
    rng = np.random.RandomState(2)
    n = rng.normal(size=(2000, 3)).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    enc, dec = (C.c_float * 2)(), (C.c_float * 3)()
    for v in n:
        L.orc_oct_encode((C.c_float * 3)(*v), enc)
        assert -1.0001 <= enc[0] <= 1.0001 and -1.0001 <= enc[1] <= 1.0001
        L.orc_oct_decode(C.c_float(enc[0]), C.c_float(enc[1]), dec)
        assert np.allclose(np.array(dec[:]), v, atol=2e-6)
    # decode of the corners / centre (common.glsl:150-156)
    for e, want in (((0, 0), (0, 0, 1)), ((1, 0), (1, 0, 0)), ((0, -1), (0, -1, 0)), ((1, 1), (0, 0, -1))):
        L.orc_oct_decode(C.c_float(e[0]), C.c_float(e[1]), dec)
        assert np.allclose(dec[:], want, atol=1e-6)


def test_world_position_from_depth_inverts_projection(oracle):
    L = oracle.lib()
    cam = synth.sponza_camera(16 / 9)
    ubo = synth.make_ubo(cam, None, synth.sponza_light())
    VP = np.asarray(ubo["view_proj"], np.float64).reshape(4, 4).T
    rng = np.random.RandomState(3)
    out = (C.c_float * 3)()
    for _ in range(200):
        p = np.array(cam.eye) + np.array([-rng.uniform(5, 600), rng.uniform(-50, 50), rng.uniform(-50, 50)])
        clip = VP @ np.append(p, 1.0)
        ndc = clip[:3] / clip[3]
        u, v = ndc[0] * 0.5 + 0.5, ndc[1] * 0.5 + 0.5
        vpi = np.ascontiguousarray(ubo["view_proj_inverse"], np.float32)
        L.orc_world_position_from_depth(C.c_float(u), C.c_float(v), C.c_float(ndc[2]), vpi.ctypes.data_as(C.c_void_p), out)
        assert np.allclose(out[:], p, rtol=2e-3, atol=0.5), (out[:], p)


def test_transcendentals_close_to_libm(oracle):
    L = oracle.lib()
    s, c = C.c_float(), C.c_float()
    for x in np.linspace(-1.0, 7.0, 4001, dtype=np.float32):
        L.orc_sincos(C.c_float(float(x)), C.byref(s), C.byref(c))
        assert abs(s.value - np.sin(np.float64(x))) < 3e-7 and abs(c.value - np.cos(np.float64(x))) < 3e-7
    for x in np.linspace(-30, 10, 2001, dtype=np.float32):
        assert abs(L.orc_exp(float(x)) / np.exp(np.float64(x)) - 1) < 3e-7
    for x in np.exp(np.linspace(-20, 20, 2001)).astype(np.float32):
        assert abs(L.orc_log(float(x)) - np.log(np.float64(x))) < 3e-7 * max(1.0, abs(np.log(np.float64(x))))
    assert L.orc_pow(0.5, 32.0) == np.float32(0.5) ** 32           # integer powers are exact squarings
    assert abs(L.orc_pow(0.37, 1.2) - 0.37 ** 1.2) < 1e-6
    assert L.orc_pow(0.0, 1.2) == 0.0 and L.orc_exp(-100.0) == 0.0


def test_ubo_layout_and_matrices():
    cam = synth.sponza_camera(16 / 9)
    prev = synth.sponza_camera(16 / 9, frame=1, dolly=2.0)
    ubo = synth.make_ubo(cam, prev, synth.sponza_light())
    assert ubo.nbytes == 416
    VP = np.asarray(ubo["view_proj"], np.float64).reshape(4, 4).T
    VPI = np.asarray(ubo["view_proj_inverse"], np.float64).reshape(4, 4).T
    assert np.allclose(VP @ VPI, np.eye(4), atol=1e-3)
    # a point in front of the camera lands inside the unit depth range, y is flipped (Vulkan)
    p = np.append(np.array(cam.target) + (np.array(cam.target) - np.array(cam.eye)) * 50, 1.0)
    clip = VP @ p
    assert 0 < clip[2] / clip[3] < 1
    L = ubo["light"]
    assert abs(np.linalg.norm(L[0:3]) - 1) < 1e-6 and L[1] > 0.5 and L[7] == np.float32(0.08) and L[12] == 0.0  # sun from above


def test_scenes():
    c = synth.cornell32()
    assert c.n_tris == 32
    s = synth.sponza_like(0.25)
    assert 15000 < s.n_tris < 30000
    lo, hi = s.bounds()
    assert np.all(hi - lo > [1000, 400, 600]) and np.all(hi - lo < [1200, 500, 800])
    assert s.tri_mesh_id.max() < 2048  # exact in fp16 (GB3.b)
    assert np.allclose(np.linalg.norm(s.normals.reshape(-1, 3), axis=1), 1, atol=1e-3)
