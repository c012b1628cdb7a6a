"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product package (hybrid_rendering_amd) never does.  Every stage is pinned bit for bit against the reference's own
shaders executed through oracle/refshim (oracle/pyref.py, tests/test_ref_shaders.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libhr_oracle.so")
_REPLAY8_PATH = os.path.join(_HERE, "_build", "libhr_replay8.so")
_lib = None
_replay8 = None

c_f32p = C.POINTER(C.c_float)
c_u8p = C.POINTER(C.c_uint8)
c_u16p = C.POINTER(C.c_uint16)
c_u32p = C.POINTER(C.c_uint32)
c_i32p = C.POINTER(C.c_int32)
c_u64p = C.POINTER(C.c_uint64)


def build(force: bool = False) -> str:
    """Compile oracle/_build/libhr_oracle.so with the committed Makefile."""
    builder = [os.path.join(_HERE, "..", "hybrid_rendering_amd", "csrc", f) for f in ("bvh_build.cpp", "bvh.h")]   # compiled into libhr_replay8.so
    if force or not os.path.exists(_LIB_PATH) or not os.path.exists(_REPLAY8_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > min(os.path.getmtime(_LIB_PATH), os.path.getmtime(_REPLAY8_PATH))
        for f in os.listdir(_HERE) if f.endswith((".cpp", ".h", "Makefile"))
    ) or any(os.path.exists(b) and os.path.getmtime(b) > os.path.getmtime(_REPLAY8_PATH) for b in builder):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def effective_cpus() -> int:
    """Host threads this process can actually run at once: the smaller of the affinity mask and the cgroup CPU quota.  The GPU box
    shows 256 hardware threads but grants a quota of 16 (cpu.max = 1600000 100000): 256 OpenMP threads on it run SLOWER than 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(round(int(q) / int(per)))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(round(q / per))))
        except Exception:
            pass
    return n


def set_threads(n: int = None) -> int:
    """OpenMP team size of the oracle library (default: effective_cpus(), unless OMP_NUM_THREADS is set)."""
    if n is None:
        if os.environ.get("OMP_NUM_THREADS"):
            return int(os.environ["OMP_NUM_THREADS"])
        n = effective_cpus()
    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(C.c_int(int(n)))
    except OSError:
        pass
    return int(n)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        set_threads()
        _lib.orc_scene_create.restype = C.c_void_p
        _lib.orc_scene_num_nodes.restype = C.c_int
        _lib.orc_f32_to_f16.restype = C.c_uint16
        _lib.orc_f32_to_f16.argtypes = [C.c_float]
        _lib.orc_f16_to_f32.restype = C.c_float
        _lib.orc_f16_to_f32.argtypes = [C.c_uint16]
        for n in ("orc_exp", "orc_log"):
            getattr(_lib, n).restype = C.c_float
            getattr(_lib, n).argtypes = [C.c_float]
        _lib.orc_pow.restype = C.c_float
        _lib.orc_pow.argtypes = [C.c_float, C.c_float]
        _lib.orc_sample_blue_noise.restype = C.c_float
    return _lib


def _p(a, t):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return a.ctypes.data_as(t)


def _ubo_ptr(ubo: np.ndarray):
    assert ubo.nbytes == 416
    return C.c_void_p(ubo.ctypes.data)


class Replay8:
    """CPU replay of the PRODUCT's 8-wide BVH (oracle/orc_replay8.cpp): the tree the GPU walks, built by the product's host builder, walked on
    the host cores with the oracle's watertight triangle test.  bench.py's cpu_baseline.trace_replay_same_tree and the tests that pin it."""

    def __init__(self, sd):
        global _replay8
        if _replay8 is None:
            if not os.path.exists(_REPLAY8_PATH):
                build()
            _replay8 = C.CDLL(_REPLAY8_PATH)
            _replay8.orc_replay8_create.restype = C.c_void_p
            set_threads()
        self.verts = np.ascontiguousarray(sd.verts, np.float32)
        self.h = C.c_void_p(_replay8.orc_replay8_create(_p(self.verts, c_f32p), C.c_int(sd.n_tris)))

    def num_nodes(self):
        return int(_replay8.orc_replay8_num_nodes(self.h))

    def num_refs(self):
        return int(_replay8.orc_replay8_num_refs(self.h))

    def any_hit(self, rays: np.ndarray, stats=False):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros(len(rays), np.uint8)
        st = np.zeros(2, np.uint64) if stats else None
        _replay8.orc_replay8_any_hit_batch(self.h, C.c_int(len(rays)), _p(rays, c_f32p), _p(out, c_u8p), _p(st, c_u64p))
        return (out, st) if stats else out

    def __del__(self):
        try:
            if self.h and _replay8 is not None:
                _replay8.orc_replay8_destroy(self.h)
                self.h = None
        except Exception:
            pass


class Scene:
    """Oracle scene handle (BVH2 + per-triangle shading data)."""

    def __init__(self, sd):
        self.sd = sd
        self.verts = np.ascontiguousarray(sd.verts, np.float32)
        self.normals = np.ascontiguousarray(sd.normals, np.float32) if sd.normals is not None else None
        self.tri_material = np.ascontiguousarray(sd.tri_material, np.uint32)
        self.tri_mesh_id = np.ascontiguousarray(sd.tri_mesh_id, np.uint32)
        self.materials = np.ascontiguousarray(sd.materials, np.float32)
        self.h = C.c_void_p(lib().orc_scene_create(
            _p(self.verts, c_f32p), C.c_int(sd.n_tris), _p(self.normals, c_f32p), _p(self.tri_material, c_u32p),
            _p(self.tri_mesh_id, c_u32p), _p(self.materials, c_f32p), C.c_int(len(self.materials))))
        if getattr(sd, "material_textures", None) is not None:
            self.uvs = np.ascontiguousarray(sd.uvs, np.float32) if sd.uvs is not None else None
            self.tangents = np.ascontiguousarray(sd.tangents, np.float32) if sd.tangents is not None else None
            self.mat_tex = np.ascontiguousarray(sd.material_textures, np.int32)
            self.textures = [np.ascontiguousarray(t, np.uint8) for t in sd.textures]
            nt = len(self.textures)
            ptrs = (C.c_void_p * nt)(*[t.ctypes.data for t in self.textures])
            tw = np.array([t.shape[1] for t in self.textures], np.int32)
            th = np.array([t.shape[0] for t in self.textures], np.int32)
            lib().orc_scene_set_textures(self.h, _p(self.uvs, c_f32p), _p(self.tangents, c_f32p), _p(self.mat_tex, c_i32p), C.c_int(len(self.mat_tex)),
                                         C.c_int(nt), ptrs, _p(tw, c_i32p), _p(th, c_i32p))

    def __del__(self):
        try:
            if self.h:
                lib().orc_scene_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def num_nodes(self):
        return lib().orc_scene_num_nodes(self.h)

    def any_hit(self, rays: np.ndarray, brute_force=False, stats=False):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros(len(rays), np.uint8)
        st = np.zeros(2, np.uint64) if stats else None
        lib().orc_any_hit_batch(self.h, C.c_int(len(rays)), _p(rays, c_f32p), _p(out, c_u8p), C.c_int(int(brute_force)), _p(st, c_u64p))
        return (out, st) if stats else out

    def closest_hit(self, rays: np.ndarray, brute_force=False):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        tuv = np.zeros((len(rays), 3), np.float32)
        prim = np.zeros(len(rays), np.int32)
        lib().orc_closest_hit_batch(self.h, C.c_int(len(rays)), _p(rays, c_f32p), _p(tuv, c_f32p), _p(prim, c_i32p), C.c_int(int(brute_force)))
        return tuv, prim

    def gbuffer(self, ubo, w, h):
        gb1 = np.zeros((h, w, 4), np.uint8)
        gb2 = np.zeros((h, w, 4), np.uint16)
        gb3 = np.zeros((h, w, 4), np.uint16)
        depth = np.zeros((h, w), np.float32)
        lib().orc_gbuffer_raycast(self.h, _ubo_ptr(ubo), C.c_int(w), C.c_int(h), _p(gb1, c_u8p), _p(gb2, c_u16p), _p(gb3, c_u16p), _p(depth, c_f32p))
        return dict(gb1=gb1, gb2=gb2, gb3=gb3, depth=depth)


class InstancedScene(Scene):
    """Oracle mirror of an instanced scene (scene_descriptor_set.glsl:30-34, :150-160; main.cpp:74): the BVH2 is built over the WORLD-space
    vertices orc_instances_flatten produces (model_matrix * vec4(p, 1), rows summed left to right), the hit shading interpolates the
    object-space attributes and then applies the matrix.  update(matrices) = the per-frame TLAS rebuild: a fresh oracle scene."""

    def __init__(self, isd, matrices=None):
        self.isd = isd
        self._arr = isd.mesh_arrays()
        self._layout = isd.layout()
        self._build(isd.matrices() if matrices is None else np.ascontiguousarray(np.asarray(matrices, np.float32).reshape(-1, 16)))

    def flatten(self, mats):
        """(world positions, world normals) [N][3][3] by the oracle's C arithmetic"""
        first, mbase, mid, n = self._layout
        N = int(n.sum())
        pos, nor = np.zeros((N, 3, 3), np.float32), np.zeros((N, 3, 3), np.float32)
        lib().orc_instances_flatten(C.c_int(len(n)), _p(mats, c_f32p), _p(first, c_u32p), _p(mbase, c_u32p), _p(n, c_u32p), _p(self._arr["positions"], c_f32p),
                                    _p(self._arr["normals"], c_f32p), _p(pos, c_f32p), _p(nor, c_f32p))
        return pos, nor

    def _build(self, mats):
        import dataclasses
        self.matrices = mats
        first, mbase, mid, n = self._layout
        pos, nor = self.flatten(mats)
        flat = self.isd.flatten(mats)    # numpy restatement: material / mesh-id / texture arrays; its vertices must equal the C ones
        assert np.array_equal(flat.verts.view(np.uint32), pos.view(np.uint32)) and np.array_equal(flat.normals.view(np.uint32), nor.view(np.uint32)), "numpy and C flatten disagree"
        Scene.__init__(self, dataclasses.replace(flat, verts=pos, normals=nor))
        a = self._arr
        lib().orc_scene_set_instances(self.h, C.c_int(len(n)), _p(mats, c_f32p), _p(first, c_u32p), _p(mbase, c_u32p), _p(mid, c_u32p), _p(n, c_u32p),
                                      C.c_int(len(a["positions"])), _p(a["positions"], c_f32p), _p(a["normals"], c_f32p), _p(a["material"], c_u32p),
                                      _p(a["uvs"], c_f32p), _p(a["tangents"], c_f32p))

    def update(self, matrices):
        if self.h:
            lib().orc_scene_destroy(self.h)
            self.h = None
        self._build(np.ascontiguousarray(np.asarray(matrices, np.float32).reshape(-1, 16)))


# ---------------------------------------------------------------------------------- shadows

def shadows_ray_trace(scene: Scene, ubo, depth, gb2, sobol, sr, bias=0.5, num_frames=0):
    h, w = depth.shape
    mask = np.zeros(((h + 3) // 4, (w + 7) // 8), np.uint32)
    rays = C.c_uint64(0)
    lib().orc_shadows_ray_trace(scene.h, _ubo_ptr(ubo), C.c_int(w), C.c_int(h), _p(depth, c_f32p), _p(gb2, c_u16p), _p(sobol, c_u8p),
                                _p(sr, c_u8p), C.c_float(bias), C.c_uint32(num_frames), _p(mask, c_u32p), C.byref(rays))
    return mask, rays.value


def shadows_gen_rays(ubo, depth, gb2, sobol, sr, bias=0.5, num_frames=0):
    h, w = depth.shape
    rays = np.zeros((h * w, 8), np.float32)
    lib().orc_shadows_gen_rays(_ubo_ptr(ubo), C.c_int(w), C.c_int(h), _p(depth, c_f32p), _p(gb2, c_u16p), _p(sobol, c_u8p), _p(sr, c_u8p),
                               C.c_float(bias), C.c_uint32(num_frames), _p(rays, c_f32p))
    return rays


def shadows_temporal(ubo, mask, cur, prev, hist_vis_var, hist_moments, alpha=0.01, moments_alpha=0.2):
    h, w = cur["depth"].shape
    out = np.zeros((h, w, 2), np.uint16)
    mom = np.zeros((h, w, 4), np.uint16)
    tiles = np.zeros(((h + 7) // 8, (w + 7) // 8), np.uint8)
    lib().orc_shadows_temporal(_ubo_ptr(ubo), C.c_int(w), C.c_int(h), _p(mask, c_u32p), _p(cur["depth"], c_f32p), _p(cur["gb2"], c_u16p),
                               _p(cur["gb3"], c_u16p), _p(prev["depth"], c_f32p), _p(prev["gb2"], c_u16p), _p(prev["gb3"], c_u16p),
                               _p(hist_vis_var, c_u16p), _p(hist_moments, c_u16p), C.c_float(alpha), C.c_float(moments_alpha),
                               _p(out, c_u16p), _p(mom, c_u16p), _p(tiles, c_u8p))
    return out, mom, tiles


def shadows_atrous(inp, gb2, gb3, tiles, step, radius=1, phi_visibility=10.0, phi_normal=32.0, sigma_depth=1.0, power=0.0):
    h, w = inp.shape[:2]
    out = np.zeros((h, w, 2), np.uint16)
    lib().orc_shadows_atrous(C.c_int(w), C.c_int(h), _p(inp, c_u16p), _p(gb2, c_u16p), _p(gb3, c_u16p), _p(tiles, c_u8p), C.c_int(radius),
                             C.c_int(step), C.c_float(phi_visibility), C.c_float(phi_normal), C.c_float(sigma_depth), C.c_float(power),
                             _p(out, c_u16p))
    return out


def upsample(full, mip, lowres, channels=1, sky_value=0.0, power=0.0):
    H, W = full["gb2"].shape[:2]
    h, w = mip["gb2"].shape[:2]
    in_ch = lowres.shape[2] if lowres.ndim == 3 else 1
    out = np.zeros((H, W, channels), np.uint16)
    lib().orc_upsample(C.c_int(W), C.c_int(H), C.c_int(w), C.c_int(h), _p(full["gb2"], c_u16p), _p(full["gb3"], c_u16p), _p(mip["gb2"], c_u16p),
                       _p(mip["gb3"], c_u16p), _p(lowres, c_u16p), C.c_int(in_ch), C.c_int(channels), C.c_float(sky_value), C.c_float(power),
                       _p(out, c_u16p))
    return out


class ShadowsPass:
    """Host-side sequencing of RayTracedShadows::render (ray_traced_shadows.cpp:100-116) on the oracle."""

    def __init__(self, w, h, bias=0.5, alpha=0.01, moments_alpha=0.2, phi_visibility=10.0, phi_normal=32.0, sigma_depth=1.0,
                 power=1.2, radius=1, filter_iterations=4, feedback_iteration=1):
        self.w, self.h = w, h
        self.p = dict(bias=bias, alpha=alpha, moments_alpha=moments_alpha, phi_visibility=phi_visibility, phi_normal=phi_normal,
                      sigma_depth=sigma_depth, power=power, radius=radius, filter_iterations=filter_iterations,
                      feedback_iteration=feedback_iteration)
        self.prev_image = np.zeros((h, w, 2), np.uint16)      # clear_images(): zero (:938-968)
        self.moments = np.zeros((h, w, 4), np.uint16)
        self.stages = {}

    def render(self, scene, ubo, cur, prev, sobol, sr, num_frames):
        p = self.p
        mask, nrays = shadows_ray_trace(scene, ubo, cur["depth"], cur["gb2"], sobol, sr, p["bias"], num_frames)
        tv, mom, tiles = shadows_temporal(ubo, mask, cur, prev, self.prev_image, self.moments, p["alpha"], p["moments_alpha"])
        self.moments = mom
        img = tv
        atrous = []
        for i in range(p["filter_iterations"]):
            power = p["power"] if i == p["filter_iterations"] - 1 else 0.0
            img = shadows_atrous(img, cur["gb2"], cur["gb3"], tiles, 1 << i, p["radius"], p["phi_visibility"], p["phi_normal"],
                                 p["sigma_depth"], power)
            atrous.append(img)
            if i == p["feedback_iteration"]:
                self.prev_image = img.copy()
        self.stages = dict(mask=mask, rays=nrays, temporal=tv, moments=mom, tiles=tiles, atrous=atrous, output=img)
        return img


# ---------------------------------------------------------------------------------- ambient occlusion

def ao_ray_trace(scene, ubo, depth, gb2, sobol, sr, bias=0.3, ray_length=7.0, num_frames=0, spp=1):
    h, w = depth.shape
    mask = np.zeros((spp, (h + 3) // 4, (w + 7) // 8), np.uint32)
    rays = C.c_uint64(0)
    lib().orc_ao_ray_trace(scene.h, _ubo_ptr(ubo), C.c_int(w), C.c_int(h), _p(depth, c_f32p), _p(gb2, c_u16p), _p(sobol, c_u8p), _p(sr, c_u8p),
                           C.c_float(bias), C.c_float(ray_length), C.c_uint32(num_frames), C.c_int(spp), _p(mask, c_u32p), C.byref(rays))
    return mask, rays.value


def ao_temporal(ubo, mask, cur, prev, hist_ao, hist_len, alpha=0.01):
    h, w = cur["depth"].shape
    out, ln = np.zeros((h, w), np.uint16), np.zeros((h, w), np.uint16)
    tiles = np.zeros(((h + 7) // 8, (w + 7) // 8), np.uint8)
    lib().orc_ao_temporal(_ubo_ptr(ubo), C.c_int(w), C.c_int(h), C.c_int(mask.shape[0]), _p(mask, c_u32p), _p(cur["depth"], c_f32p),
                          _p(cur["gb2"], c_u16p), _p(cur["gb3"], c_u16p), _p(prev["depth"], c_f32p), _p(prev["gb2"], c_u16p),
                          _p(prev["gb3"], c_u16p), _p(hist_ao, c_u16p), _p(hist_len, c_u16p), C.c_float(alpha), _p(out, c_u16p),
                          _p(ln, c_u16p), _p(tiles, c_u8p))
    return out, ln, tiles


def ao_blur(inp, depth, gb2, tiles, zbp, direction, radius=4):
    h, w = inp.shape
    out = np.zeros((h, w), np.uint16)
    zbp = np.ascontiguousarray(zbp, np.float32)
    lib().orc_ao_blur(C.c_int(w), C.c_int(h), _p(inp, c_u16p), _p(depth, c_f32p), _p(gb2, c_u16p), _p(tiles, c_u8p), _p(zbp, c_f32p),
                      C.c_int(direction[0]), C.c_int(direction[1]), C.c_int(radius), _p(out, c_u16p))
    return out


class AOPass:
    """Host-side sequencing of RayTracedAO::render (ray_traced_ao.cpp:98-112) on the oracle."""

    def __init__(self, w, h, bias=0.3, ray_length=7.0, alpha=0.01, blur_radius=4, power=1.2, spp=1, zbp=(-0.999, 1.0, -0.999, 1.0)):
        self.w, self.h = w, h
        self.p = dict(bias=bias, ray_length=ray_length, alpha=alpha, blur_radius=blur_radius, power=power, spp=spp)
        self.zbp = np.asarray(zbp, np.float32)
        self.hist_ao = np.zeros((h, w), np.uint16)    # clear_images(): zero (:831-857)
        self.hist_len = np.zeros((h, w), np.uint16)
        self.stages = {}

    def render(self, scene, ubo, cur, prev, sobol, sr, num_frames, full=None):
        p = self.p
        mask, nrays = ao_ray_trace(scene, ubo, cur["depth"], cur["gb2"], sobol, sr, p["bias"], p["ray_length"], num_frames, p["spp"])
        out, ln, tiles = ao_temporal(ubo, mask, cur, prev, self.hist_ao, self.hist_len, p["alpha"])
        self.hist_ao, self.hist_len = out, ln
        b0 = ao_blur(out, cur["depth"], cur["gb2"], tiles, self.zbp, (1, 0), p["blur_radius"])
        b1 = ao_blur(b0, cur["depth"], cur["gb2"], tiles, self.zbp, (0, 1), p["blur_radius"])
        up = None
        if full is not None:
            up = upsample(full, cur, b1[..., None], channels=1, sky_value=1.0, power=p["power"])
        self.stages = dict(mask=mask, rays=nrays, temporal=out, length=ln, tiles=tiles, blur0=b0, blur1=b1, upsample=up, output=b1 if up is None else up)
        return self.stages["output"]


def f16(a):
    """uint16 fp16 bit patterns -> float32 values."""
    return np.asarray(a, np.uint16).view(np.float16).astype(np.float32)
