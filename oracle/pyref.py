"""ORACLE — TEST INFRASTRUCTURE ONLY.

Runs the REFERENCE'S OWN SHADERS on the CPU: oracle/refshim/translate.py rewrites a shader from
/root/reference/src/shaders into C++ for the GLSL shim (oracle/refshim/glsl.h), g++ compiles it into
oracle/_ref/<name>.so, and this module binds numpy arrays to its descriptors and dispatches it.  Used by
tests/test_ref_shaders.py (oracle restatement == reference shader, bit for bit) and by tests/golden/make_ref_golden.py
(fixtures the GPU tests check the HIP kernels against).  /root/reference is only needed to (re)build; a prebuilt
oracle/_ref/*.so is used as is.  Nothing in the product imports this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF_DIR = os.path.join(_HERE, "_ref")
_SHIM = os.path.join(_HERE, "refshim")
REFERENCE_SHADERS = "/root/reference/src/shaders"

FMT = dict(r8=0, rgba8=1, r16f=2, rg16f=3, rgba16f=4, r32f=5, rg32f=6, rgba32f=7, r32ui=8)
_FMT_DTYPE = {0: (np.uint8, 1), 1: (np.uint8, 4), 2: (np.uint16, 1), 3: (np.uint16, 2), 4: (np.uint16, 4), 5: (np.float32, 1), 6: (np.float32, 2),
              7: (np.float32, 4), 8: (np.uint32, 1)}


class _TexLevel(C.Structure):
    _fields_ = [("data", C.c_void_p), ("w", C.c_int), ("h", C.c_int)]


class _Texture(C.Structure):
    _fields_ = [("fmt", C.c_int), ("n_levels", C.c_int), ("linear", C.c_int), ("repeat", C.c_int), ("layers", C.c_int), ("pad", C.c_int),
                ("lv", _TexLevel * 16)]


class Tex:
    """A numpy array (or a mip chain of them) seen as a GLSL image / sampler.  fp16 formats take uint16 bit patterns."""

    def __init__(self, levels, fmt, linear=False, repeat=False, layers=1):
        if isinstance(levels, np.ndarray):
            levels = [levels]
        dt, ch = _FMT_DTYPE[FMT[fmt]]
        self.levels = []
        self.c = _Texture(fmt=FMT[fmt], n_levels=len(levels), linear=int(linear), repeat=int(repeat), layers=layers)
        for i, a in enumerate(levels):
            assert a.dtype == dt and a.flags["C_CONTIGUOUS"], (a.dtype, dt)
            h, w = a.shape[:2]
            assert a.size == h * w * ch, (a.shape, ch)
            self.levels.append(a)
            self.c.lv[i] = _TexLevel(a.ctypes.data, w, h)

    @property
    def ptr(self):
        return C.addressof(self.c)


def available() -> bool:
    return os.path.isdir(REFERENCE_SHADERS) or os.path.isdir(_REF_DIR)


def build(rel: str, defines=(), force=False) -> str:
    """translate + compile one reference shader; returns the .so path (a prebuilt one is reused when the reference is absent)"""
    sys.path.insert(0, _SHIM)
    import translate  # noqa: E402
    sys.path.pop(0)
    import re
    name = re.sub(r"\W", "_", os.path.splitext(os.path.basename(rel))[0]) + "".join("_" + re.sub(r"\W", "_", d.split("=")[0]) for d in defines)
    so = os.path.join(_REF_DIR, name + ".so")
    if not os.path.isdir(REFERENCE_SHADERS):
        if not os.path.exists(so):
            raise FileNotFoundError(f"{so}: not prebuilt and /root/reference is absent")
        return so
    deps = [os.path.join(_SHIM, f) for f in ("glsl.h", "runtime.inc", "translate.py", "swizzles.inc")] + [os.path.join(REFERENCE_SHADERS, rel)]
    if force or not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        if not os.path.exists(os.path.join(_SHIM, "swizzles.inc")):
            subprocess.check_call([sys.executable, os.path.join(_SHIM, "gen_swizzles.py")])
        src = translate.translate(rel, list(defines))
        _compile(src, so)
    return so


def _compile(src, so):
    # GLSL evaluates function and constructor arguments left to right (vec2(next_float(rng), next_float(rng)) in the
    # reference's random.glsl depends on it); C++ leaves the order unspecified.  clang evaluates left to right, g++ right
    # to left, so the translated shaders are built with ROCm's clang; RefShader checks the order at load time.
    cxx = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(cxx):
        cxx = "clang++"
    subprocess.check_call([cxx, "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-w", "-I" + _SHIM, "-o", so, src])


def build_pipeline(name: str, stages, force=False, variant=None) -> str:
    """a ray-tracing pipeline: stages = [(rel, kind)], kind 0 = ray generation, 1 = closest hit, 2 = miss"""
    so = os.path.join(_REF_DIR, name + ".so")
    if not os.path.isdir(REFERENCE_SHADERS):
        if not os.path.exists(so):
            raise FileNotFoundError(f"{so}: not prebuilt and /root/reference is absent")
        return so
    deps = [os.path.join(_SHIM, f) for f in ("glsl.h", "runtime.inc", "runtime_rt.inc", "translate.py", "swizzles.inc")]
    deps += [os.path.join(REFERENCE_SHADERS, r) for r, _ in stages]
    if force or not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        sys.path.insert(0, _SHIM)
        import translate  # noqa: E402
        sys.path.pop(0)
        _compile(translate.translate_pipeline(name, list(stages), variant=variant), so)
    return so


class RefShader:
    """One compiled reference shader: set block members / bind textures by their GLSL names, then dispatch."""

    def __init__(self, rel: str, defines=(), so=None):
        self.lib = C.CDLL(so or build(rel, defines))

        class Reg(C.Structure):
            _fields_ = [("name", C.c_char_p), ("ptr", C.c_void_p), ("size", C.c_size_t)]
        assert self.lib.ref_eval_order_ok() == 1, "translated shader was compiled with right-to-left argument evaluation"
        self.lib.ref_regs.restype = C.POINTER(Reg)
        n = C.c_int(0)
        regs = self.lib.ref_regs(C.byref(n))
        self.regs = {regs[i].name.decode(): (regs[i].ptr, regs[i].size) for i in range(n.value)}
        ls = (C.c_int * 3)()
        self.lib.ref_local_size(ls)
        self.local_size = tuple(ls)
        self._keep = {}

    def set(self, name, value):
        """value: numpy scalar/array or bytes with exactly the member's size (matrices column-major)"""
        ptr, size = self.regs[name]
        b = value if isinstance(value, (bytes, bytearray)) else np.ascontiguousarray(value).tobytes()
        assert len(b) == size, f"{name}: {len(b)} bytes given, member has {size}"
        C.memmove(ptr, b, size)

    def set_at(self, name, value, offset=0):
        """writes value's bytes at a byte offset inside a registered variable (descriptor arrays)"""
        ptr, size = self.regs[name]
        b = value if isinstance(value, (bytes, bytearray)) else np.ascontiguousarray(value).tobytes()
        assert offset + len(b) <= size
        C.memmove(ptr + offset, b, len(b))

    def set_f(self, name, v):
        self.set(name, np.float32(v))

    def set_i(self, name, v):
        self.set(name, np.int32(v))

    def set_u(self, name, v):
        self.set(name, np.uint32(v))

    def bind(self, name, tex: Tex):
        self._keep[name] = tex
        self.set(name, np.uint64(tex.ptr))

    def bind_buffer(self, name, arr: np.ndarray):
        """unsized-array member of a buffer block"""
        self._keep[name] = arr
        self.set(name, np.uint64(arr.ctypes.data))

    def set_any_hit(self, fn):
        self._keep["any_hit"] = fn
        self.lib.ref_set_any_hit(fn)

    def fragments(self, w, h, tex_coord_name, out_name):
        """full-screen pass of a fragment shader; returns the vec4 output as float32 [h][w][4]"""
        out = np.zeros((h, w, 4), np.float32)
        self.lib.ref_fragments(C.c_int(w), C.c_int(h), C.c_void_p(self.regs[tex_coord_name][0]), C.c_void_p(self.regs[out_name][0]),
                               out.ctypes.data_as(C.c_void_p))
        return out

    def dispatch(self, gx, gy=1, gz=1):
        self.lib.ref_dispatch(C.c_int(gx), C.c_int(gy), C.c_int(gz))


class RefPipeline(RefShader):
    """A ray-tracing pipeline (ray generation + closest hit + miss stages in one library).  Registry names carry the
    stage index ('0:ubo.view_proj', '1:s_Cubemap'); the *_all helpers address a name in every stage that declares it."""

    def __init__(self, name, stages, variant=None):
        super().__init__(None, so=build_pipeline(name, stages, variant=variant))
        self.n_stages = len(stages)

    def stages_with(self, name):
        return [i for i in range(self.n_stages) if f"{i}:{name}" in self.regs]

    def set_all(self, name, value):
        for i in self.stages_with(name):
            self.set(f"{i}:{name}", value)

    def bind_all(self, name, tex):
        for i in self.stages_with(name):
            self.bind(f"{i}:{name}", tex)

    def set_at_all(self, name, value, offset=0):
        for i in self.stages_with(name):
            self.set_at(f"{i}:{name}", value, offset)

    def trace_rays(self, w, h, d=1):
        self.lib.ref_trace_rays(C.c_int(w), C.c_int(h), C.c_int(d))


# every reference shader the tests run: the recipe behind oracle/_ref/
COMPUTE_SHADERS = (
    "shadows/shadows_ray_trace.comp", "shadows/shadows_denoise_reprojection.comp", "shadows/shadows_denoise_copy_shadow_tiles.comp",
    "shadows/shadows_denoise_atrous.comp", "shadows/shadows_upsample.comp",
    "ao/ao_ray_trace.comp", "ao/ao_denoise_reprojection.comp", "ao/ao_denoise_bilateral_blur.comp", "ao/ao_upsample.comp",
    "reflections/reflections_denoise_reprojection.comp", "reflections/reflections_denoise_copy_tiles.comp",
    "reflections/reflections_denoise_atrous.comp", "reflections/reflections_upsample.comp",
    "gi/gi_irradiance_probe_update.comp", "gi/gi_depth_probe_update.comp", "gi/gi_irradiance_border_update.comp",
    "gi/gi_depth_border_update.comp", "gi/gi_sample_probe_grid.comp", "taa.comp", "deferred.frag", "tone_map.frag")
PIPELINES = {
    "gi_ray_trace": [("gi/gi_ray_trace.rgen", 0), ("gi/gi_ray_trace.rchit", 1), ("gi/gi_ray_trace.rmiss", 2)],
    "reflections_ray_trace": [("reflections/reflections_ray_trace.rgen", 0), ("reflections/reflections_ray_trace.rchit", 1),
                              ("reflections/reflections_ray_trace.rmiss", 2)],
    "ground_truth_path_trace": [("ground_truth/ground_truth_path_trace.rgen", 0), ("ground_truth/ground_truth_path_trace.rchit", 1),
                                ("ground_truth/ground_truth_path_trace.rmiss", 2)],
}


def build_all(force=False):
    """translate + compile every reference shader into oracle/_ref/ (needs /root/reference); returns the .so paths"""
    return ([build(rel, force=force) for rel in COMPUTE_SHADERS] + [build_pipeline(n, st, force=force) for n, st in PIPELINES.items()]
            + [build_pipeline("ground_truth_path_trace_bounces", PIPELINES["ground_truth_path_trace"], force=force, variant="bounces")])


if __name__ == "__main__":
    for so in build_all("--force" in sys.argv):
        print(so)
