"""Host-side mirror of the reference's pass classes over the C ABI (include/hr_api.h).

Class / method names follow the reference (src/ray_traced_shadows.h etc.): a pass is constructed
for a resolution + scale, ``render()`` records one frame on a HIP stream, ``output()`` replaces
``output_ds()``.  torch is used only for device memory and streams; every kernel is the in-tree HIP
library ``libhybrid_rendering_amd.so``.  There is NO CPU fallback: if the library cannot be loaded
the import fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# HR_LIBRARY: developer A/B builds (python -m hybrid_rendering_amd.build --variant NAME with HR_CFLAGS=...) — a HIP library either way
LIB_PATH = os.environ.get("HR_LIBRARY") or os.path.join(_HERE, "libhybrid_rendering_amd.so")

# ------------------------------------------------------------------------------ ctypes structs


class hr_light(C.Structure):
    _fields_ = [("data0", C.c_float * 4), ("data1", C.c_float * 4), ("data2", C.c_float * 4), ("data3", C.c_float * 4)]


class hr_ubo(C.Structure):
    _fields_ = [("view_inverse", C.c_float * 16), ("proj_inverse", C.c_float * 16), ("view_proj_inverse", C.c_float * 16),
                ("prev_view_proj", C.c_float * 16), ("view_proj", C.c_float * 16), ("cam_pos", C.c_float * 4),
                ("current_prev_jitter", C.c_float * 4), ("light", hr_light)]


assert C.sizeof(hr_ubo) == 416


class hr_gbuffer_level(C.Structure):
    _fields_ = [("gb1", C.c_void_p), ("gb2", C.c_void_p), ("gb3", C.c_void_p), ("depth", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32)]


class hr_frame_inputs(C.Structure):
    _fields_ = [("cur", hr_gbuffer_level), ("prev", hr_gbuffer_level), ("cur_full", hr_gbuffer_level), ("ubo", hr_ubo),
                ("num_frames", C.c_uint32), ("ping_pong", C.c_int32), ("sobol", C.c_void_p), ("scrambling_ranking", C.c_void_p),
                ("z_buffer_params", C.c_float * 4)]


class hr_image_view(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("row_pitch_bytes", C.c_int32), ("format", C.c_int)]


class hr_texture(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32)]


class hr_scene_desc(C.Structure):
    _fields_ = [("positions", C.c_void_p), ("normals", C.c_void_p), ("tri_material", C.c_void_p), ("tri_mesh_id", C.c_void_p),
                ("n_tris", C.c_int32), ("materials", C.c_void_p), ("n_materials", C.c_int32),
                ("uvs", C.c_void_p), ("tangents", C.c_void_p), ("material_textures", C.c_void_p), ("textures", C.POINTER(hr_texture)),
                ("n_textures", C.c_int32)]


class hr_scene_info(C.Structure):
    _fields_ = [("n_tris", C.c_int32), ("n_nodes", C.c_int32), ("max_depth", C.c_int32), ("node_bytes", C.c_uint64), ("tri_bytes", C.c_uint64),
                ("bounds_lo", C.c_float * 3), ("bounds_hi", C.c_float * 3), ("box_pad", C.c_float)]


class hr_band(C.Structure):
    _fields_ = [("band_y0", C.c_int32), ("band_y1", C.c_int32), ("halo", C.c_int32), ("history_halo", C.c_int32)]


HR_MAX_STAGES = 16


class hr_stage_times(C.Structure):
    _fields_ = [("n_stages", C.c_int32), ("name", C.c_char_p * HR_MAX_STAGES), ("ms", C.c_float * HR_MAX_STAGES), ("bytes", C.c_uint64 * HR_MAX_STAGES)]


class hr_shadows_params(C.Structure):
    _fields_ = [("denoise", C.c_int32), ("bias", C.c_float), ("alpha", C.c_float), ("moments_alpha", C.c_float), ("phi_visibility", C.c_float),
                ("phi_normal", C.c_float), ("sigma_depth", C.c_float), ("power", C.c_float), ("radius", C.c_int32),
                ("filter_iterations", C.c_int32), ("feedback_iteration", C.c_int32), ("exact", C.c_int32)]


class hr_ao_params(C.Structure):
    _fields_ = [("denoise", C.c_int32), ("ray_length", C.c_float), ("bias", C.c_float), ("alpha", C.c_float), ("blur_radius", C.c_int32),
                ("power", C.c_float), ("spp", C.c_int32), ("exact", C.c_int32)]


HR_FORMAT = {1: ("R32_UINT", 4), 2: ("R16F", 2), 3: ("RG16F", 4), 4: ("RGBA16F", 8), 5: ("R32F", 4), 6: ("RGBA8", 4), 0: ("R8", 1)}
OUTPUT_RAY_TRACE, OUTPUT_TEMPORAL_ACCUMULATION, OUTPUT_ATROUS, OUTPUT_UPSAMPLE = 0, 1, 2, 3
SCALE_FULL_RES, SCALE_HALF_RES, SCALE_QUARTER_RES = 0, 1, 2

# every symbol include/hr_api.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "hr_status_string", "hr_last_error", "hr_version", "hr_api_revision", "hr_ctx_create", "hr_ctx_destroy", "hr_ctx_device", "hr_scene_create", "hr_scene_get_info", "hr_scene_id", "hr_scene_create_instanced", "hr_scene_update_instances", "hr_scene_instance_count", "hr_scene_rebuild_top_level", "hr_scene_top_level_rebuilds", "hr_scene_read_bvh", "hr_set_markers", "hr_markers_log",
    "hr_scene_destroy", "hr_trace_any_hit", "hr_trace_closest_hit", "hr_gbuffer_raycast", "hr_shadows_default_params", "hr_shadows_create",
    "hr_shadows_render", "hr_shadows_output", "hr_shadows_reset_history", "hr_shadows_destroy", "hr_shadows_ray_trace", "hr_shadows_denoise", "hr_shadows_temporal",
    "hr_shadows_atrous_iteration", "hr_shadows_upsample", "hr_shadows_image", "hr_shadows_history_apron_exceeded", "hr_shadows_set_profiling", "hr_shadows_get_stage_times",
    "hr_gbuffer_mip_nearest", "hr_bvh_build_info", "hr_bvh_selfcheck", "hr_shadows_ray_count", "hr_shadows_tile_ray_counts", "hr_shadows_trace_stats", "hr_shadows_trace_stats_timed", "hr_shadows_launch_order", "hr_shadows_trace_divergence", "hr_selftest_math",
]

_lib = None


class HRError(RuntimeError):
    pass


def lib():
    """Load the HIP library.  Raises if it is missing — there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HRError(f"{LIB_PATH} not built: run `python -m hybrid_rendering_amd.build` (hipcc, gfx950). No CPU fallback exists.")
        try:
            import torch  # noqa: F401  (loads the HIP runtime first so both share one libamdhip64)
        except Exception:
            pass
        _lib = C.CDLL(LIB_PATH)
        _lib.hr_status_string.restype = C.c_char_p
        _lib.hr_last_error.restype = C.c_char_p
        _lib.hr_version.restype = C.c_char_p
        _lib.hr_scene_id.restype = C.c_uint64
        _lib.hr_scene_id.argtypes = [C.c_void_p]
    return _lib


def _check(status: int, what: str):
    if status != 0:
        L = lib()
        raise HRError(f"{what}: {L.hr_status_string(status).decode()} — {L.hr_last_error().decode()}")


def make_ubo(np_ubo: np.ndarray) -> hr_ubo:
    assert np_ubo.nbytes == 416
    u = hr_ubo()
    C.memmove(C.byref(u), np_ubo.ctypes.data, 416)
    return u


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream_ptr(stream=None):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


# ------------------------------------------------------------------------------ context / scene


class hr_mesh_desc(C.Structure):
    _fields_ = [("positions", C.c_void_p), ("normals", C.c_void_p), ("tri_material", C.c_void_p), ("uvs", C.c_void_p), ("tangents", C.c_void_p), ("n_tris", C.c_int32)]


class hr_instance(C.Structure):
    _fields_ = [("model_matrix", C.c_float * 16), ("mesh_idx", C.c_uint32), ("mesh_id", C.c_uint32)]


class hr_instanced_scene_desc(C.Structure):
    _fields_ = [("meshes", C.POINTER(hr_mesh_desc)), ("n_meshes", C.c_int32), ("instances", C.POINTER(hr_instance)), ("n_instances", C.c_int32),
                ("materials", C.c_void_p), ("n_materials", C.c_int32), ("material_textures", C.c_void_p), ("textures", C.c_void_p), ("n_textures", C.c_int32)]


class Context:
    def __init__(self, device: int = 0):
        self.h = C.c_void_p()
        _check(lib().hr_ctx_create(C.c_int(device), C.byref(self.h)), "hr_ctx_create")
        self.device = device

    def close(self):
        if self.h:
            lib().hr_ctx_destroy(self.h)
            self.h = C.c_void_p()


class Scene:
    """Replaces dw::RayTracedScene: host triangles -> compressed 8-wide BVH in HBM."""

    def __init__(self, ctx: Context, sd):
        self.ctx = ctx
        self._keep = [np.ascontiguousarray(sd.verts, np.float32), None if sd.normals is None else np.ascontiguousarray(sd.normals, np.float32),
                      np.ascontiguousarray(sd.tri_material, np.uint32), np.ascontiguousarray(sd.tri_mesh_id, np.uint32),
                      np.ascontiguousarray(sd.materials, np.float32)]
        v, n, m, i, mats = self._keep
        d = hr_scene_desc(v.ctypes.data, n.ctypes.data if n is not None else None, m.ctypes.data, i.ctypes.data, sd.n_tris, mats.ctypes.data, len(mats))
        if getattr(sd, "material_textures", None) is not None:        # textured materials (optional)
            uv = None if sd.uvs is None else np.ascontiguousarray(sd.uvs, np.float32)
            tg = None if sd.tangents is None else np.ascontiguousarray(sd.tangents, np.float32)
            mt = np.ascontiguousarray(sd.material_textures, np.int32)
            tex = [np.ascontiguousarray(t, np.uint8) for t in sd.textures]
            arr = (hr_texture * len(tex))(*[hr_texture(t.ctypes.data, t.shape[1], t.shape[0]) for t in tex])
            self._keep += [uv, tg, mt, tex, arr]
            d.uvs, d.tangents = (uv.ctypes.data if uv is not None else None), (tg.ctypes.data if tg is not None else None)
            d.material_textures, d.textures, d.n_textures = mt.ctypes.data, arr, len(tex)
        self.h = C.c_void_p()
        _check(lib().hr_scene_create(ctx.h, C.byref(d), C.byref(self.h)), "hr_scene_create")
        self.info = hr_scene_info()
        _check(lib().hr_scene_get_info(self.h, C.byref(self.info)), "hr_scene_get_info")

    @property
    def id(self) -> int:
        """dw::Scene::id() (hr_scene_id)"""
        return int(lib().hr_scene_id(self.h))

    def close(self):
        if self.h:
            lib().hr_scene_destroy(self.h)
            self.h = C.c_void_p()

    def read_bvh(self):
        """(nodes [n][80] uint8, triangle references [m][48] uint8): host copies of the device BVH (csrc/bvh.h layouts)"""
        self.refresh_info()
        nodes, tris = np.zeros((self.info.n_nodes, 80), np.uint8), np.zeros((int(self.info.tri_bytes) // 48, 48), np.uint8)
        _check(lib().hr_scene_read_bvh(self.h, C.c_void_p(nodes.ctypes.data), C.c_void_p(tris.ctypes.data)), "hr_scene_read_bvh")
        return nodes, tris

    def refresh_info(self):
        _check(lib().hr_scene_get_info(self.h, C.byref(self.info)), "hr_scene_get_info")
        return self.info

    def any_hit(self, rays, stats=False, stream=None):
        """rays: cuda float32 [n,8] (origin, t_max, dir, t_min) -> uint8 [n] (1 = occluded)."""
        import torch
        rays = rays.contiguous()
        out = torch.zeros(rays.shape[0], dtype=torch.uint8, device=rays.device)
        st = torch.zeros(2, dtype=torch.int64, device=rays.device) if stats else None
        _check(lib().hr_trace_any_hit(self.h, C.c_int64(rays.shape[0]), _ptr(rays), _ptr(out), _ptr(st), _stream_ptr(stream)), "hr_trace_any_hit")
        return (out, st) if stats else out

    def closest_hit(self, rays, stream=None):
        import torch
        rays = rays.contiguous()
        tuv = torch.zeros((rays.shape[0], 3), dtype=torch.float32, device=rays.device)
        prim = torch.zeros(rays.shape[0], dtype=torch.int32, device=rays.device)
        _check(lib().hr_trace_closest_hit(self.h, C.c_int64(rays.shape[0]), _ptr(rays), _ptr(tuv), _ptr(prim), _stream_ptr(stream)), "hr_trace_closest_hit")
        return tuv, prim

    def gbuffer(self, np_ubo, w, h, device="cuda", stream=None):
        """GPU G-buffer synthesis (stands in for the raster GBuffer pass).  Returns dict of cuda tensors."""
        import torch
        gb1 = torch.zeros((h, w, 4), dtype=torch.uint8, device=device)
        gb2 = torch.zeros((h, w, 4), dtype=torch.float16, device=device)
        gb3 = torch.zeros((h, w, 4), dtype=torch.float16, device=device)
        depth = torch.zeros((h, w), dtype=torch.float32, device=device)
        u = make_ubo(np_ubo)
        _check(lib().hr_gbuffer_raycast(self.h, C.byref(u), C.c_int32(w), C.c_int32(h), _ptr(gb1), _ptr(gb2), _ptr(gb3), _ptr(depth), _stream_ptr(stream)),
               "hr_gbuffer_raycast")
        return dict(gb1=gb1, gb2=gb2, gb3=gb3, depth=depth)


class InstancedScene(Scene):
    """dw::RayTracedScene as the reference holds it — meshes + instances — with the per-frame update of main.cpp:74 (build_tlas):
    hr_scene_create_instanced / hr_scene_update_instances.  ``isd``: synth.InstancedSceneData.  Every pass takes it like a Scene."""

    def __init__(self, ctx: Context, isd):
        self.ctx, self.isd = ctx, isd
        f32 = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
        keep, meshes = [], (hr_mesh_desc * len(isd.meshes))()
        for k, m in enumerate(isd.meshes):
            arrs = [f32(m.verts), f32(m.normals), np.ascontiguousarray(m.tri_material, np.uint32), f32(m.uvs), f32(m.tangents)]
            keep.append(arrs)
            meshes[k] = hr_mesh_desc(*[(a.ctypes.data if a is not None else None) for a in arrs], m.n_tris)
        inst = (hr_instance * len(isd.instances))()
        for i, (mat, mesh_idx, mesh_id) in enumerate(isd.instances):
            inst[i].model_matrix[:] = [float(v) for v in np.asarray(mat, np.float32).reshape(16)]
            inst[i].mesh_idx, inst[i].mesh_id = int(mesh_idx), int(mesh_id)
        mats = np.ascontiguousarray(isd.materials, np.float32)
        d = hr_instanced_scene_desc(meshes, len(isd.meshes), inst, len(isd.instances), mats.ctypes.data, len(mats), None, None, 0)
        if isd.material_textures is not None:
            mt = np.ascontiguousarray(isd.material_textures, np.int32)
            tex = [np.ascontiguousarray(t, np.uint8) for t in isd.textures]
            arr = (hr_texture * len(tex))(*[hr_texture(t.ctypes.data, t.shape[1], t.shape[0]) for t in tex])
            keep += [mt, tex, arr]
            d.material_textures, d.textures, d.n_textures = mt.ctypes.data, C.cast(arr, C.c_void_p), len(tex)
        self._keep = [keep, meshes, inst, mats]
        self.h = C.c_void_p()
        _check(lib().hr_scene_create_instanced(ctx.h, C.byref(d), C.byref(self.h)), "hr_scene_create_instanced")
        self.info = hr_scene_info()
        self.refresh_info()

    def rebuild_top_level(self, stream=None):
        _check(lib().hr_scene_rebuild_top_level(self.h, _stream_ptr(stream)), "hr_scene_rebuild_top_level")

    @property
    def top_level_rebuilds(self) -> int:
        return int(lib().hr_scene_top_level_rebuilds(self.h))

    def update(self, matrices, stream=None):
        """hr_scene_update_instances: matrices [n_instances][16] column-major (host); enqueued on ``stream`` (default: torch's current stream)"""
        m = np.ascontiguousarray(np.asarray(matrices, np.float32).reshape(-1, 16))
        assert m.shape[0] == lib().hr_scene_instance_count(self.h)
        _check(lib().hr_scene_update_instances(self.h, m.ctypes.data_as(C.POINTER(C.c_float)), _stream_ptr(stream)), "hr_scene_update_instances")


def bvh_build_info(verts) -> hr_scene_info:
    """Host-only BVH build (no GPU): the shape hr_scene_create would produce for triangles ``verts`` [n,3,3]."""
    v = np.ascontiguousarray(verts, np.float32)
    info = hr_scene_info()
    _check(lib().hr_bvh_build_info(v.ctypes.data_as(C.POINTER(C.c_float)), C.c_int32(v.shape[0]), C.byref(info)), "hr_bvh_build_info")
    return info


def bvh_selfcheck(verts, samples_per_triangle: int = 12) -> int:
    """Host-only: (triangle, surface point) pairs that the BVH built over ``verts`` fails to cover (0 for a correct tree)."""
    v = np.ascontiguousarray(verts, np.float32)
    bad = C.c_int64(-1)
    L = lib()
    L.hr_bvh_selfcheck.argtypes = [C.POINTER(C.c_float), C.c_int32, C.c_int32, C.POINTER(C.c_int64)]
    _check(L.hr_bvh_selfcheck(v.ctypes.data_as(C.POINTER(C.c_float)), C.c_int32(v.shape[0]), C.c_int32(samples_per_triangle), C.byref(bad)), "hr_bvh_selfcheck")
    return int(bad.value)


def gbuffer_mip(g, level, stream=None):
    """Nearest mip `level` of a G-buffer dict of cuda tensors (g_buffer.cpp:240-243) -> new dict at (H >> level, W >> level)."""
    import torch
    h, w = g["depth"].shape
    hh, ww = h >> level, w >> level
    out = dict(gb2=torch.empty((hh, ww, 4), dtype=torch.float16, device="cuda"), gb3=torch.empty((hh, ww, 4), dtype=torch.float16, device="cuda"),
               depth=torch.empty((hh, ww), dtype=torch.float32, device="cuda"))
    if g.get("gb1") is not None:
        out["gb1"] = torch.empty((hh, ww, 4), dtype=torch.uint8, device="cuda")
    src, dst = gbuffer_level(g), gbuffer_level(out)
    _check(lib().hr_gbuffer_mip_nearest(C.byref(src), C.byref(dst), C.c_int32(level), _stream_ptr(stream)), "hr_gbuffer_mip_nearest")
    return out


# ------------------------------------------------------------------------------ frame inputs


def gbuffer_level(g) -> hr_gbuffer_level:
    if g is None:
        return hr_gbuffer_level()
    h, w = g["depth"].shape
    return hr_gbuffer_level(_ptr(g.get("gb1")), _ptr(g["gb2"]), _ptr(g["gb3"]), _ptr(g["depth"]), w, h)


def frame_inputs(cur, prev, np_ubo, num_frames, ping_pong, sobol, scrambling_ranking, cur_full=None, z_buffer_params=(0, 0, 0, 0)) -> hr_frame_inputs:
    """cur/prev/cur_full: dicts of cuda tensors gb1 (u8 HxWx4), gb2/gb3 (f16 HxWx4), depth (f32 HxW)."""
    f = hr_frame_inputs()
    f.cur = gbuffer_level(cur)
    f.prev = gbuffer_level(prev if prev is not None else cur)
    f.cur_full = gbuffer_level(cur_full if cur_full is not None else cur)
    f.ubo = make_ubo(np_ubo)
    f.num_frames = int(num_frames)
    f.ping_pong = int(bool(ping_pong))
    f.sobol = _ptr(sobol)
    f.scrambling_ranking = _ptr(scrambling_ranking)
    for i in range(4):
        f.z_buffer_params[i] = float(z_buffer_params[i])
    f._keep = (cur, prev, cur_full, sobol, scrambling_ranking)
    return f


def view_to_tensor(v: hr_image_view, device="cuda"):
    """Zero-copy torch view of a pass-owned image (valid until the next render()/destroy)."""
    import torch
    name, bpp = HR_FORMAT[int(v.format)]
    # build a tensor from the raw device pointer via __cuda_array_interface__
    dt = {"R32_UINT": ("<i4", 1, torch.int32), "R16F": ("<f2", 1, torch.float16), "RG16F": ("<f2", 2, torch.float16),
          "RGBA16F": ("<f2", 4, torch.float16), "R32F": ("<f4", 1, torch.float32), "RGBA8": ("|u1", 4, torch.uint8), "R8": ("|u1", 1, torch.uint8)}[name]

    class _W:
        pass

    w = _W()
    shape = (v.height, v.width, dt[1]) if dt[1] > 1 else (v.height, v.width)
    w.__cuda_array_interface__ = dict(shape=shape, typestr=dt[0], data=(int(v.data), False), version=2)
    return torch.as_tensor(w, device=device)


# ------------------------------------------------------------------------------ RayTracedShadows


class _Pass:
    _prefix = ""

    def stage_times(self):
        st = hr_stage_times()
        _check(getattr(lib(), self._prefix + "_get_stage_times")(self.h, C.byref(st)), self._prefix + "_get_stage_times")
        return [(st.name[i].decode(), float(st.ms[i]), int(st.bytes[i])) for i in range(st.n_stages)]

    def set_profiling(self, on=True):
        _check(getattr(lib(), self._prefix + "_set_profiling")(self.h, C.c_int32(int(on))), self._prefix + "_set_profiling")

    def image(self, which):
        v = hr_image_view()
        _check(getattr(lib(), self._prefix + "_image")(self.h, C.c_int32(which), C.byref(v)), self._prefix + "_image")
        return view_to_tensor(v)

    def output(self, kind=OUTPUT_UPSAMPLE):
        v = hr_image_view()
        _check(getattr(lib(), self._prefix + "_output")(self.h, C.c_int(kind), C.byref(v)), self._prefix + "_output")
        return view_to_tensor(v)

    def history_apron_exceeded(self) -> bool:
        """row bands: a history tap fell on an image row this GPU does not hold since the last call (motion beyond history_halo)"""
        v = C.c_int32(0)
        _check(getattr(lib(), self._prefix + "_history_apron_exceeded")(self.h, C.byref(v)), self._prefix + "_history_apron_exceeded")
        return bool(v.value)

    def reset_history(self):
        _check(getattr(lib(), self._prefix + "_reset_history")(self.h), self._prefix + "_reset_history")

    def launch_order(self) -> np.ndarray:
        """the trace kernel's launch list (launch slot -> 8x8 tile) as its next launch will read it: always a permutation; the identity
        until the first sort has run (hr_shadows_launch_order / hr_ao_launch_order)"""
        fn = getattr(lib(), self._prefix + "_launch_order")
        n = C.c_int32(0)
        _check(fn(self.h, None, C.byref(n)), self._prefix + "_launch_order")
        out = np.zeros(n.value, np.uint32)
        if n.value:
            _check(fn(self.h, out.ctypes.data_as(C.POINTER(C.c_uint32)), None), self._prefix + "_launch_order")
        return out

    def close(self):
        if self.h:
            getattr(lib(), self._prefix + "_destroy")(self.h)
            self.h = C.c_void_p()


class RayTracedShadows(_Pass):
    """src/ray_traced_shadows.h:7-142.  ``render(scene, frame_inputs)`` = RayTracedShadows::render(cmd_buf)."""
    _prefix = "hr_shadows"
    IMG_MASK, IMG_TEMPORAL, IMG_MOMENTS0, IMG_MOMENTS1, IMG_PREV, IMG_ATROUS0, IMG_ATROUS1, IMG_UPSAMPLE, IMG_TILES, IMG_GEO = range(10)   # IMG_GEO: tolerance mode, the geometry records of the last temporal stage

    def __init__(self, ctx: Context, width: int, height: int, scale: int = SCALE_FULL_RES, band=None):
        self.ctx = ctx
        self.params = hr_shadows_params()
        lib().hr_shadows_default_params(C.byref(self.params))
        self.h = C.c_void_p()
        b = hr_band(*band) if band else None
        _check(lib().hr_shadows_create(ctx.h, C.c_int32(width), C.c_int32(height), C.c_int(scale), C.byref(b) if b else None, C.byref(self.h)), "hr_shadows_create")
        self.scale = scale
        self.width, self.height = width >> scale, height >> scale

    def render(self, scene: Scene, inputs: hr_frame_inputs, stream=None):
        _check(lib().hr_shadows_render(self.h, scene.h, C.byref(inputs), C.byref(self.params), _stream_ptr(stream)), "hr_shadows_render")

    # stage-level entry points (multi-GPU halo exchange happens between them)
    def ray_trace(self, scene, inputs, stream=None):
        _check(lib().hr_shadows_ray_trace(self.h, scene.h, C.byref(inputs), C.byref(self.params), _stream_ptr(stream)), "hr_shadows_ray_trace")

    def denoise(self, inputs, stream=None):
        """everything of render() after the trace (the fused launches in tolerance mode)"""
        _check(lib().hr_shadows_denoise(self.h, C.byref(inputs), C.byref(self.params), _stream_ptr(stream)), "hr_shadows_denoise")

    def temporal(self, inputs, stream=None):
        _check(lib().hr_shadows_temporal(self.h, C.byref(inputs), C.byref(self.params), _stream_ptr(stream)), "hr_shadows_temporal")

    def atrous_iteration(self, inputs, i, stream=None):
        _check(lib().hr_shadows_atrous_iteration(self.h, C.byref(inputs), C.byref(self.params), C.c_int32(i), _stream_ptr(stream)), "hr_shadows_atrous_iteration")

    def upsample(self, inputs, stream=None):
        _check(lib().hr_shadows_upsample(self.h, C.byref(inputs), C.byref(self.params), _stream_ptr(stream)), "hr_shadows_upsample")

    def ray_count(self) -> int:
        n = C.c_uint64(0)
        _check(lib().hr_shadows_ray_count(self.h, C.byref(n)), "hr_shadows_ray_count")
        return n.value

    def tile_ray_counts(self) -> np.ndarray:
        """[tiles_y, tiles_x] uint16: rays fired per 8x8 tile by the last ray_trace"""
        tx, ty = C.c_int32(0), C.c_int32(0)
        _check(lib().hr_shadows_tile_ray_counts(self.h, None, C.byref(tx), C.byref(ty)), "hr_shadows_tile_ray_counts")
        out = np.zeros((ty.value, tx.value), np.uint16)
        _check(lib().hr_shadows_tile_ray_counts(self.h, out.ctypes.data_as(C.POINTER(C.c_uint16)), None, None), "hr_shadows_tile_ray_counts")
        return out

    def trace_stats(self, scene, inputs, stream=None, timed=False):
        """(rays, nodes visited, triangles tested) from the instrumented trace kernel.  timed=False: the full walk (occluder cache bypassed);
        timed=True: the kernel render() launches in the pass's present state, cache on (hr_shadows_trace_stats_timed)"""
        out = (C.c_uint64 * 3)()
        fn = lib().hr_shadows_trace_stats_timed if timed else lib().hr_shadows_trace_stats
        _check(fn(self.h, scene.h, C.byref(inputs), C.byref(self.params), out, _stream_ptr(stream)), "hr_shadows_trace_stats")
        return int(out[0]), int(out[1]), int(out[2])


class RayTracedAO(_Pass):
    """src/ray_traced_ao.h:7-126.  ``render(scene, frame_inputs)`` = RayTracedAO::render(cmd_buf)."""
    _prefix = "hr_ao"
    IMG_MASK, IMG_AO0, IMG_AO1, IMG_LEN0, IMG_LEN1, IMG_BLUR0, IMG_BLUR1, IMG_UPSAMPLE, IMG_TILES = range(9)

    def __init__(self, ctx: Context, width: int, height: int, scale: int = SCALE_HALF_RES, band=None):
        self.ctx = ctx
        self.params = hr_ao_params()
        lib().hr_ao_default_params(C.byref(self.params))
        self.h = C.c_void_p()
        b = hr_band(*band) if band else None
        _check(lib().hr_ao_create(ctx.h, C.c_int32(width), C.c_int32(height), C.c_int(scale), C.byref(b) if b else None, C.byref(self.h)), "hr_ao_create")
        self.scale = scale
        self.width, self.height = width >> scale, height >> scale

    def render(self, scene: Scene, inputs: hr_frame_inputs, stream=None):
        _check(lib().hr_ao_render(self.h, scene.h, C.byref(inputs), C.byref(self.params), _stream_ptr(stream)), "hr_ao_render")

    def ray_trace(self, scene, inputs, stream=None):
        _check(lib().hr_ao_ray_trace(self.h, scene.h, C.byref(inputs), C.byref(self.params), _stream_ptr(stream)), "hr_ao_ray_trace")

    def denoise(self, inputs, stream=None):
        _check(lib().hr_ao_denoise(self.h, C.byref(inputs), C.byref(self.params), _stream_ptr(stream)), "hr_ao_denoise")

    def temporal(self, inputs, stream=None):
        _check(lib().hr_ao_temporal(self.h, C.byref(inputs), C.byref(self.params), _stream_ptr(stream)), "hr_ao_temporal")

    def blur(self, inputs, which, stream=None):
        _check(lib().hr_ao_blur(self.h, C.byref(inputs), C.byref(self.params), C.c_int32(which), _stream_ptr(stream)), "hr_ao_blur")

    def upsample(self, inputs, stream=None):
        _check(lib().hr_ao_upsample(self.h, C.byref(inputs), C.byref(self.params), _stream_ptr(stream)), "hr_ao_upsample")

    def ray_count(self) -> int:
        n = C.c_uint64(0)
        _check(lib().hr_ao_ray_count(self.h, C.byref(n)), "hr_ao_ray_count")
        return n.value

    def trace_stats(self, scene, inputs, stream=None):
        out = (C.c_uint64 * 3)()
        _check(lib().hr_ao_trace_stats(self.h, scene.h, C.byref(inputs), C.byref(self.params), out, _stream_ptr(stream)), "hr_ao_trace_stats")
        return int(out[0]), int(out[1]), int(out[2])


ABI_SYMBOLS += ["hr_ao_default_params", "hr_ao_create", "hr_ao_render", "hr_ao_output", "hr_ao_reset_history", "hr_ao_destroy", "hr_ao_ray_trace",
                "hr_ao_denoise", "hr_ao_temporal", "hr_ao_blur", "hr_ao_upsample", "hr_ao_image", "hr_ao_history_apron_exceeded", "hr_ao_set_profiling", "hr_ao_get_stage_times", "hr_ao_ray_count",
                "hr_ao_trace_stats", "hr_ao_launch_order"]
