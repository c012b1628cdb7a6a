"""Synthetic inputs for the ray-trace + denoise hot path.

The reference's assets (Sponza OBJ, Heitz blue-noise PNGs) are not in its tree
(SURVEY.md §8c), so tests and bench use deterministic procedural stand-ins:

* ``cornell32()``     — the 32-triangle Cornell box of BASELINE.json configs[0]
* ``sponza_like()``   — a ~260k-triangle colonnaded atrium with the extents of the
                        reference's Sponza instance (common.cpp:528: scale 0.3)
* ``Camera`` / ``make_ubo`` — the 416-byte per-frame UBO of common.h:161-179,
                        filled the way main.cpp:937-972 does
* ``blue_noise_tables`` — Sobol-256x4 + 128x128 scrambling/ranking tiles with the
                        texture layout of blue_noise.cpp:5-19 / bnd_sampler.glsl

Pure numpy; no GPU, no oracle.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

# --------------------------------------------------------------------------- scene container


@dataclass
class SceneData:
    verts: np.ndarray          # [n,3,3] float32 world-space positions
    normals: np.ndarray        # [n,3,3] float32 vertex normals
    tri_material: np.ndarray   # [n] uint32
    tri_mesh_id: np.ndarray    # [n] uint32
    materials: np.ndarray      # [m,8] float32: albedo rgb, metallic, roughness, emissive rgb
    name: str = "scene"
    meta: dict = field(default_factory=dict)
    # textured materials (scene_descriptor_set.glsl:20-27): all optional
    uvs: np.ndarray = None                # [n,3,2] float32
    tangents: np.ndarray = None           # [n,3,3] float32 (normal-mapped materials only)
    material_textures: np.ndarray = None  # [m,6] int32: albedo, normal, roughness, metallic texture (-1 none), roughness channel, metallic channel
    textures: list = None                 # list of [h,w,4] uint8 (RGBA8 UNORM)

    @property
    def n_tris(self) -> int:
        return int(self.verts.shape[0])

    def bounds(self):
        v = self.verts.reshape(-1, 3)
        return v.min(0), v.max(0)


@dataclass
class InstancedSceneData:
    """A scene as the reference holds it (scene_descriptor_set.glsl:30-34): meshes in OBJECT space + instances {model_matrix, mesh_idx}.
    meshes: SceneData whose verts / normals / uvs / tangents are object space and whose tri_material indexes `materials` (their
    tri_mesh_id / materials members are unused).  instances: (matrix16 column-major float32, mesh_idx, mesh_id)."""
    meshes: list
    instances: list
    materials: np.ndarray
    name: str = "instanced"
    material_textures: np.ndarray = None
    textures: list = None

    def matrices(self) -> np.ndarray:
        return np.ascontiguousarray(np.stack([np.asarray(m, np.float32).reshape(16) for m, _, _ in self.instances]), np.float32)

    def layout(self):
        """per instance: first_tri, mesh_tri_base, mesh_id, n_tris (uint32 arrays) — instance i's triangles follow instance i - 1's"""
        base = np.cumsum([0] + [m.n_tris for m in self.meshes]).astype(np.uint32)
        n = np.array([self.meshes[k].n_tris for _, k, _ in self.instances], np.uint32)
        first = np.concatenate([[0], np.cumsum(n)[:-1]]).astype(np.uint32)
        return first, np.array([base[k] for _, k, _ in self.instances], np.uint32), np.array([i for _, _, i in self.instances], np.uint32), n

    def mesh_arrays(self):
        cat = lambda xs: None if any(x is None for x in xs) else np.ascontiguousarray(np.concatenate(xs), xs[0].dtype)
        return dict(positions=cat([np.asarray(m.verts, np.float32) for m in self.meshes]), normals=cat([m.normals for m in self.meshes]),
                    material=cat([np.asarray(m.tri_material, np.uint32) for m in self.meshes]), uvs=cat([m.uvs for m in self.meshes]),
                    tangents=cat([m.tangents for m in self.meshes]))

    def flatten(self, matrices=None) -> "SceneData":
        """World-space SceneData in the pinned arithmetic of transform_vertex (numpy float32, one rounding per operation:
        ((m0 x + m1 y) + m2 z) + m3) — the arithmetic of csrc/instances.hip k_instances_transform; normals = mat3(model) * n."""
        mats = self.matrices() if matrices is None else np.asarray(matrices, np.float32).reshape(-1, 16)
        V, Nn, M, I, U, T = [], [], [], [], [], []
        for (m0, k, mid), m in zip(self.instances, mats):
            me = self.meshes[k]
            p = np.asarray(me.verts, np.float32).reshape(-1, 3)
            w = np.stack([((m[r] * p[:, 0] + m[4 + r] * p[:, 1]) + m[8 + r] * p[:, 2]) + m[12 + r] * np.float32(1.0) for r in range(3)], 1)
            V.append(w.reshape(-1, 3, 3).astype(np.float32))
            q = np.asarray(me.normals, np.float32).reshape(-1, 3)
            Nn.append(np.stack([(m[r] * q[:, 0] + m[4 + r] * q[:, 1]) + m[8 + r] * q[:, 2] for r in range(3)], 1).reshape(-1, 3, 3).astype(np.float32))
            M.append(np.asarray(me.tri_material, np.uint32)); I.append(np.full(me.n_tris, mid, np.uint32))
            U.append(me.uvs); T.append(me.tangents)
        cat = lambda xs: None if any(x is None for x in xs) else np.ascontiguousarray(np.concatenate(xs))
        return SceneData(verts=cat(V), normals=cat(Nn), tri_material=cat(M), tri_mesh_id=cat(I), materials=np.asarray(self.materials, np.float32), name=self.name + "_flat",
                         uvs=cat(U), tangents=cat(T), material_textures=self.material_textures, textures=self.textures)


def model_matrix(translate=(0.0, 0.0, 0.0), axis=(0.0, 1.0, 0.0), angle=0.0, scale=1.0) -> np.ndarray:
    """column-major mat4 = T * R(axis, angle) * S (float32), the shape of Instance::model_matrix"""
    a = np.asarray(axis, np.float64)
    a = a / max(np.linalg.norm(a), 1e-30)
    x, y, z = a
    c, s_ = math.cos(angle), math.sin(angle)
    R = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s_, x * z * (1 - c) + y * s_],
                  [y * x * (1 - c) + z * s_, c + y * y * (1 - c), y * z * (1 - c) - x * s_],
                  [z * x * (1 - c) - y * s_, z * y * (1 - c) + x * s_, c + z * z * (1 - c)]])
    sc = np.broadcast_to(np.asarray(scale, np.float64), (3,))
    M = np.eye(4)
    M[:3, :3] = R * sc[None, :]
    M[:3, 3] = translate
    return np.ascontiguousarray(M.T.reshape(16), np.float32)   # column-major


def instanced_cornell(n_boxes: int = 5, seed: int = 3, frame: int = 0, textured: bool = False) -> InstancedSceneData:
    """The Cornell room as ONE mesh (identity instance) + a unit-cube mesh and a small pyramid mesh instanced `n_boxes` times with
    rotations, non-uniform scales and translations; `frame` moves every second instance (a per-frame TLAS update, main.cpp:74)."""
    room = _Builder()
    S = 100.0
    room.box((0, 0, 0), (S, S, S), 0, faces="yYz", inward=True)
    room.box((0, 0, 0), (S, S, S), 1, faces="x", inward=True)
    room.box((0, 0, 0), (S, S, S), 2, faces="X", inward=True)
    room.quad((35, S - 0.5, 35), (65, S - 0.5, 35), (65, S - 0.5, 65), (35, S - 0.5, 65), 4)
    mats = cornell32().materials
    cube = _Builder()
    cube.box((-0.5, -0.5, -0.5), (0.5, 0.5, 0.5), 0, faces="xXyYzZ")
    pyr = _Builder()
    apex, b0, b1, b2, b3 = (0, 1, 0), (-0.5, 0, -0.5), (0.5, 0, -0.5), (0.5, 0, 0.5), (-0.5, 0, 0.5)
    pyr.add(np.array([[b0, apex, b1], [b1, apex, b2], [b2, apex, b3], [b3, apex, b0], [b0, b1, b2], [b0, b2, b3]], np.float32), None, 3)
    meshes = [room.finish(mats, "room"), cube.finish(mats, "cube"), pyr.finish(mats, "pyramid")]
    if textured:
        # per-mesh (object-space) texture coordinates and tangents; the textures and the material -> texture table are the scene's
        meshes = [with_textures(m) for m in meshes]
        return InstancedSceneData(meshes=meshes, instances=instanced_cornell_instances(n_boxes, seed, frame), materials=mats, name="instanced_cornell_textured",
                                  material_textures=meshes[0].material_textures, textures=meshes[0].textures)
    return InstancedSceneData(meshes=meshes, instances=instanced_cornell_instances(n_boxes, seed, frame), materials=mats, name="instanced_cornell")


def instanced_cornell_instances(n_boxes: int = 5, seed: int = 3, frame: int = 0):
    rng = np.random.RandomState(seed)
    inst = [(model_matrix(), 0, 1)]
    for i in range(n_boxes):
        pos = np.array([rng.uniform(15, 85), rng.uniform(8, 40), rng.uniform(15, 85)])
        axis, ang = rng.uniform(-1, 1, 3), rng.uniform(0, 2 * math.pi)
        sc = rng.uniform(8, 28, 3)
        if i % 2 == 1:   # the moving ones
            pos = pos + np.array([1.7, 0.9, -1.3]) * frame
            ang = ang + 0.11 * frame
        inst.append((model_matrix(pos, axis, ang, sc), 1 + (i % 2), 2 + i))
    return inst


class _Builder:
    def __init__(self):
        self.v, self.n, self.mat, self.mid = [], [], [], []
        self.next_mesh = 1

    def add(self, verts, normals, material, mesh_id=None):
        verts = np.asarray(verts, np.float32).reshape(-1, 3, 3)
        if normals is None:
            e1 = verts[:, 1] - verts[:, 0]
            e2 = verts[:, 2] - verts[:, 0]
            fn = np.cross(e1, e2)
            fn /= np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-20)
            normals = np.repeat(fn[:, None, :], 3, axis=1)
        normals = np.asarray(normals, np.float32).reshape(-1, 3, 3)
        if mesh_id is None:
            mesh_id = self.next_mesh
            self.next_mesh += 1
        self.v.append(verts)
        self.n.append(normals)
        self.mat.append(np.full(len(verts), material, np.uint32))
        self.mid.append(np.full(len(verts), mesh_id, np.uint32))
        return mesh_id

    def grid(self, P, N, material, mesh_id=None, flip=False):
        """P,N: [nu,nv,3] grids -> 2 triangles per cell."""
        a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
        na, nb, nc, nd = N[:-1, :-1], N[1:, :-1], N[1:, 1:], N[:-1, 1:]
        if flip:
            t1, t2 = np.stack([a, c, b], -2), np.stack([a, d, c], -2)
            n1, n2 = np.stack([na, nc, nb], -2), np.stack([na, nd, nc], -2)
        else:
            t1, t2 = np.stack([a, b, c], -2), np.stack([a, c, d], -2)
            n1, n2 = np.stack([na, nb, nc], -2), np.stack([na, nc, nd], -2)
        V = np.concatenate([t1.reshape(-1, 3, 3), t2.reshape(-1, 3, 3)])
        Nn = np.concatenate([n1.reshape(-1, 3, 3), n2.reshape(-1, 3, 3)])
        return self.add(V, Nn, material, mesh_id)

    def quad(self, p0, p1, p2, p3, material, nu=1, nv=1, mesh_id=None):
        """Planar quad p0->p1 (u), p0->p3 (v), tessellated nu x nv; normal = (p1-p0) x (p3-p0)."""
        p0, p1, p2, p3 = (np.asarray(p, np.float32) for p in (p0, p1, p2, p3))
        u = np.linspace(0, 1, nu + 1, dtype=np.float32)[:, None, None]
        v = np.linspace(0, 1, nv + 1, dtype=np.float32)[None, :, None]
        P = (1 - u) * (1 - v) * p0 + u * (1 - v) * p1 + u * v * p2 + (1 - u) * v * p3
        n = np.cross(p1 - p0, p3 - p0)
        n = n / max(np.linalg.norm(n), 1e-20)
        N = np.broadcast_to(n.astype(np.float32), P.shape)
        return self.grid(P, N, material, mesh_id)

    def box(self, lo, hi, material, faces="xXyYzZ", mesh_id=None, inward=False):
        lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
        x0, y0, z0 = lo
        x1, y1, z1 = hi
        if mesh_id is None:
            mesh_id = self.next_mesh
            self.next_mesh += 1
        F = {
            "x": [(x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0)],
            "X": [(x1, y0, z0), (x1, y1, z0), (x1, y1, z1), (x1, y0, z1)],
            "y": [(x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1)],
            "Y": [(x0, y1, z0), (x0, y1, z1), (x1, y1, z1), (x1, y1, z0)],
            "z": [(x0, y0, z0), (x0, y1, z0), (x1, y1, z0), (x1, y0, z0)],
            "Z": [(x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)],
        }
        for f in faces:
            p = F[f]
            if inward:
                p = [p[0], p[3], p[2], p[1]]
            self.quad(p[0], p[1], p[2], p[3], material, mesh_id=mesh_id)
        return mesh_id

    def cylinder(self, base, radius, height, material, seg=32, stacks=8, mesh_id=None):
        th = np.linspace(0, 2 * math.pi, seg + 1, dtype=np.float32)[:, None]
        hh = np.linspace(0, height, stacks + 1, dtype=np.float32)[None, :]
        c, s = np.cos(th), np.sin(th)
        P = np.stack([base[0] + radius * c + 0 * hh, base[1] + hh + 0 * c, base[2] + radius * s + 0 * hh], -1)
        N = np.stack([c + 0 * hh, 0 * c + 0 * hh, s + 0 * hh], -1)
        return self.grid(P.astype(np.float32), N.astype(np.float32), material, mesh_id, flip=True)

    def sphere(self, center, radius, material, seg=32, rings=16, mesh_id=None, bump=0.0, seed=0):
        th = np.linspace(0, 2 * math.pi, seg + 1, dtype=np.float32)[:, None]
        ph = np.linspace(0, math.pi, rings + 1, dtype=np.float32)[None, :]
        d = np.stack([np.cos(th) * np.sin(ph), np.cos(ph) + 0 * th, np.sin(th) * np.sin(ph)], -1)
        r = radius
        if bump > 0:
            r = radius * (1 + bump * np.sin(5 * th + seed) * np.sin(7 * ph + 2 * seed))[..., None]
        P = np.asarray(center, np.float32) + r * d
        return self.grid(P.astype(np.float32), d.astype(np.float32), material, mesh_id, flip=False)

    def arch(self, c0, c1, y, R, r, material, arc_seg=24, ring_seg=12, mesh_id=None):
        """Half-torus arch spanning c0->c1 (xz points) with spring line at height y."""
        c0, c1 = np.asarray(c0, np.float32), np.asarray(c1, np.float32)
        mid = 0.5 * (c0 + c1)
        ax = (c1 - c0)
        L = np.linalg.norm(ax)
        ax = ax / L
        R = 0.5 * L if R is None else R
        a = np.linspace(0, math.pi, arc_seg + 1, dtype=np.float32)[:, None]
        b = np.linspace(0, 2 * math.pi, ring_seg + 1, dtype=np.float32)[None, :]
        # centre line in the plane spanned by ax (xz) and up
        cx = -np.cos(a) * R
        cy = np.sin(a) * R
        # radial dir (in-plane) and binormal (perp to plane, horizontal)
        rad = np.stack([-np.cos(a) * ax[0], np.sin(a), -np.cos(a) * ax[1]], -1)  # [arc,1,3]
        bn = np.array([-ax[1], 0.0, ax[0]], np.float32)
        nrm = np.cos(b)[..., None] * rad + np.sin(b)[..., None] * bn
        ctr = np.stack([mid[0] + cx * ax[0], y + cy, mid[1] + cx * ax[1]], -1)
        P = ctr + r * nrm
        return self.grid(P.astype(np.float32), nrm.astype(np.float32), material, mesh_id, flip=False)

    def curtain(self, p0, du, dv, material, nu=48, nv=48, amp=6.0, waves=5.0, mesh_id=None):
        p0, du, dv = (np.asarray(p, np.float32) for p in (p0, du, dv))
        u = np.linspace(0, 1, nu + 1, dtype=np.float32)[:, None]
        v = np.linspace(0, 1, nv + 1, dtype=np.float32)[None, :]
        n = np.cross(du, dv)
        n = n / np.linalg.norm(n)
        off = amp * np.sin(2 * math.pi * waves * u) * (0.3 + 0.7 * v)
        P = p0 + u[..., None] * du + v[..., None] * dv + off[..., None] * n
        dPu = du + (amp * 2 * math.pi * waves * np.cos(2 * math.pi * waves * u) * (0.3 + 0.7 * v))[..., None] * n
        dPv = dv + (amp * np.sin(2 * math.pi * waves * u) * 0.7 + 0 * v)[..., None] * n
        N = np.cross(dPu, dPv)
        N /= np.maximum(np.linalg.norm(N, axis=-1, keepdims=True), 1e-20)
        return self.grid(P.astype(np.float32), N.astype(np.float32), material, mesh_id)

    def cards(self, center, radius, count, size, material, rng, mesh_id=None):
        """`count` randomly oriented quads (two triangles each) scattered in a ball: foliage cards — thin, overlapping, incoherent"""
        c = np.asarray(center, np.float32)
        d = rng.normal(size=(count, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        pos = c + d * (radius * rng.uniform(0.2, 1.0, size=(count, 1)) ** (1 / 3)).astype(np.float32)
        a = rng.normal(size=(count, 3)).astype(np.float32)
        a /= np.linalg.norm(a, axis=1, keepdims=True)
        b = np.cross(a, rng.normal(size=(count, 3)).astype(np.float32))
        b /= np.maximum(np.linalg.norm(b, axis=1, keepdims=True), 1e-6)
        su = (size * rng.uniform(0.5, 1.5, size=(count, 1))).astype(np.float32)
        sv = (size * 0.45 * rng.uniform(0.5, 1.5, size=(count, 1))).astype(np.float32)
        p0, p1, p2, p3 = pos - a * su - b * sv, pos + a * su - b * sv, pos + a * su + b * sv, pos - a * su + b * sv
        V = np.concatenate([np.stack([p0, p1, p2], 1), np.stack([p0, p2, p3], 1)])
        return self.add(V, None, material, mesh_id)

    def finish(self, materials, name, **meta) -> SceneData:
        return SceneData(np.concatenate(self.v), np.concatenate(self.n), np.concatenate(self.mat), np.concatenate(self.mid),
                         np.asarray(materials, np.float32), name, dict(meta))


def with_textures(sd: SceneData, seed: int = 5) -> SceneData:
    """A textured variant of a scene: planar texture coordinates along the dominant axis of every triangle, Gram-Schmidt
    vertex tangents, three procedural RGBA8 textures (colour checker, bump normal map, roughness/metallic noise) and a
    material -> texture table that exercises every fetch_* branch (none / albedo / albedo+normal / all four)."""
    rng = np.random.RandomState(seed)
    v, n = sd.verts.astype(np.float32), sd.normals.astype(np.float32)
    fn = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    axis = np.argmax(np.abs(fn), axis=1)
    ua, va = (axis + 1) % 3, (axis + 2) % 3
    lo, hi = sd.bounds()
    scale = np.float32(8.0 / max(float((hi - lo).max()), 1e-6))
    idx = np.arange(len(v))
    uvs = np.stack([v[idx, :, ua] * scale + np.float32(0.13), v[idx, :, va] * scale - np.float32(0.29)], -1).astype(np.float32)   # [n,3,2]
    e = np.zeros((len(v), 3), np.float32)
    e[idx, ua] = 1.0
    t = e[:, None, :] - n * np.sum(n * e[:, None, :], -1, keepdims=True)
    t = (t / np.maximum(np.linalg.norm(t, axis=-1, keepdims=True), 1e-8)).astype(np.float32)
    yy, xx = np.mgrid[0:64, 0:64]
    alb = np.zeros((64, 64, 4), np.uint8)
    chk = ((xx // 8 + yy // 8) & 1).astype(bool)
    alb[..., 0] = np.where(chk, 220, 60) + rng.randint(-20, 20, (64, 64))
    alb[..., 1] = np.where(chk, 90, 200) + rng.randint(-20, 20, (64, 64))
    alb[..., 2] = 128 + (64 * np.sin(xx * 0.4)).astype(np.int32)
    alb[..., 3] = 255
    y2, x2 = np.mgrid[0:32, 0:32]
    hx, hy = 0.6 * np.cos(x2 * 0.7), 0.6 * np.sin(y2 * 0.9)
    nm = np.stack([-hx, -hy, np.ones_like(hx)], -1)
    nm /= np.linalg.norm(nm, axis=-1, keepdims=True)
    nmap = np.concatenate([np.clip((nm * 0.5 + 0.5) * 255.0 + 0.5, 0, 255), np.full((32, 32, 1), 255.0)], -1).astype(np.uint8)
    rm = rng.randint(0, 256, (16, 16, 4)).astype(np.uint8)
    rm[..., 1] = rng.randint(10, 240, (16, 16))      # roughness channel g
    m = len(sd.materials)
    mt = np.full((m, 6), -1, np.int32)
    mt[:, 4], mt[:, 5] = 1, 2
    for i in range(m):
        k = (3, 1, 2, 0)[i % 4]      # material 0 (usually the largest surfaces) gets all four textures
        if k >= 1:
            mt[i, 0] = 0
        if k >= 2:
            mt[i, 1] = 1
        if k == 3:
            mt[i, 2] = mt[i, 3] = 2
    return SceneData(sd.verts, sd.normals, sd.tri_material, sd.tri_mesh_id, sd.materials, sd.name + "_textured", dict(sd.meta),
                     uvs=np.ascontiguousarray(uvs), tangents=np.ascontiguousarray(t), material_textures=mt,
                     textures=[np.ascontiguousarray(alb), np.ascontiguousarray(nmap), np.ascontiguousarray(rm)])


# --------------------------------------------------------------------------- scenes

def cornell32() -> SceneData:
    """32 triangles: 5 walls x2, two boxes x (5 faces x2), ceiling light quad x2."""
    mats = [
        [0.73, 0.73, 0.73, 0.0, 0.8, 0, 0, 0],   # white
        [0.65, 0.05, 0.05, 0.0, 0.8, 0, 0, 0],   # red
        [0.12, 0.45, 0.15, 0.0, 0.8, 0, 0, 0],   # green
        [0.9, 0.9, 0.9, 1.0, 0.02, 0, 0, 0],     # mirror box
        [1.0, 1.0, 1.0, 0.0, 0.5, 10, 10, 10],   # light
    ]
    b = _Builder()
    S = 100.0
    b.box((0, 0, 0), (S, S, S), 0, faces="yYz", inward=True)       # floor, ceiling, back wall (6 tris)
    b.box((0, 0, 0), (S, S, S), 1, faces="x", inward=True)        # left  (2)
    b.box((0, 0, 0), (S, S, S), 2, faces="X", inward=True)        # right (2)
    b.box((15, 0, 15), (45, 60, 45), 0, faces="xXYzZ")            # tall box, no bottom (10)
    b.box((55, 0, 50), (85, 30, 80), 3, faces="xXYzZ")            # short box (10)
    b.quad((35, S - 0.5, 35), (65, S - 0.5, 35), (65, S - 0.5, 65), (35, S - 0.5, 65), 4)  # light (2), faces down
    sc = b.finish(mats, "cornell32")
    assert sc.n_tris == 32, sc.n_tris
    return sc


def sponza_like(detail: float = 1.0, seed: int = 1234, tier: str = "standard") -> SceneData:
    """Colonnaded two-storey atrium, open roof; ~262k triangles at detail=1.

    Extents ~1100 x 450 x 700 (x: length, y: up, z: width) like the reference's Sponza
    instance (common.cpp:528).  ``detail`` scales tessellation (0.25 -> ~20k tris).

    ``tier="hard"`` (the standard scene costs only ~6 BVH nodes per shadow ray): the same building plus what makes
    real Sponza deep to traverse — three layers of finely folded fabric per curtain bay, foliage (thousands of randomly oriented thin
    cards in plant-sized clusters on the floor and the galleries), hanging chains of thin links, denser clutter: ~2.5 M triangles at
    detail=1 (1.0 M at detail=0.62), thin overlapping geometry that no split plane separates."""
    rng = np.random.RandomState(seed)
    mats = [
        [0.70, 0.66, 0.58, 0.0, 0.85, 0, 0, 0],  # 0 stone wall
        [0.55, 0.52, 0.48, 0.0, 0.60, 0, 0, 0],  # 1 floor
        [0.75, 0.72, 0.65, 0.0, 0.45, 0, 0, 0],  # 2 column
        [0.60, 0.10, 0.10, 0.0, 0.90, 0, 0, 0],  # 3 red curtain
        [0.10, 0.25, 0.55, 0.0, 0.90, 0, 0, 0],  # 4 blue curtain
        [0.10, 0.45, 0.15, 0.0, 0.90, 0, 0, 0],  # 5 green curtain
        [0.90, 0.75, 0.30, 1.0, 0.03, 0, 0, 0],  # 6 polished brass (mirror regime)
        [0.80, 0.80, 0.85, 1.0, 0.25, 0, 0, 0],  # 7 brushed metal (GGX regime)
        [0.35, 0.25, 0.18, 0.0, 0.78, 0, 0, 0],  # 8 rough wood (DDGI regime)
        [0.85, 0.85, 0.85, 0.0, 0.04, 0, 0, 0],  # 9 polished marble (mirror regime)
        [0.45, 0.40, 0.35, 0.0, 0.35, 0, 0, 0],  # 10 vase ceramic
    ]
    d = max(detail, 0.05)

    def t(n, lo=2):
        return max(lo, int(round(n * d)))

    b = _Builder()
    X0, X1, Z0, Z1, H = -550.0, 550.0, -350.0, 350.0, 450.0
    # floor with polished centre strip, walls
    b.quad((X0, 0, Z0), (X0, 0, Z1), (X1, 0, Z1), (X1, 0, Z0), 1, t(64), t(96))
    b.quad((-400, 0.05, -60), (-400, 0.05, 60), (400, 0.05, 60), (400, 0.05, -60), 9, t(16), t(48))
    b.quad((X0, 0, Z0), (X1, 0, Z0), (X1, H, Z0), (X0, H, Z0), 0, t(64), t(24))    # z- wall (normal +z)
    b.quad((X1, 0, Z1), (X0, 0, Z1), (X0, H, Z1), (X1, H, Z1), 0, t(64), t(24))    # z+ wall
    b.quad((X0, 0, Z1), (X0, 0, Z0), (X0, H, Z0), (X0, H, Z1), 0, t(40), t(24))    # x- wall
    b.quad((X1, 0, Z0), (X1, 0, Z1), (X1, H, Z1), (X1, H, Z0), 0, t(40), t(24))    # x+ wall
    # upper gallery slabs along both long sides (between wall and colonnade)
    zc = 170.0
    for sgn in (-1, 1):
        za, zb = sorted((sgn * zc - sgn * 18, sgn * 350.0))
        b.box((X0, 200, za), (X1, 215, zb), 0)
        b.box((X0, 410, za), (X1, 425, zb), 0)  # partial roof over galleries (atrium centre open)
    # colonnades: two floors, two sides
    ncol = 12
    xs = np.linspace(-480, 480, ncol)
    for floor_y, col_h, rad in ((0.0, 150.0, 16.0), (215.0, 130.0, 12.0)):
        for sgn in (-1, 1):
            z = sgn * zc
            for i, x in enumerate(xs):
                b.box((x - rad * 1.4, floor_y, z - rad * 1.4), (x + rad * 1.4, floor_y + 10, z + rad * 1.4), 2)
                b.cylinder((x, floor_y + 10, z), rad, col_h - 20, 2, seg=t(40, 8), stacks=t(14, 2))
                b.box((x - rad * 1.5, floor_y + col_h - 10, z - rad * 1.5), (x + rad * 1.5, floor_y + col_h, z + rad * 1.5), 2)
            for i in range(ncol - 1):
                b.arch((xs[i], z), (xs[i + 1], z), floor_y + col_h, None, 9.0, 0, arc_seg=t(28, 6), ring_seg=t(14, 4))
    # curtains hanging between upper columns, alternating colours
    for sgn in (-1, 1):
        for i in range(0, ncol - 1):
            if (i + (sgn > 0)) % 2:
                continue
            m = 3 + (i // 2) % 3
            x0, x1 = xs[i] + 14, xs[i + 1] - 14
            z = sgn * (zc - 4)
            b.curtain((x0, 345, z), (x1 - x0, 0, 0), (0, -125, 0), m, nu=t(56, 6), nv=t(56, 6), amp=5.0, waves=4.0)
    # long banners across the atrium
    for x in (-300.0, 0.0, 300.0):
        b.curtain((x, 400, -120), (0, 0, 240), (0, -150, 0), 3 + int(x > -1) + int(x > 1), nu=t(64, 6), nv=t(48, 6), amp=7.0, waves=3.0)
    # vases / spheres on the floor (varied materials for the three reflection regimes)
    for k in range(16):
        x = -450 + 60 * k + rng.uniform(-10, 10)
        z = rng.choice([-90.0, 90.0]) + rng.uniform(-10, 10)
        r = rng.uniform(14, 24)
        b.sphere((x, r, z), r, [6, 7, 8, 10][k % 4], seg=t(40, 8), rings=t(20, 4), bump=0.08 * (k % 3), seed=k)
    # "lion heads": large bumpy spheres at the ends
    for x, m in ((-500.0, 7), (500.0, 6)):
        b.sphere((x, 120, 0), 60, m, seg=t(160, 12), rings=t(80, 6), bump=0.12, seed=int(abs(x)))
    # clutter boxes on the gallery
    for k in range(40):
        x = rng.uniform(-500, 500)
        z = rng.choice([-1.0, 1.0]) * rng.uniform(200, 330)
        s = rng.uniform(8, 25)
        b.box((x - s, 215, z - s), (x + s, 215 + 2 * s, z + s), 8 if k % 2 else 10)
    if tier == "hard":
        mats.append([0.12, 0.35, 0.10, 0.0, 0.70, 0, 0, 0])   # 11 leaf
        mats.append([0.25, 0.22, 0.20, 1.0, 0.40, 0, 0, 0])   # 12 iron
        # fabric: two more layers behind every curtain, finer folds, slightly different phase (overlapping thin sheets)
        for sgn in (-1, 1):
            for i in range(0, ncol - 1):
                x0, x1 = xs[i] + 14, xs[i + 1] - 14
                for layer in range(3):
                    z = sgn * (zc - 10 - 5 * layer)
                    b.curtain((x0, 345 - 2 * layer, z), (x1 - x0, 0, 0), (0, -125 - 60 * (layer == 2), 0), 3 + (i + layer) % 3, nu=t(120, 8), nv=t(96, 8),
                              amp=4.0 + 2.0 * layer, waves=6.0 + 2.5 * layer)
        for x in np.linspace(-420, 420, 8):
            for layer in range(2):
                b.curtain((x + 6 * layer, 405, -150), (0, 0, 300), (0, -170, 0), 3 + int(x > 0) + layer, nu=t(140, 8), nv=t(90, 8), amp=6.0 + 3 * layer, waves=5.0 + layer)
        # foliage: plant-sized clusters of thin cards along the floor edges and on the galleries
        for k in range(140):
            on_gallery = k % 3 == 0
            x = rng.uniform(-520, 520)
            z = rng.choice([-1.0, 1.0]) * (rng.uniform(200, 330) if on_gallery else rng.uniform(100, 150))
            y = (215.0 if on_gallery else 0.0) + rng.uniform(20, 45)
            b.cards((x, y, z), rng.uniform(18, 34), t(900, 40), rng.uniform(3.0, 6.0), 11, rng)
        # hanging chains: thin vertical links from the roof slabs
        for k in range(60):
            x, z = rng.uniform(-520, 520), rng.choice([-1.0, 1.0]) * rng.uniform(175, 340)
            for j in range(t(24, 4)):
                y = 405 - 7.5 * j
                b.cylinder((x + (j % 2) * 0.8, y - 6, z), 0.7, 6.0, 12, seg=t(10, 4), stacks=1)
        for k in range(160):
            x = rng.uniform(-500, 500)
            z = rng.choice([-1.0, 1.0]) * rng.uniform(200, 330)
            sz = rng.uniform(4, 14)
            b.box((x - sz, 215, z - sz), (x + sz, 215 + 2 * sz, z + sz), 8 if k % 2 else 10)
    return b.finish(mats, "sponza_like" + ("_hard" if tier == "hard" else ""), detail=detail, seed=seed, tier=tier)


def sponza_hard_light() -> np.ndarray:
    """a low sun (about 20 degrees above the horizon, along the atrium): grazing shadow rays that run the length of the building
    through the fabric and the foliage"""
    d = np.array([0.82, 0.36, 0.44])
    return make_light(LIGHT_DIRECTIONAL, direction_to_light=d / np.linalg.norm(d), radius=0.08, intensity=10.0)


# --------------------------------------------------------------------------- camera / UBO

def _normalize(v):
    v = np.asarray(v, np.float64)
    return v / np.linalg.norm(v)


def look_at(eye, target, up=(0, 1, 0)) -> np.ndarray:
    """Right-handed view matrix (glm::lookAt)."""
    eye = np.asarray(eye, np.float64)
    f = _normalize(np.asarray(target, np.float64) - eye)
    s = _normalize(np.cross(f, up))
    u = np.cross(s, f)
    M = np.eye(4)
    M[0, :3], M[1, :3], M[2, :3] = s, u, -f
    M[0, 3], M[1, 3], M[2, 3] = -s @ eye, -u @ eye, f @ eye
    return M


def perspective(fov_deg, aspect, near, far) -> np.ndarray:
    """Right-handed, depth 0..1 (Vulkan), y flipped so +y world is up on screen."""
    f = 1.0 / math.tan(math.radians(fov_deg) / 2)
    P = np.zeros((4, 4))
    P[0, 0] = f / aspect
    P[1, 1] = -f
    P[2, 2] = far / (near - far)
    P[2, 3] = -(far * near) / (far - near)
    P[3, 2] = -1.0
    return P


UBO_DTYPE = np.dtype([
    ("view_inverse", np.float32, 16), ("proj_inverse", np.float32, 16), ("view_proj_inverse", np.float32, 16),
    ("prev_view_proj", np.float32, 16), ("view_proj", np.float32, 16), ("cam_pos", np.float32, 4),
    ("current_prev_jitter", np.float32, 4), ("light", np.float32, 16),
])
assert UBO_DTYPE.itemsize == 416

LIGHT_DIRECTIONAL, LIGHT_POINT, LIGHT_SPOT = 0, 1, 2


def make_light(light_type=LIGHT_DIRECTIONAL, direction_to_light=(0.3, 0.9, 0.2), position=(0, 0, 0), radius=0.08,
               color=(1, 1, 1), intensity=10.0, cone_inner_deg=40.0, cone_outer_deg=50.0) -> np.ndarray:
    """Light packed as common.h:106-158 (data0.xyz = direction TO the light; main.cpp:963)."""
    d = _normalize(direction_to_light)
    L = np.zeros(16, np.float32)
    L[0:3], L[3] = d, intensity
    L[4:7], L[7] = position, radius
    L[8:11] = color
    L[12], L[13], L[14] = float(light_type), math.cos(math.radians(cone_outer_deg)), math.cos(math.radians(cone_inner_deg))
    return L


@dataclass
class Camera:
    eye: tuple
    target: tuple
    fov: float = 60.0
    near: float = 1.0      # CAMERA_NEAR_PLANE common.h:19
    far: float = 1000.0    # CAMERA_FAR_PLANE  common.h:20
    aspect: float = 16 / 9

    def view_proj(self):
        V = look_at(self.eye, self.target)
        P = perspective(self.fov, self.aspect, self.near, self.far)
        return V, P


def z_buffer_params(near=1.0, far=1000.0) -> np.ndarray:
    """main.cpp:253-254."""
    x = -1.0 + near / far
    return np.array([x, 1.0, x / near, 1.0 / near], np.float32)


def make_ubo(cam: Camera, prev_cam: Camera | None, light: np.ndarray, use_ao: float = 1.0) -> np.ndarray:
    """Fill the per-frame UBO the way main.cpp:951-966 does (no TAA jitter).  Column-major matrices."""
    V, P = cam.view_proj()
    VP = P @ V
    if prev_cam is None:
        prev_cam = cam
    pV, pP = prev_cam.view_proj()
    u = np.zeros((), UBO_DTYPE)

    def cm(M):
        return np.asarray(M, np.float64).T.reshape(16).astype(np.float32)

    u["view_inverse"] = cm(np.linalg.inv(V))
    u["proj_inverse"] = cm(np.linalg.inv(P))
    u["view_proj"] = cm(VP)
    u["view_proj_inverse"] = cm(np.linalg.inv(VP))
    u["prev_view_proj"] = cm(pP @ pV)
    u["cam_pos"] = np.array([*cam.eye, use_ao], np.float32)
    u["current_prev_jitter"] = 0
    u["light"] = light
    return u


def sponza_camera(aspect=16 / 9, frame: int = 0, dolly: float = 0.0) -> Camera:
    """Camera near the reference's Sponza preset position (main.cpp:1131) looking down the atrium."""
    eye = np.array([279.5372, 35.164913 + 40.0, -20.101242]) + np.array([-1.0, 0.0, 0.0]) * dolly * frame
    return Camera(tuple(eye), tuple(eye + np.array([-1.0, 0.12, 0.08])), aspect=aspect)


def camera_for_bounds(bounds, aspect=16 / 9, frame: int = 0, dolly: float = 0.0) -> Camera:
    """A camera for an arbitrary scene (bench.py --obj): inside the box at 80 % of its length and 35 % of its height, looking down the
    long horizontal axis as the Sponza preset does, dollying `dolly` scene-relative units per frame (1 unit = 1/1100 of the length)."""
    lo, hi = (np.asarray(b, np.float64) for b in bounds)
    ext = hi - lo
    ax = 0 if ext[0] >= ext[2] else 2
    fwd = np.zeros(3); fwd[ax] = -1.0
    eye = lo + ext * np.array([0.5, 0.35, 0.5])
    eye[ax] = lo[ax] + 0.8 * ext[ax]
    eye = eye + fwd * dolly * frame * (ext[ax] / 1100.0)
    return Camera(tuple(eye), tuple(eye + fwd + np.array([0.0, 0.12, 0.0]) + np.roll(np.array([0.0, 0.0, 0.08]), 0 if ax == 0 else 1)), aspect=aspect)


def sponza_light() -> np.ndarray:
    """Directional sun of the Sponza preset (main.cpp:869-874): radius 0.08, intensity 10,
    transform = rotZ(30deg) * rotX(-10deg); direction = mat3(T) * (0,-1,0); UBO stores -direction."""
    a, bx = math.radians(30.0), math.radians(-10.0)
    Rz = np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]])
    Rx = np.array([[1, 0, 0], [0, math.cos(bx), -math.sin(bx)], [0, math.sin(bx), math.cos(bx)]])
    d = (Rz @ Rx) @ np.array([0.0, -1.0, 0.0])
    return make_light(LIGHT_DIRECTIONAL, direction_to_light=-d, radius=0.08, intensity=10.0)


def cornell_camera(aspect=1.0) -> Camera:
    return Camera((50.0, 50.0, 235.0), (50.0, 50.0, 0.0), fov=40.0, aspect=aspect)


def cornell_light(hard=True) -> np.ndarray:
    """Point light just below the ceiling quad; radius 0 => hard shadows (lighting.glsl:44-47)."""
    return make_light(LIGHT_POINT, position=(50.0, 95.0, 50.0), radius=0.0 if hard else 3.0, intensity=5000.0)


# --------------------------------------------------------------------------- blue noise stand-in

def blue_noise_tables(seed: int = 7):
    """Deterministic stand-in for the Heitz-2019 1spp tables (blue_noise.cpp:5-19).

    Returns (sobol [256,4] uint8, scrambling_ranking [128,128,4] uint8).  The sobol rows are a
    (0,2)-sequence in base 2 (van der Corput / Sobol dim 2) for dims 0,1 and XOR-scrambled
    copies for dims 2,3; scrambling (rg) and ranking (b) tiles are white noise."""
    rng = np.random.RandomState(seed)
    i = np.arange(256, dtype=np.uint32)

    def vdc(n):
        r = np.zeros_like(n)
        for b in range(8):
            r |= ((n >> b) & 1) << (7 - b)
        return r

    def sobol2(n):
        r = np.zeros_like(n)
        v = np.uint32(1 << 7)
        nn = n.copy()
        for _ in range(8):
            r ^= np.where(nn & 1, v, 0).astype(np.uint32)
            nn >>= 1
            v = np.uint32(v ^ (v >> 1))
        return r

    s = np.zeros((256, 4), np.uint8)
    s[:, 0] = vdc(i)
    s[:, 1] = sobol2(i)
    s[:, 2] = vdc(i) ^ 0x5A
    s[:, 3] = sobol2(i) ^ 0xC3
    sr = rng.randint(0, 256, size=(128, 128, 4)).astype(np.uint8)
    sr[..., 3] = 255
    return s, sr
