"""Asset ingestion: Wavefront OBJ/MTL and glTF 2.0 (.gltf / .glb) -> the triangle / normal / material arrays hr_scene_desc takes
(the layout of scene_descriptor_set.glsl:5-34 after instance flattening), and PNG -> the Heitz-2019 blue-noise tables
(blue_noise.cpp:5-19: ``sobol_256_4d.png`` row 0 and a 128x128 ``scrambling_ranking_128x128_2d_*spp.png``).

The reference loads meshes through assimp and images through stb_image (external/dwSampleFramework, absent here); these
are self-contained readers for the subset those assets use: triangulated or polygonal ``f`` records with v / v/vt / v//vn /
v/vt/vn indices (negative indices allowed), ``usemtl`` + ``mtllib`` (Kd, Pr / Ns, Pm, Ke), ``o`` / ``g`` groups -> mesh ids;
8-bit non-interlaced PNG of colour type 0, 2, 4 or 6 with all five scanline filters.
"""
from __future__ import annotations

import os
import struct
import zlib

import numpy as np

from .synth import SceneData


# ------------------------------------------------------------------------------------------------ OBJ / MTL
def load_mtl(path: str) -> dict:
    """name -> [albedo rgb, metallic, roughness, emissive rgb] (float32[8]).  Pr/Pm are the PBR extension; without Pr the
    Phong exponent maps to roughness = sqrt(2 / (Ns + 2)) (the usual Blinn-Phong -> GGX conversion)."""
    mats, cur, has_pr = {}, None, set()
    with open(path, "r", errors="replace") as f:
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            k = t[0].lower()
            if k == "newmtl":
                name = " ".join(t[1:])
                cur = np.array([0.8, 0.8, 0.8, 0.0, 0.5, 0.0, 0.0, 0.0], np.float32)
                mats[name] = cur
            elif cur is None:
                continue
            elif k == "kd":
                cur[0:3] = [float(x) for x in t[1:4]]
            elif k == "pr":
                cur[4] = float(t[1])
                has_pr.add(name)
            elif k == "pm":
                cur[3] = float(t[1])
            elif k == "ns" and name not in has_pr:
                cur[4] = float(np.sqrt(2.0 / (float(t[1]) + 2.0)))
            elif k == "ke":
                cur[5:8] = [float(x) for x in t[1:4]]
    return mats


def load_obj(path: str, scale: float = 1.0, default_material=(0.8, 0.8, 0.8, 0.0, 0.5, 0.0, 0.0, 0.0)) -> SceneData:
    """Triangles in file order; polygons are fanned; missing normals become the geometric face normal (what assimp's
    aiProcess_GenNormals yields for flat faces)."""
    pos, nrm = [], []
    tri_v, tri_n, tri_mat, tri_mesh = [], [], [], []
    mat_names, mat_index, mtl = [], {}, {}
    cur_mat, cur_mesh, n_mesh = None, 1, 1
    base = os.path.dirname(os.path.abspath(path))

    def mat_id(name):
        if name not in mat_index:
            mat_index[name] = len(mat_names)
            mat_names.append(name)
        return mat_index[name]

    with open(path, "r", errors="replace") as f:
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            k = t[0]
            if k == "v":
                pos.append([float(t[1]) * scale, float(t[2]) * scale, float(t[3]) * scale])
            elif k == "vn":
                nrm.append([float(t[1]), float(t[2]), float(t[3])])
            elif k == "f":
                idx = []
                for c in t[1:]:
                    p = c.split("/")
                    vi = int(p[0])
                    ni = int(p[2]) if len(p) > 2 and p[2] else 0
                    idx.append((vi - 1 if vi > 0 else len(pos) + vi, (ni - 1 if ni > 0 else len(nrm) + ni) if ni else -1))
                for i in range(1, len(idx) - 1):
                    tri = (idx[0], idx[i], idx[i + 1])
                    tri_v.append([t_[0] for t_ in tri])
                    tri_n.append([t_[1] for t_ in tri])
                    tri_mat.append(mat_id(cur_mat))
                    tri_mesh.append(cur_mesh)
            elif k == "usemtl":
                cur_mat = " ".join(t[1:])
            elif k == "mtllib":
                mp = os.path.join(base, " ".join(t[1:]))
                if os.path.exists(mp):
                    mtl.update(load_mtl(mp))
            elif k in ("o", "g"):
                n_mesh += 1
                cur_mesh = n_mesh
    if not tri_v:
        raise ValueError(f"{path}: no faces")
    P = np.asarray(pos, np.float32)
    verts = P[np.asarray(tri_v, np.int64)]                                  # [n,3,3]
    fn = np.cross(verts[:, 1] - verts[:, 0], verts[:, 2] - verts[:, 0])
    fn = fn / np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-20)
    normals = np.repeat(fn[:, None, :], 3, axis=1).astype(np.float32)
    ni = np.asarray(tri_n, np.int64)
    if len(nrm):
        N = np.asarray(nrm, np.float32)
        has = ni >= 0
        normals[has] = N[ni[has]]
    materials = np.stack([np.asarray(mtl.get(n, default_material), np.float32) for n in mat_names]) if mat_names else np.asarray([default_material], np.float32)
    return SceneData(verts=np.ascontiguousarray(verts), normals=np.ascontiguousarray(normals), tri_material=np.asarray(tri_mat, np.uint32),
                     tri_mesh_id=np.asarray(tri_mesh, np.uint32), materials=np.ascontiguousarray(materials), name=os.path.basename(path),
                     meta=dict(materials=mat_names))


# ------------------------------------------------------------------------------------------------ PNG
_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def load_png(path: str) -> np.ndarray:
    """[H, W, C] uint8 (C = 1, 2, 3 or 4).  8-bit, non-interlaced."""
    return decode_png(open(path, "rb").read(), path)


def decode_png(data: bytes, path: str = "<memory>") -> np.ndarray:
    if data[:8] != _PNG_SIG:
        raise ValueError(f"{path}: not a PNG")
    o, idat, ihdr = 8, [], None
    while o < len(data):
        n, typ = struct.unpack(">I4s", data[o:o + 8])
        body = data[o + 8:o + 8 + n]
        if zlib.crc32(typ + body) & 0xffffffff != struct.unpack(">I", data[o + 8 + n:o + 12 + n])[0]:
            raise ValueError(f"{path}: CRC mismatch in {typ!r}")
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
        o += 12 + n
    w, h, depth, ctype, _, _, interlace = ihdr
    if depth != 8 or interlace != 0 or ctype not in (0, 2, 4, 6):
        raise ValueError(f"{path}: only 8-bit non-interlaced grey / RGB / grey+alpha / RGBA PNGs are supported")
    c = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8)
    stride = w * c
    rows = raw.reshape(h, stride + 1)
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        ft, line = int(rows[y, 0]), rows[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft in (1, 3, 4):
            cur = np.zeros(stride, np.int32)
            for i in range(stride):                     # left-dependent filters: sequential along the scanline
                a = cur[i - c] if i >= c else 0
                b = prev[i]
                if ft == 1:
                    p = a
                elif ft == 3:
                    p = (a + b) >> 1
                else:
                    cc = prev[i - c] if i >= c else 0
                    pa, pb, pc = abs(b - cc), abs(a - cc), abs(a + b - 2 * cc)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else cc)
                cur[i] = (line[i] + p) & 255
        else:
            raise ValueError(f"{path}: bad filter type {ft}")
        out[y] = cur
        prev = cur
    return out.reshape(h, w, c)


def _rgba(img: np.ndarray) -> np.ndarray:
    h, w, c = img.shape
    if c == 4:
        return img
    out = np.full((h, w, 4), 255, np.uint8)
    if c == 1:
        out[..., :3] = img
    elif c == 2:
        out[..., :3] = img[..., :1]; out[..., 3] = img[..., 1]
    else:
        out[..., :3] = img
    return out


def load_blue_noise(sobol_png: str, scrambling_ranking_png: str):
    """(sobol [256,4] uint8, scrambling_ranking [128,128,4] uint8) as hr_frame_inputs takes them: texelFetch(sobol,
    ivec2(i, 0)) reads row 0 only (bnd_sampler.glsl:16), the scrambling/ranking tile is 128x128 RGBA."""
    s, r = _rgba(load_png(sobol_png)), _rgba(load_png(scrambling_ranking_png))
    if s.shape[1] < 256 or r.shape[0] < 128 or r.shape[1] < 128:
        raise ValueError("blue-noise tables must be at least 256 wide (sobol) and 128x128 (scrambling/ranking)")
    return np.ascontiguousarray(s[0, :256]), np.ascontiguousarray(r[:128, :128])


# ------------------------------------------------------------------------------------------------ glTF 2.0
def _quat_to_mat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], np.float64)


def load_gltf(path: str, scale: float = 1.0, instanced: bool = False):
    """instanced=True: see load_gltf_instanced.  glTF 2.0 (.gltf with external / base64 buffers, or .glb) -> SceneData: the scene's node hierarchy is flattened to world
    space (matrix or TRS nodes), every TRIANGLES primitive contributes its (indexed) triangles, NORMAL is transformed by the
    inverse transpose (missing: geometric normal), materials come from pbrMetallicRoughness (baseColorFactor, metallicFactor,
    roughnessFactor) + emissiveFactor.  One mesh id per (node, primitive).  The reference loads meshes/*.gltf through assimp
    (common.cpp:347-488).  TEXCOORD_0, TANGENT and the PNG images behind baseColorTexture / normalTexture /
    metallicRoughnessTexture (roughness = G, metallic = B) become SceneData.uvs / tangents / material_textures / textures
    (the hit shaders' fetch_* inputs); images in other encodings (JPEG) are skipped, the factor then applies."""
    import base64
    import json
    raw = open(path, "rb").read()
    base = os.path.dirname(os.path.abspath(path))
    glb_bin = None
    if raw[:4] == b"glTF":
        _, _, total = struct.unpack("<4sII", raw[:12])
        o, doc = 12, None
        while o < total:
            n, typ = struct.unpack("<II", raw[o:o + 8])
            body = raw[o + 8:o + 8 + n]
            if typ == 0x4E4F534A:
                doc = json.loads(body.decode("utf-8"))
            elif typ == 0x004E4942 and glb_bin is None:
                glb_bin = bytes(body)
            o += 8 + n
    else:
        doc = json.loads(raw.decode("utf-8"))
    buffers = []
    for b in doc.get("buffers", []):
        uri = b.get("uri")
        if uri is None:
            buffers.append(glb_bin)
        elif uri.startswith("data:"):
            buffers.append(base64.b64decode(uri.split(",", 1)[1]))
        else:
            buffers.append(open(os.path.join(base, uri), "rb").read())
    comp = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
    ncomp = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}

    def accessor(i):
        a = doc["accessors"][i]
        dt, nc, cnt = np.dtype(comp[a["componentType"]]), ncomp[a["type"]], a["count"]
        if "bufferView" not in a:
            return np.zeros((cnt, nc), dt)
        bv = doc["bufferViews"][a["bufferView"]]
        off = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        stride = bv.get("byteStride", 0) or dt.itemsize * nc
        buf = np.frombuffer(buffers[bv["buffer"]], np.uint8)
        rows = np.lib.stride_tricks.as_strided(buf[off:], shape=(cnt, dt.itemsize * nc), strides=(stride, 1))
        return np.ascontiguousarray(rows).view(dt).reshape(cnt, nc)

    textures, image_slot = [], {}

    def texture_slot(tex_info):
        """glTF textureInfo -> index into `textures` (-1: absent or not a PNG)"""
        if not tex_info:
            return -1
        src = doc["textures"][tex_info["index"]].get("source")
        if src is None:
            return -1
        if src not in image_slot:
            img, data = doc["images"][src], None
            if "uri" in img:
                data = base64.b64decode(img["uri"].split(",", 1)[1]) if img["uri"].startswith("data:") else open(os.path.join(base, img["uri"]), "rb").read()
            elif "bufferView" in img:
                bv = doc["bufferViews"][img["bufferView"]]
                data = bytes(buffers[bv["buffer"]][bv.get("byteOffset", 0):bv.get("byteOffset", 0) + bv["byteLength"]])
            if data is not None and data[:8] == _PNG_SIG:
                image_slot[src] = len(textures)
                textures.append(np.ascontiguousarray(_rgba(decode_png(data))))
            else:
                image_slot[src] = -1
        return image_slot[src]

    mats, mat_tex = [], []
    for m in doc.get("materials", []):
        pbr = m.get("pbrMetallicRoughness", {})
        bc = pbr.get("baseColorFactor", [1, 1, 1, 1])
        em = m.get("emissiveFactor", [0, 0, 0])
        mats.append([bc[0], bc[1], bc[2], pbr.get("metallicFactor", 1.0), pbr.get("roughnessFactor", 1.0), em[0], em[1], em[2]])
        bct, nrm = texture_slot(pbr.get("baseColorTexture")), texture_slot(m.get("normalTexture"))
        mr = texture_slot(pbr.get("metallicRoughnessTexture"))
        mat_tex.append([bct, nrm, mr, mr, 1, 2])
    default_mat = len(mats)
    mats.append([0.8, 0.8, 0.8, 0.0, 0.5, 0.0, 0.0, 0.0])
    mat_tex.append([-1, -1, -1, -1, 1, 2])

    verts, norms, tmat, tmesh, uvs, tans = [], [], [], [], [], []
    mesh_id = [0]
    node_instances = []   # instanced: (column-major world matrix, glTF mesh index) per node that carries a mesh

    def visit(ni, parent):
        node = doc["nodes"][ni]
        if "matrix" in node:
            local = np.asarray(node["matrix"], np.float64).reshape(4, 4).T
        else:
            local = np.eye(4)
            local[:3, :3] = _quat_to_mat(node.get("rotation", [0, 0, 0, 1])) @ np.diag(node.get("scale", [1, 1, 1]))
            local[:3, 3] = node.get("translation", [0, 0, 0])
        M = parent @ local
        if instanced:
            if "mesh" in node:
                Ms = M.copy()
                Ms[:3, :] *= scale          # world = scale * (M * p)
                node_instances.append((np.ascontiguousarray(Ms.T.reshape(16), np.float32), int(node["mesh"])))
            for c in node.get("children", []):
                visit(c, M)
            return
        if "mesh" in node:
            nm = np.linalg.inv(M[:3, :3]).T
            for prim in doc["meshes"][node["mesh"]]["primitives"]:
                if prim.get("mode", 4) != 4:
                    continue
                pos = accessor(prim["attributes"]["POSITION"]).astype(np.float64)
                idx = accessor(prim["indices"]).reshape(-1).astype(np.int64) if "indices" in prim else np.arange(len(pos))
                idx = idx[: len(idx) // 3 * 3].reshape(-1, 3)
                wp = (pos @ M[:3, :3].T + M[:3, 3]) * scale
                tri = wp[idx]
                if "NORMAL" in prim["attributes"]:
                    n = accessor(prim["attributes"]["NORMAL"]).astype(np.float64) @ nm.T
                    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-20)
                    tn = n[idx]
                else:
                    fn = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
                    fn /= np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-20)
                    tn = np.repeat(fn[:, None, :], 3, axis=1)
                uv = accessor(prim["attributes"]["TEXCOORD_0"]).astype(np.float32)[idx] if "TEXCOORD_0" in prim["attributes"] else np.zeros((len(tri), 3, 2), np.float32)
                if "TANGENT" in prim["attributes"]:
                    tg = accessor(prim["attributes"]["TANGENT"]).astype(np.float64)[:, :3] @ M[:3, :3].T
                    tg /= np.maximum(np.linalg.norm(tg, axis=1, keepdims=True), 1e-20)
                    tg = tg[idx]
                else:
                    tg = np.zeros_like(tn)
                    tg[..., 0] = 1.0
                uvs.append(uv); tans.append(tg)
                mesh_id[0] += 1
                verts.append(tri); norms.append(tn)
                tmat.append(np.full(len(tri), prim.get("material", default_mat), np.uint32))
                tmesh.append(np.full(len(tri), mesh_id[0], np.uint32))
        for c in node.get("children", []):
            visit(c, M)

    scenes = doc.get("scenes") or [{"nodes": list(range(len(doc.get("nodes", []))))}]
    for root in scenes[doc.get("scene", 0)]["nodes"]:
        visit(root, np.eye(4))
    if instanced:
        from .synth import InstancedSceneData
        # one mesh per glTF mesh (all its TRIANGLES primitives, OBJECT space): scene_descriptor_set.glsl's per-mesh vertex / index / submesh buffers
        used = sorted({k for _, k in node_instances})
        if not used:
            raise ValueError(f"{path}: no node carries a mesh")
        meshes, slot = [], {}
        for k in used:
            mv, mn, mm, mu, mt_ = [], [], [], [], []
            for prim in doc["meshes"][k]["primitives"]:
                if prim.get("mode", 4) != 4:
                    continue
                pos = accessor(prim["attributes"]["POSITION"]).astype(np.float32)
                idx = accessor(prim["indices"]).reshape(-1).astype(np.int64) if "indices" in prim else np.arange(len(pos))
                idx = idx[: len(idx) // 3 * 3].reshape(-1, 3)
                tri = pos[idx]
                if "NORMAL" in prim["attributes"]:
                    tn = accessor(prim["attributes"]["NORMAL"]).astype(np.float32)[idx]
                else:
                    fn = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]).astype(np.float64)
                    fn /= np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-20)
                    tn = np.repeat(fn[:, None, :], 3, axis=1).astype(np.float32)
                mu.append(accessor(prim["attributes"]["TEXCOORD_0"]).astype(np.float32)[idx] if "TEXCOORD_0" in prim["attributes"] else np.zeros((len(tri), 3, 2), np.float32))
                if "TANGENT" in prim["attributes"]:
                    mt_.append(accessor(prim["attributes"]["TANGENT"]).astype(np.float32)[:, :3][idx])
                else:
                    tg = np.zeros_like(tn)
                    tg[..., 0] = 1.0
                    mt_.append(tg)
                mv.append(tri); mn.append(tn)
                mm.append(np.full(len(tri), prim.get("material", default_mat), np.uint32))
            if not mv:
                continue
            slot[k] = len(meshes)
            meshes.append(SceneData(verts=np.ascontiguousarray(np.concatenate(mv), np.float32), normals=np.ascontiguousarray(np.concatenate(mn), np.float32),
                                    tri_material=np.concatenate(mm), tri_mesh_id=np.zeros(sum(len(v) for v in mv), np.uint32), materials=np.asarray(mats, np.float32),
                                    name=doc["meshes"][k].get("name", f"mesh{k}"),
                                    uvs=np.ascontiguousarray(np.concatenate(mu), np.float32) if textures else None,
                                    tangents=np.ascontiguousarray(np.concatenate(mt_), np.float32) if textures else None))
        inst = [(m, slot[k], i + 1) for i, (m, k) in enumerate(node_instances) if k in slot]
        if not inst:
            raise ValueError(f"{path}: no triangle primitives")
        return InstancedSceneData(meshes=meshes, instances=inst, materials=np.asarray(mats, np.float32), name=os.path.basename(path),
                                  material_textures=np.asarray(mat_tex, np.int32) if textures else None, textures=textures if textures else None)
    if not verts:
        raise ValueError(f"{path}: no triangle primitives")
    sd = SceneData(verts=np.ascontiguousarray(np.concatenate(verts), np.float32), normals=np.ascontiguousarray(np.concatenate(norms), np.float32),
                   tri_material=np.concatenate(tmat), tri_mesh_id=np.concatenate(tmesh), materials=np.asarray(mats, np.float32),
                   name=os.path.basename(path), meta=dict(n_materials=len(mats) - 1, n_textures=len(textures)))
    if textures:
        sd.uvs = np.ascontiguousarray(np.concatenate(uvs), np.float32)
        sd.tangents = np.ascontiguousarray(np.concatenate(tans), np.float32)
        sd.material_textures = np.asarray(mat_tex, np.int32)
        sd.textures = textures
    return sd


def load_gltf_instanced(path: str, scale: float = 1.0):
    """glTF 2.0 -> synth.InstancedSceneData, the layout the reference's scene has on the GPU (scene_descriptor_set.glsl:5-34): one mesh per glTF
    mesh in OBJECT space (positions, normals, texture coordinates, tangents, per-triangle material), one instance { model_matrix, mesh_idx } per
    node that carries a mesh (world matrix of the node hierarchy, times `scale`; mesh id = 1 + the node's order of appearance).  A mesh referenced
    by several nodes is stored once.  hr.InstancedScene(ctx, it) builds it; its flatten() gives the world-space triangles load_gltf() returns
    (up to the rounding of the pinned fp32 transform; normals through mat3(model), as transform_vertex does, not the inverse transpose)."""
    return load_gltf(path, scale, instanced=True)
