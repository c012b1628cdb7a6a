"""Asset ingestion: OBJ/MTL and PNG readers against files written by the test itself."""
import os
import struct
import zlib

import numpy as np
import pytest

from hybrid_rendering_amd import assets


def _write_png(path, img, filters):
    """8-bit PNG encoder with a chosen scanline filter per row (cycled), to exercise every unfilter branch."""
    h, w, c = img.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[c]
    raw = bytearray()
    prev = np.zeros(w * c, np.int32)
    for y in range(h):
        line = img[y].reshape(-1).astype(np.int32)
        ft = filters[y % len(filters)]
        a = np.concatenate([np.zeros(c, np.int32), line[:-c]])
        b = prev
        cc = np.concatenate([np.zeros(c, np.int32), prev[:-c]])
        if ft == 0: pred = np.zeros_like(line)
        elif ft == 1: pred = a
        elif ft == 2: pred = b
        elif ft == 3: pred = (a + b) >> 1
        else:
            pa, pb, pc = np.abs(b - cc), np.abs(a - cc), np.abs(a + b - 2 * cc)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, cc))
        raw.append(ft)
        raw += bytes(((line - pred) & 255).astype(np.uint8))
        prev = line
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    z = zlib.compress(bytes(raw))
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) + chunk(b"IDAT", z[: len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b""))


@pytest.mark.parametrize("c", [1, 2, 3, 4])
def test_png_roundtrip_all_filters(tmp_path, c):
    rng = np.random.RandomState(c)
    img = rng.randint(0, 256, (37, 53, c)).astype(np.uint8)
    img[5:9] = 7                                               # flat rows (long zero runs after filtering)
    p = str(tmp_path / "t.png")
    _write_png(p, img, [0, 1, 2, 3, 4])
    assert np.array_equal(assets.load_png(p), img)


def test_png_rejects_corruption(tmp_path):
    p = str(tmp_path / "t.png")
    _write_png(p, np.zeros((4, 4, 3), np.uint8), [0])
    d = bytearray(open(p, "rb").read())
    d[40] ^= 0xff
    open(p, "wb").write(bytes(d))
    with pytest.raises(ValueError):
        assets.load_png(p)
    open(p, "wb").write(b"not a png at all")
    with pytest.raises(ValueError):
        assets.load_png(p)


def test_blue_noise_tables_from_png_drive_the_sampler(tmp_path, oracle):
    """tables written as PNGs and read back feed sample_blue_noise exactly like the in-memory ones"""
    from hybrid_rendering_amd import synth
    sob, sr = synth.blue_noise_tables()
    sob_img = np.zeros((1, 256, 4), np.uint8); sob_img[0] = sob
    _write_png(str(tmp_path / "sobol_256_4d.png"), sob_img, [1])
    _write_png(str(tmp_path / "scrambling_ranking_128x128_2d_1spp.png"), sr, [4, 2, 3])
    s2, r2 = assets.load_blue_noise(str(tmp_path / "sobol_256_4d.png"), str(tmp_path / "scrambling_ranking_128x128_2d_1spp.png"))
    assert np.array_equal(s2, sob) and np.array_equal(r2, sr) and s2.flags.c_contiguous and r2.flags.c_contiguous
    _write_png(str(tmp_path / "small.png"), np.zeros((8, 8, 4), np.uint8), [0])
    with pytest.raises(ValueError):
        assets.load_blue_noise(str(tmp_path / "small.png"), str(tmp_path / "small.png"))


OBJ = """# two quads and a triangle
mtllib scene.mtl
o floor
v 0 0 0
v 1 0 0
v 1 0 1
v 0 0 1
vn 0 1 0
usemtl red
f 1//1 2//1 3//1 4//1
o wall
v 0 0 0
v 0 1 0
v 1 1 0
usemtl shiny
f -3 -2 -1
g lid
v 0 2 0
v 1 2 0
v 1 2 1
v 0 2 1
vt 0 0
f 8/1 9/1 10/1 11/1
"""
MTL = """newmtl red
Kd 0.9 0.1 0.2
Ns 98
newmtl shiny
Kd 0.5 0.5 0.5
Pr 0.05
Pm 1.0
Ns 10
Ke 1 2 3
"""


def test_obj_mtl_loader(tmp_path):
    (tmp_path / "scene.obj").write_text(OBJ)
    (tmp_path / "scene.mtl").write_text(MTL)
    sd = assets.load_obj(str(tmp_path / "scene.obj"), scale=2.0)
    assert sd.n_tris == 5 and sd.verts.dtype == np.float32 and sd.verts.shape == (5, 3, 3)
    assert np.array_equal(sd.verts[0], np.array([[0, 0, 0], [2, 0, 0], [2, 0, 2]], np.float32))      # fan of the first quad, scaled
    assert np.array_equal(sd.verts[1], np.array([[0, 0, 0], [2, 0, 2], [0, 0, 2]], np.float32))
    assert np.array_equal(sd.verts[2], np.array([[0, 0, 0], [0, 2, 0], [2, 2, 0]], np.float32))      # negative indices
    assert np.all(sd.normals[:2] == np.array([0, 1, 0], np.float32))                                  # vn from the file
    assert np.allclose(sd.normals[2], [0, 0, -1])                                                     # generated face normal
    assert np.allclose(sd.normals[3], [0, -1, 0]) or np.allclose(sd.normals[3], [0, 1, 0])
    assert list(sd.tri_material) == [0, 0, 1, 1, 1] and sd.meta["materials"] == ["red", "shiny"]
    assert list(sd.tri_mesh_id) == [2, 2, 3, 4, 4]                                                    # o / g start new meshes
    red, shiny = sd.materials
    assert np.allclose(red[:3], [0.9, 0.1, 0.2]) and abs(red[4] - np.sqrt(2 / 100)) < 1e-6 and red[3] == 0
    assert shiny[4] == np.float32(0.05) and shiny[3] == 1.0 and np.allclose(shiny[5:], [1, 2, 3])      # Pr wins over Ns
    lo, hi = sd.bounds()
    assert np.array_equal(lo, [0, 0, 0]) and np.array_equal(hi, [2, 4, 2])


def test_obj_scene_builds_and_traces(tmp_path, oracle):
    """the loaded arrays go straight into the (oracle) scene: a ray down onto the floor quad hits it"""
    (tmp_path / "scene.obj").write_text(OBJ)
    (tmp_path / "scene.mtl").write_text(MTL)
    sd = assets.load_obj(str(tmp_path / "scene.obj"))
    sc = oracle.Scene(sd)
    rays = np.array([[0.5, 0.5, 0.5, 10.0, 0, -1, 0, 0.001], [0.5, 0.5, 0.5, 1.0, 0, 1, 0, 0.001]], np.float32)   # origin, t_max, direction, t_min
    hit = sc.any_hit(rays)
    assert list(np.asarray(hit).astype(int)) == [1, 0]


def _gltf_doc(tmp_path, embed):
    import base64
    import json
    # buffer: quad positions (4 x vec3 f32) | indices (6 x u16) | interleaved triangle: pos+normal per vertex (stride 24)
    quad = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32)
    idx = np.array([0, 1, 2, 0, 2, 3], np.uint16)
    inter = np.array([[0, 0, 0, 0, 0, 1], [2, 0, 0, 0, 0, 1], [0, 2, 0, 0, 0, 1]], np.float32)
    blob = quad.tobytes() + idx.tobytes() + inter.tobytes()
    uri = ("data:application/octet-stream;base64," + base64.b64encode(blob).decode()) if embed else "geo.bin"
    if not embed:
        (tmp_path / "geo.bin").write_bytes(blob)
    doc = {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}],
        "nodes": [{"children": [1, 2], "translation": [10, 0, 0]},
                  {"mesh": 0, "scale": [2, 2, 2]},
                  {"mesh": 1, "rotation": [0, 0, 0.70710678, 0.70710678], "translation": [0, 5, 0]}],   # 90 deg about z
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0}, "indices": 1, "material": 0}]},
                   {"primitives": [{"attributes": {"POSITION": 2, "NORMAL": 3}}]}],
        "materials": [{"pbrMetallicRoughness": {"baseColorFactor": [0.2, 0.4, 0.6, 1], "metallicFactor": 0.0, "roughnessFactor": 0.3}, "emissiveFactor": [1, 0, 0]}],
        "buffers": [{"byteLength": len(blob), "uri": uri}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 48}, {"buffer": 0, "byteOffset": 48, "byteLength": 12},
                        {"buffer": 0, "byteOffset": 60, "byteLength": 72, "byteStride": 24}],
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3"},
                      {"bufferView": 1, "componentType": 5123, "count": 6, "type": "SCALAR"},
                      {"bufferView": 2, "byteOffset": 0, "componentType": 5126, "count": 3, "type": "VEC3"},
                      {"bufferView": 2, "byteOffset": 12, "componentType": 5126, "count": 3, "type": "VEC3"}],
    }
    return doc, blob


@pytest.mark.parametrize("form", ["external", "embedded", "glb"])
def test_gltf_loader(tmp_path, form):
    import json
    doc, blob = _gltf_doc(tmp_path, embed=(form == "embedded"))
    if form == "glb":
        del doc["buffers"][0]["uri"]
        js = json.dumps(doc).encode()
        js += b" " * (-len(js) % 4)
        bn = blob + b"\0" * (-len(blob) % 4)
        data = struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(bn)) + struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(bn), 0x004E4942) + bn
        path = tmp_path / "scene.glb"
        path.write_bytes(data)
    else:
        path = tmp_path / "scene.gltf"
        path.write_text(json.dumps(doc))
    sd = assets.load_gltf(str(path))
    assert sd.n_tris == 3 and list(sd.tri_material) == [0, 0, 1] and list(sd.tri_mesh_id) == [1, 1, 2]
    # quad: scaled by 2 then translated by (10,0,0)
    assert np.allclose(sd.verts[0], [[10, 0, 0], [12, 0, 0], [12, 2, 0]]) and np.allclose(sd.verts[1], [[10, 0, 0], [12, 2, 0], [10, 2, 0]])
    assert np.allclose(sd.normals[:2], [0, 0, 1])                                   # generated face normal
    # interleaved triangle: rotated 90 deg about z ((x,y) -> (-y,x)), then +(0,5,0), then parent +(10,0,0)
    assert np.allclose(sd.verts[2], [[10, 5, 0], [10, 7, 0], [8, 5, 0]], atol=1e-5)
    assert np.allclose(sd.normals[2], [0, 0, 1], atol=1e-6)                         # NORMAL accessor through the byteStride
    assert np.allclose(sd.materials[0], [0.2, 0.4, 0.6, 0.0, 0.3, 1, 0, 0]) and sd.materials.shape == (2, 8)


def _png_bytes(img):
    """minimal 8-bit RGBA PNG encoder (filter 0) for the tests"""
    import zlib
    h, w, c = img.shape
    assert c == 4
    raw = b"".join(b"\0" + img[y].tobytes() for y in range(h))

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xffffffff)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")


def test_gltf_textures_reach_the_hit_shading(tmp_path, oracle):
    """TEXCOORD_0 + baseColorTexture / metallicRoughnessTexture PNGs (one as a data URI, one as a file) -> SceneData.uvs /
    material_textures / textures -> the oracle's (reference-pinned) fetch_albedo: a ground-truth frame changes colour"""
    import base64
    import json
    from oracle import pyoracle_post as opost
    from hybrid_rendering_amd import synth, synth_env
    doc, blob = _gltf_doc(tmp_path, embed=True)
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    blob2 = blob + uv.tobytes()
    doc["buffers"][0] = {"byteLength": len(blob2), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob2).decode()}
    doc["bufferViews"].append({"buffer": 0, "byteOffset": len(blob), "byteLength": 32})
    doc["accessors"].append({"bufferView": 3, "componentType": 5126, "count": 4, "type": "VEC2"})
    doc["meshes"][0]["primitives"][0]["attributes"]["TEXCOORD_0"] = 4
    red = np.zeros((4, 4, 4), np.uint8); red[..., 0] = 255; red[..., 3] = 255
    mr = np.zeros((2, 2, 4), np.uint8); mr[..., 1] = 200; mr[..., 2] = 0; mr[..., 3] = 255
    (tmp_path / "mr.png").write_bytes(_png_bytes(mr))
    doc["images"] = [{"uri": "data:image/png;base64," + base64.b64encode(_png_bytes(red)).decode()}, {"uri": "mr.png"}, {"uri": "photo.jpg"}]
    (tmp_path / "photo.jpg").write_bytes(b"\xff\xd8\xff\xe0 not decodable here")
    doc["textures"] = [{"source": 0}, {"source": 1}, {"source": 2}]
    doc["materials"][0]["pbrMetallicRoughness"]["baseColorTexture"] = {"index": 0}
    doc["materials"][0]["pbrMetallicRoughness"]["metallicRoughnessTexture"] = {"index": 1}
    doc["materials"][0]["normalTexture"] = {"index": 2}        # JPEG: skipped
    path = tmp_path / "tex.gltf"
    path.write_text(json.dumps(doc))
    sd = assets.load_gltf(str(path))
    assert sd.uvs.shape == (3, 3, 2) and np.allclose(sd.uvs[0], [[0, 0], [1, 0], [1, 1]]) and np.all(sd.uvs[2] == 0)
    assert len(sd.textures) == 2 and sd.textures[0].shape == (4, 4, 4) and np.array_equal(sd.textures[1], mr)
    assert sd.material_textures.tolist() == [[0, -1, 1, 1, 1, 2], [-1, -1, -1, -1, 1, 2]]
    # the textured quad (z = 0 plane, x 10..12, y 0..2) seen head-on under a light from the front
    cam = synth.Camera((11.0, 1.0, 6.0), (11.0, 1.0, 0.0), fov=40.0, aspect=1.0)
    light = synth.make_light(direction_to_light=(0.2, 0.3, 1.0), radius=0.0, intensity=3.0)
    ubo = synth.make_ubo(cam, None, light)
    sky = synth_env.sky_cubemap(8)
    plain = assets.load_gltf(str(path))
    plain.material_textures = None
    outs = []
    for s in (sd, plain):
        gt = opost.GroundTruthPass(32, 32)
        outs.append(oracle.f16(gt.render(oracle.Scene(s), ubo, sky))[12:20, 12:20, :3].mean((0, 1)))
    assert outs[0][0] > 4 * outs[0][2] and outs[0][0] > 4 * outs[0][1]            # red albedo texture
    assert not np.allclose(outs[0], outs[1])                                        # factor (0.2, 0.4, 0.6) without the textures


def test_gltf_instanced_loader_keeps_meshes_and_instances(tmp_path, oracle):
    """load_gltf_instanced: the reference's scene layout (scene_descriptor_set.glsl:5-34) — meshes in object space, stored once however many nodes
    reference them, one instance { model_matrix, mesh_idx } per node; its flatten() is the world-space scene load_gltf() returns, and the oracle
    answers ray queries on both alike"""
    import json
    doc, blob = _gltf_doc(tmp_path, embed=True)
    doc["nodes"].append({"mesh": 0, "translation": [0, 0, 3], "rotation": [0.38268343, 0, 0, 0.92387953]})   # the quad a second time, tilted 45 deg about x
    doc["nodes"][0]["children"].append(3)
    path = tmp_path / "inst.gltf"
    path.write_text(json.dumps(doc))
    isd = assets.load_gltf_instanced(str(path), scale=2.0)
    assert [m.n_tris for m in isd.meshes] == [2, 1] and [(k, i) for _, k, i in isd.instances] == [(0, 1), (1, 2), (0, 3)]
    assert np.allclose(isd.meshes[0].verts[0], [[0, 0, 0], [1, 0, 0], [1, 1, 0]])                      # object space, untouched
    flat, world = isd.flatten(), assets.load_gltf(str(path), scale=2.0)
    assert flat.n_tris == world.n_tris == 5
    order = [0, 1, 2, 3, 4]            # same traversal order of the node hierarchy
    assert np.allclose(flat.verts[order], world.verts, atol=1e-4) and list(flat.tri_material) == list(world.tri_material)
    fn = flat.normals / np.linalg.norm(flat.normals, axis=-1, keepdims=True)
    assert np.allclose(fn, world.normals, atol=1e-5)                                                   # rigid + uniform scale: mat3(model) n and the inverse transpose agree
    oi, of = oracle.InstancedScene(isd), oracle.Scene(world)
    rng = np.random.RandomState(2)
    rays = np.zeros((500, 8), np.float32)
    lo, hi = world.bounds()
    rays[:, :3] = rng.uniform(lo - 1, hi + 1, (500, 3)); rays[:, 2] = 30.0
    rays[:, 4:7] = [0, 0, -1]; rays[:, 3] = 100.0; rays[:, 7] = 0.001
    hit_i, hit_f = oi.any_hit(rays), of.any_hit(rays)
    assert hit_i.sum() > 5 and (hit_i != hit_f).mean() < 0.01          # (the two transforms round differently: a ray on an edge may differ)
