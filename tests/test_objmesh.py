"""obj_mesh geometry (parser/JsonParser.cpp:529-754): the OBJ / MTL reader against
hand-checked expectations, and the per-vertex colours only an OBJ file can carry
(trimesh.cpp:179-189) through the oracle.  The reference holds no .obj file and no test for
this path, so these expectations come from reading its loader: parity unpinned."""
import json
import math
import os

import numpy as np
import pytest

from path_tracer_amd import abi, objfile
from path_tracer_amd.scene import SceneFormatError, parse_json_scene

CUBE_OBJ = """# a unit quad and a triangle in two groups
mtllib things.mtl
v 0 0 0 1 0 0
v 1 0 0 0 1 0
v 1 1 0 0 0 1
v 0 1 0 1 1 1
v 0.35 0.5 1 0.5 0.5 0.5
vn 0 0 2
vt 0 0
vt 1 0
vt 1 1
vt 0 1
g quad
usemtl second
f 1 2 3 4
o tri
f -1 1 2
"""

THINGS_MTL = """newmtl first
Kd 0.25 0.5 0.75
Ka 0.1 0.1 0.1
Tf 0 0 0
Ni 1.5
newmtl second
Kd 1 0 0
"""

SCENE = [
    {"camera": {"position": [0.5, 0.5, 5], "viewdir": [0, 0, -1], "updir": [0, 1, 0]}},
    {"point_light": {"position": [0.5, 0.5, 4], "color": [1, 1, 1], "constant_attenuation_coeff": 1.0,
                     "linear_attenuation_coeff": 0.0, "quadratic_attenuation_coeff": 0.0}},
    {"material": {"diffuse": {"constant": [0.9, 0.9, 0.9]}}},
    {"translate": [[0, 0, -1], [{"obj_mesh": {"objfile": "things.obj"}}]]},
]


def write_scene(tmp_path, obj=CUBE_OBJ, mtl=THINGS_MTL):
    (tmp_path / "things.obj").write_text(obj)
    if mtl is not None:
        (tmp_path / "things.mtl").write_text(mtl)
    return parse_json_scene(json.dumps(SCENE), str(tmp_path))


def test_numbers_are_assembled_like_the_reference_loader():
    # 3 * 0.1 + 5 * 0.01, not the correctly rounded 0.35 (tiny_obj_loader.h:994-1020)
    ok, v = objfile.parse_double("0.35")
    assert ok and v == 3 * 0.1 + 5 * 0.01 and v != 0.35
    assert objfile.parse_double("-.5") == (True, -0.5)
    assert objfile.parse_double("12") == (True, 12.0)
    assert objfile.parse_double("1e2") == (True, math.ldexp(1.0 * 5.0 ** 2, 2))
    assert objfile.parse_double("2.5E-1")[1] == math.ldexp(2.5 * math.pow(5.0, -1), -1)
    assert objfile.parse_double("7abc") == (True, 7.0)  # trailing characters are ignored
    for bad in ("", "abc", "e5", "1e", "+", "nan"):
        assert objfile.parse_double(bad)[0] is False, bad


def test_obj_shapes_corners_and_materials(tmp_path):
    sc = write_scene(tmp_path)
    # two shapes -> two meshes, each with the FIRST material of the library, not the scene's
    assert len(sc.meshes) == 2 and len(sc.objects) == 2
    for o in sc.objects:
        assert o.type == abi.OBJ_TRIMESH
        m = sc.materials[o.material]
        assert np.allclose(m.kd, [0.25, 0.5, 0.75]) and np.allclose(m.ka, 0.1) and m.index == 1.5
        assert m.roughness == 1.0 and m.metallic == 0.0  # Material() defaults, material.h:152-158
        assert o.xform[2, 3] == -1.0  # the transform stays on the object
    quad, tri = sc.meshes
    # both diagonals of the unit square are equally long: "shorter 0-2" is false -> (0,1,3), (1,2,3)
    # in OBJ numbering; mesh vertices are numbered in order of first use: 0, 1, 3, 2
    assert quad.faces.tolist() == [[0, 1, 2], [1, 3, 2]]
    assert quad.vertices.tolist() == [[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]]  # order of first use
    assert quad.normals is None and quad.uvs is None
    assert quad.colors.tolist() == [[1, 0, 0], [0, 1, 0], [1, 1, 1], [0, 0, 1]]
    # -1 is the last vertex read so far; 0.35 as the reference's reader assembles it
    assert tri.vertices[0].tolist() == [3 * 0.1 + 5 * 0.01, 0.5, 1.0]
    assert tri.faces.tolist() == [[0, 1, 2]]


def test_corner_forms_normals_and_uvs(tmp_path):
    obj = """v 0 0 0
v 1 0 0
v 0 1 0
vn 0 0 2
vt 0.25 0.75
f 1/1/1 2/1/1 3/1/1
f 1//1 2//1 3//1
"""
    with pytest.raises(SceneFormatError, match="Wrong number of UV"):
        write_scene(tmp_path, obj, None)  # the second face's corners have no vt: doubleCheck fails
    obj = obj.replace("f 1//1 2//1 3//1\n", "f 3/1/1 2/1/1 1/1/1\n")
    sc = write_scene(tmp_path, obj, None)
    (m,) = sc.meshes
    assert m.faces.tolist() == [[0, 1, 2], [2, 1, 0]]  # same (v, vt, vn) corners are shared
    assert np.array_equal(m.normals, [[0, 0, 1]] * 3)  # normalised
    assert np.array_equal(m.uvs, [[2 * 0.1 + 5 * 0.01, 7 * 0.1 + 5 * 0.01]] * 3)  # digit by digit
    assert m.colors is None
    mat = sc.materials[sc.objects[0].material]
    assert np.all(mat.kd == 0) and mat.index == 1.0  # no library: a default Material


def test_colours_need_every_vertex_and_quads_take_the_shorter_diagonal(tmp_path):
    obj = """v 0 0 0 1 0 0
v 2 0 0
v 2 1 0 0 0 1
v -1 1 0 0 0 1
f 1 2 3 4
"""
    sc = write_scene(tmp_path, obj, None)
    (m,) = sc.meshes
    assert m.colors is None
    # |v2 - v0|^2 = 5 < |v3 - v1|^2 = 10 -> (0,1,2), (0,2,3)
    assert m.faces.tolist() == [[0, 1, 2], [0, 2, 3]]
    with pytest.raises(SceneFormatError, match="out of range"):
        write_scene(tmp_path, obj + "v 0 2 0\nf 1 2 3 4 9\n", None)
    with pytest.raises(SceneFormatError, match="zero value"):
        write_scene(tmp_path, obj + "f 0 1 2\n", None)


def test_polygons_are_ear_clipped_like_the_reference_loader(tmp_path):
    # an L-shaped hexagon, counter-clockwise.  Worked by hand through tiny_obj_loader.h:1886-2132:
    # corner (0,1,2) fails the sign test (cross 2, "area" (1*1 - 1*3)/2 = -1), (1,2,3) is the
    # first ear, then (3,4,5), (3,5,0), and (0,1,3) is what remains.
    obj = "v 1 1 0\nv 3 1 0\nv 3 2 0\nv 2 2 0\nv 2 3 0\nv 1 3 0\nf 1 2 3 4 5 6\n"
    sc = write_scene(tmp_path, obj, None)
    (m,) = sc.meshes
    order = [1, 2, 3, 4, 5, 0]  # mesh vertex k is OBJ vertex order[k] (numbered by first use)
    tris = [[order[i] for i in f] for f in m.faces.tolist()]
    assert tris == [[1, 2, 3], [3, 4, 5], [3, 5, 0], [0, 1, 3]]
    # a convex pentagon becomes a fan around its first corner
    pts = [(math.cos(2 * math.pi * k / 5) + 2, math.sin(2 * math.pi * k / 5) + 2, 0.0) for k in range(5)]
    clipped = objfile.ear_clip([(k, -1, -1) for k in range(5)], pts)
    assert [c[0] for c in clipped] == [0, 1, 2, 0, 2, 3, 0, 3, 4]


def test_texture_names_and_gennormals(tmp_path):
    from PIL import Image

    Image.fromarray(np.full((2, 2, 3), 200, dtype=np.uint8)).save(tmp_path / "my tex.png")
    mtl = "newmtl a\nmap_Kd -s 1 1 1 -clamp on my tex.png\n"
    obj = "mtllib things.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nf 1/1 2/1 3/1\n"
    (tmp_path / "things.obj").write_text(obj)
    (tmp_path / "things.mtl").write_text(mtl)
    scene = list(SCENE)
    scene[3] = {"obj_mesh": {"objfile": "things.obj", "gennormals": True}}
    sc = parse_json_scene(json.dumps(scene), str(tmp_path))
    assert len(sc.textures) == 1 and sc.materials[sc.objects[0].material].kd_texture == 1
    assert np.allclose(sc.meshes[0].normals, [[0, 0, 1]] * 3)


def test_vertex_colours_replace_the_diffuse_colour(tmp_path, oracle_lib):
    sc = write_scene(tmp_path)
    orc = oracle_lib.Oracle(sc)
    # straight down onto the quad, clear of the triangle shape above it.  (0.1, 0.8) lies in
    # (v0, v1, v3) = red, green, white with weights 0.1, 0.1, 0.8
    org = np.array([[0.1, 0.8, 3.0], [0.9, 0.8, 3.0]])
    d = np.array([[0.0, 0.0, -1.0]] * 2)
    uv, kd, ka = orc.hit_material(org, d)
    assert np.allclose(kd[0], 0.1 * np.array([1, 0, 0]) + 0.1 * np.array([0, 1, 0]) + 0.8 * np.array([1, 1, 1]))
    # (0.9, 0.8) lies in (v1, v2, v3) = green, blue, white with weights 0.2, 0.7, 0.1
    assert np.allclose(kd[1], 0.2 * np.array([0, 1, 0]) + 0.7 * np.array([0, 0, 1]) + 0.1 * np.array([1, 1, 1]))
    assert np.allclose(ka, 0.1)
    # a mesh with UVs never reads its colours (trimesh.cpp:170-178)
    sc.meshes[0].uvs = np.zeros((4, 2))
    uv, kd, ka = oracle_lib.Oracle(sc).hit_material(org, d)
    assert np.allclose(kd, [0.25, 0.5, 0.75])
    # and the colours show in a render
    sc.meshes[0].uvs = None
    p = abi.default_params(32, 32, spp=4)
    img = orc.render(p, threads=4)
    sc.meshes[0].colors = None
    plain = oracle_lib.Oracle(sc).render(p, threads=4)
    assert img.max() > 0.05 and np.abs(img - plain).max() > 0.02


# ------------------------------------------------------------------- GPU
def coloured_mesh_scene():
    from path_tracer_amd import synth

    sc = synth.make_scene(20000, seed=9)
    rng = np.random.default_rng(5)
    for m in sc.meshes:
        assert m.uvs is None
        m.colors = rng.uniform(0.05, 0.95, size=(len(m.vertices), 3))
    return sc


@pytest.mark.gpu
@pytest.mark.parametrize("builder", ["host", "gpu"])
def test_gpu_vertex_colours_match_the_oracle(builder, gpu, oracle_lib, monkeypatch):
    from tests import util as U

    if builder == "gpu":
        monkeypatch.setenv("MIPT_BUILDER", "gpu")
    sc = coloured_mesh_scene()
    orc = oracle_lib.Oracle(sc)
    gpu.upload_scene(sc)
    org, d = U.camera_rays(sc, 30000, seed=8)
    uo, kdo, kao = orc.hit_material(org, d)
    ug, kdg, kag = gpu.hit_material(org, d)
    plain = np.array([m.kd for m in sc.materials])
    on_mesh = (np.abs(kdo[:, None, :] - plain[None, :, :]).max(axis=2) > 1e-12).all(axis=1) & (np.abs(kdo).sum(axis=1) > 0)
    assert on_mesh.mean() > 0.2  # the colours, not a material constant, on a good part of the hits
    assert np.abs(kdg - kdo).max() <= 1e-9 and np.abs(kag - kao).max() <= 1e-12
    for spp, flags in ((1, abi.FLAG_FIXED_AREA_PATTERN), (4, 0)):
        p = abi.default_params(128, 72, spp=spp, seed=3, flags=flags)
        f32, rgb8 = gpu.render(p)
        ref = orc.render(p, threads=8)
        diff = np.abs(f32.astype(np.float64) - ref).max(axis=2)
        assert (diff <= 1e-5).mean() >= 0.995, f"{(diff > 1e-5).sum()} pixels differ (max {diff.max():.3e})"


@pytest.mark.gpu
def test_gpu_obj_scene_renders_like_the_oracle(tmp_path, gpu, oracle_lib):
    sc = write_scene(tmp_path)
    orc = oracle_lib.Oracle(sc)
    gpu.upload_scene(sc)
    p = abi.default_params(96, 96, spp=4, seed=2)
    f32, _ = gpu.render(p)
    ref = orc.render(p, threads=4)
    assert ref.max() > 0.05
    assert (np.abs(f32.astype(np.float64) - ref).max(axis=2) <= 1e-5).mean() >= 0.998
