"""The C++ host (host/): its scene loader builds the same flat description as the Python
loader, and the `ray` command line renders through the C ABI."""
import json
import math
import os
import subprocess

import numpy as np
import pytest

from path_tracer_amd import abi
from path_tracer_amd.scene import load_json_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "host")

SCENE = [
    {"camera": {"position": [0.0, 2.0, 9.0], "viewdir": [0.0, -0.2, -1.0], "updir": [0.0, 1.0, 0.0], "fov": 40.0, "aspectratio": 2.0}},
    {"ambient_light": {"color": [0.2, 0.2, 0.2]}},
    {"material": {"diffuse": {"constant": [0.3, 0.4, 0.5]}, "roughness": 0.7, "index": 1.3}},
    {"rectangle_area_light": {"position": [0.0, 6.0, -1.0], "color": [1.0, 0.9, 0.8], "u": [1.0, 0.0, 0.0], "v": [0.0, 0.0, 1.0], "uL": 2.0, "vL": 2.0,
                              "constant_attenuation_coeff": 1.0, "quadratic_attenuation_coeff": 0.0}},
    {"point_light": {"position": [3.0, 4.0, 2.0], "color": [0.5, 0.5, 0.5], "constant_attenuation_coeff": 0.5, "linear_attenuation_coeff": 0.05}},
    {"directional_light": {"direction": [-1.0, -2.0, -0.5], "color": [0.4, 0.4, 0.4]}},
    {"translate": [[-1.5, 0.0, 0.0], [{"rotate": [[0.0, 1.0, 1.0], 0.6, [{"scale": [[1.0, 1.5, 1.0], [
        {"sphere": {"material": {"ambient": {"constant": [0.1, 0.1, 0.1]}, "metallic": 0.5, "roughness": 0.0}}}]]}]]}]]},
    {"transform": [[1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 1.5, 0.2, -1.0, 1], [{"box": {}}]]},
    {"translate": [[0.0, -1.0, 0.0], [{"scale": [[30.0, 30.0, 30.0], [{"rotate": [[1.0, 0.0, 0.0], -1.5707963267948966, [
        {"square": {"material": {"diffuse": {"constant": [0.8, 0.8, 0.8]}}}}]]}]]}]]},
    {"tri_mesh": {"points": [[-1, 0, -3], [1, 0, -3], [1, 2, -3], [-1, 2, -3]], "faces": [[0, 1, 2, 3]], "gennormals": True,
                  "material": {"diffuse": {"constant": [0.9, 0.2, 0.2]}}}},
]


@pytest.fixture(scope="module")
def host_bins():
    subprocess.check_call(["make", "-C", HOST, "dump_scene"], stdout=subprocess.DEVNULL)
    return HOST


@pytest.fixture()
def scene_file(tmp_path):
    p = tmp_path / "scene.json"
    p.write_text(json.dumps(SCENE))
    return str(p)


def parse_dump(text):
    out = {"objects": [], "meshes": [], "lights": []}
    for line in text.splitlines():
        tok = line.split()
        if tok[0] == "camera":
            out["camera"] = np.array(tok[1:], float)
        elif tok[0] == "ambient":
            out["ambient"] = np.array(tok[1:], float)
        elif tok[0] == "object":
            out["objects"].append((int(tok[1]), int(tok[2]), np.array(tok[3:19], float), np.array(tok[20:32], float)))
        elif tok[0] == "mesh":
            nv, nf, hn = int(tok[1]), int(tok[2]), int(tok[3])
            v = np.array(tok[4:4 + 3 * nv], float)
            f = np.array(tok[4 + 3 * nv:4 + 3 * nv + 3 * nf], int)
            n = np.array(tok[4 + 3 * nv + 3 * nf:], float) if hn else None
            out["meshes"].append((v, f, n))
        elif tok[0] == "mattex":
            out.setdefault("mattex", []).append((int(tok[1]), int(tok[2])))
        elif tok[0] == "uv":
            out.setdefault("uvs", []).append(np.array(tok[1:], float))
        elif tok[0] == "color":
            out.setdefault("colors", []).append(np.array(tok[1:], float))
        elif tok[0] == "texture":
            out.setdefault("textures", []).append([int(x) for x in tok[1:]])
        elif tok[0] == "light":
            out["lights"].append((int(tok[1]), np.array(tok[2:], float)))
    return out


def test_cpp_loader_matches_python_loader(host_bins, scene_file):
    d = parse_dump(subprocess.check_output([os.path.join(host_bins, "dump_scene"), scene_file]).decode())
    sc = load_json_scene(scene_file)
    cam = np.concatenate([sc.camera.eye, sc.camera.look, sc.camera.u, sc.camera.v])
    assert np.allclose(d["camera"], cam, rtol=1e-15, atol=1e-15)
    assert np.allclose(d["ambient"], sc.ambient)
    assert len(d["objects"]) == len(sc.objects) == 4
    for (t, mesh, xf, mat), o in zip(d["objects"], sc.objects):
        assert (t, mesh) == (o.type, o.mesh)
        assert np.allclose(xf, np.asarray(o.xform).T.reshape(16), rtol=1e-14, atol=1e-15)
        m = sc.materials[o.material]
        assert np.allclose(mat, np.concatenate([m.kd, m.ka, m.kt, [m.roughness, m.metallic, m.index]]))
    (v, f, n), pm = d["meshes"][0], sc.meshes[0]
    assert np.allclose(v, pm.vertices.ravel()) and np.array_equal(f, pm.faces.ravel()) and np.allclose(n, pm.normals.ravel())
    assert [t for t, _ in d["lights"]] == [L.type for L in sc.lights]
    for (_, vals), L in zip(d["lights"], sc.lights):
        want = np.concatenate([L.color, L.position, L.direction, L.u, L.v, [L.uL, L.vL], L.atten.astype(float)])
        assert np.allclose(vals, want, rtol=1e-7)


def test_cpp_loader_reads_uvs_and_ppm_textures(host_bins, tmp_path):
    # a mapped diffuse colour on a mesh with UV coordinates; the C++ host reads binary PPM textures
    tex = np.random.RandomState(5).randint(0, 256, size=(4, 6, 3)).astype(np.uint8)  # rows top-down in the file
    with open(tmp_path / "tex.ppm", "wb") as fh:
        fh.write(b"P6\n6 4\n255\n" + tex.tobytes())
    scene = [
        {"camera": {"position": [0.0, 0.0, 4.0], "viewdir": [0.0, 0.0, -1.0], "updir": [0.0, 1.0, 0.0]}},
        {"material": {"ambient": {"mapped": "tex.ppm"}}},
        {"tri_mesh": {"points": [[0, 0, 0], [1, 0, 0], [1, 1, 0]], "faces": [[0, 1, 2]], "UV": [[0, 1], [1, 1], [1, 0]],
                      "material": {"diffuse": {"mapped": "tex.ppm"}, "normal": {"mapped": "tex.ppm"}}}},
        {"square": {"material": {"ambient": {"constant": [0.1, 0.2, 0.3]}}}},
    ]
    path = tmp_path / "scene.json"
    path.write_text(json.dumps(scene))
    d = parse_dump(subprocess.check_output([os.path.join(host_bins, "dump_scene"), str(path)]).decode())
    sc = load_json_scene(str(path))
    assert d["mattex"] == [(m.kd_texture, m.ka_texture) for m in sc.materials] == [(1, 1), (0, 0)]
    assert np.allclose(d["uvs"][0], sc.meshes[0].uvs.ravel())
    assert len(d["textures"]) == len(sc.textures) == 1
    w, h, _, r, g, b = d["textures"][0]
    assert (w, h) == (6, 4) and [r, g, b] == list(sc.textures[0][0, 0]) == list(tex[-1, 0])  # row 0 = bottom row of the file
    # other formats are refused by this host, with the reference's wording
    scene[1] = {"material": {"ambient": {"mapped": "tex.png"}}}
    path.write_text(json.dumps(scene))
    r = subprocess.run([os.path.join(host_bins, "dump_scene"), str(path)], capture_output=True)
    assert r.returncode == 2 and b"Unable to load texture map" in r.stderr


def test_cpp_loader_reads_obj_meshes_like_the_python_loader(host_bins, tmp_path):
    from tests.test_objmesh import CUBE_OBJ

    tex = np.random.RandomState(7).randint(0, 256, size=(2, 3, 3)).astype(np.uint8)
    with open(tmp_path / "wood grain.ppm", "wb") as fh:
        fh.write(b"P6\n3 2\n255\n" + tex.tobytes())
    (tmp_path / "things.obj").write_text(CUBE_OBJ)
    (tmp_path / "things.mtl").write_text("newmtl first\nKd 0.35 0.5 0.75\nKa 0.1 0.1 0.1\nTf 0.2 0 0\nNi 1.5\nmap_Kd -clamp on wood grain.ppm\nnewmtl second\nKd 1 0 0\n")
    # a second file with normals and texture coordinates, smooth normals regenerated on top
    (tmp_path / "fan.obj").write_text("v 0 0 0\nv 1 0 0.125\nv 1 1 0\nv 0 1 0.35\nv 1 1 1\nv 3 1 1\nv 3 2 1\nv 2 2 1\nv 2 3 1\nv 1 3 1\n"
                                        "vn 0 0 2\nvt 0.25 0.75\nvt 1 0\nf 1/1/1 2/2/1 3/1/1 4/2/1\nf 5/1/1 6/1/1 7/1/1 8/1/1 9/1/1 10/1/1\n")
    scene = [
        {"camera": {"position": [0.5, 0.5, 5], "viewdir": [0, 0, -1], "updir": [0, 1, 0]}},
        {"material": {"diffuse": {"constant": [0.9, 0.9, 0.9]}}},
        {"translate": [[0, 0, -1], [{"obj_mesh": {"objfile": "things.obj"}}]]},
        {"obj_mesh": {"objfile": "fan.obj", "gennormals": True}},
    ]
    path = tmp_path / "scene.json"
    path.write_text(json.dumps(scene))
    d = parse_dump(subprocess.check_output([os.path.join(host_bins, "dump_scene"), str(path)]).decode())
    sc = load_json_scene(str(path))
    assert len(d["meshes"]) == len(sc.meshes) == 3 and len(d["objects"]) == 3
    for (v, f, n), pm in zip(d["meshes"], sc.meshes):
        assert np.array_equal(v, pm.vertices.ravel()) and np.array_equal(f, pm.faces.ravel())  # the same doubles, bit for bit
        assert (n is None) == (pm.normals is None)
        if n is not None:
            assert np.allclose(n, pm.normals.ravel(), rtol=1e-15, atol=0)
    assert len(d["colors"]) == 2 and all(np.array_equal(c, pm.colors.ravel()) for c, pm in zip(d["colors"], sc.meshes[:2]))
    assert sc.meshes[2].colors is None and len(d["uvs"]) == 1 and np.array_equal(d["uvs"][0], sc.meshes[2].uvs.ravel())
    for (t, mesh, xf, mat), o in zip(d["objects"], sc.objects):
        m = sc.materials[o.material]
        assert np.array_equal(mat, np.concatenate([m.kd, m.ka, m.kt, [m.roughness, m.metallic, m.index]]))
    assert d["mattex"] == [(m.kd_texture, m.ka_texture) for m in (sc.materials[o.material] for o in sc.objects)] == [(1, 0), (1, 0), (0, 0)]
    assert len(d["textures"]) == 1 and d["textures"][0][:2] == [3, 2]
    # errors carry the reference's prefix
    (tmp_path / "fan.obj").write_text("v 0 0 0\nf 0 1 1\n")
    r = subprocess.run([os.path.join(host_bins, "dump_scene"), str(path)], capture_output=True)
    assert r.returncode == 2 and b"Error while parsing OBJ file" in r.stderr


def test_cpp_loader_reports_errors(host_bins, tmp_path):
    exe = os.path.join(host_bins, "dump_scene")
    bad = tmp_path / "bad.json"
    bad.write_text('[{"teapot": {}}]')
    r = subprocess.run([exe, str(bad)], capture_output=True)
    assert r.returncode == 2 and b"Unknown scene object type" in r.stderr
    bad.write_text('[{"sphere": {}')
    assert subprocess.run([exe, str(bad)], capture_output=True).returncode == 2
    assert subprocess.run([exe, str(tmp_path / "missing.json")], capture_output=True).returncode == 2
    bad.write_text("[" * 200000)  # nesting is bounded: an error, not a stack overflow
    r = subprocess.run([exe, str(bad)], capture_output=True)
    assert r.returncode == 2 and b"nested too deeply" in r.stderr


@pytest.mark.gpu
def test_ray_cli_renders_the_same_bytes_as_the_library(scene_file, tmp_path, gpu):
    exe = os.path.join(HOST, "ray")
    assert os.path.exists(exe), "host/ray is not built (python -c 'import __graft_entry__ as g; g.build()')"
    out = tmp_path / "out.ppm"
    r = subprocess.run([exe, "-w", "64", "-H", "40", "-s", "3", scene_file, str(out)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    data = out.read_bytes()
    assert data.startswith(b"P6\n64 40\n255\n")
    img = np.frombuffer(data[len(b"P6\n64 40\n255\n"):], np.uint8).reshape(40, 64, 3)[::-1]   # the file is top-down
    gpu.upload_scene(load_json_scene(scene_file))
    _, rgb8 = gpu.render(abi.default_params(64, 40, spp=3))
    assert np.array_equal(img, rgb8)
    png = tmp_path / "out.png"
    assert subprocess.run([exe, "-w", "32", "-s", "1", scene_file, str(png)], capture_output=True).returncode == 0
    from PIL import Image

    im = np.asarray(Image.open(png))
    assert im.shape == (32, 32, 3)            # "aspectratio" is ignored, so the frame is square (CommandLineUI.cpp:73-74)
    _, rgb8 = gpu.render(abi.default_params(32, 32, spp=1))
    assert np.array_equal(im[::-1], rgb8)
    r = subprocess.run([exe, str(tmp_path / "nope.json"), str(out)], capture_output=True)
    assert r.returncode == 1 and b"Unable to load ray file" in r.stderr


@pytest.mark.gpu
def test_ray_cli_shards_add_up_to_the_single_gpu_frame(scene_file, tmp_path):
    """One process per GPU from the C++ host: three shards of row bands (here one after the other on one
    GPU), combined, are byte for byte the unsharded frame."""
    exe = os.path.join(HOST, "ray")
    whole = tmp_path / "whole.ppm"
    common = ["-w", "64", "-H", "44", "-s", "3"]
    assert subprocess.run([exe] + common + [scene_file, str(whole)], capture_output=True).returncode == 0
    parts = []
    for k in range(3):
        part = tmp_path / f"part{k}.ppm"
        r = subprocess.run([exe] + common + ["--shard", f"{k}/3", "--band-rows", "4", scene_file, str(part)], capture_output=True)
        assert r.returncode == 0, r.stderr.decode()
        parts.append(str(part))
        img = np.frombuffer(part.read_bytes()[len(b"P6\n64 44\n255\n"):], np.uint8).reshape(44, 64, 3)[::-1]
        rows = np.arange(44)
        assert img[(rows // 4) % 3 != k].sum() == 0 and img[(rows // 4) % 3 == k].sum() > 0
    out = tmp_path / "sum.ppm"
    r = subprocess.run([exe, "--combine", str(out)] + parts, capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    assert out.read_bytes() == whole.read_bytes()
    r = subprocess.run([exe, "--combine", str(out), parts[0], parts[0]], capture_output=True)
    assert r.returncode == 1 and b"overlaps" in r.stderr
    assert subprocess.run([exe] + common + ["--shard", "3/3", scene_file, str(out)], capture_output=True).returncode == 1


@pytest.mark.gpu
def test_ray_cli_ranks_over_the_collective_library(scene_file, tmp_path):
    """`ray --ranks N`: the program starts its own ranks, they join through libmiptdist.so (RCCL) and rank 0 writes
    the frame.  One GPU is leased to a test, and RCCL wants a GPU per rank, so the group here has one rank (launch,
    id hand-over, the render through mipt_dist_render and the resolve on the root all run; the collective of a group
    of one is the identity); both shardings give the plain frame byte for byte.  Two ranks on one GPU must fail
    cleanly, not hang."""
    exe = os.path.join(HOST, "ray")
    whole = tmp_path / "whole.ppm"
    common = ["-w", "64", "-H", "44", "-s", "4"]
    assert subprocess.run([exe] + common + [scene_file, str(whole)], capture_output=True).returncode == 0
    for mode in ("samples", "tiles"):
        out = tmp_path / f"ranks_{mode}.ppm"
        r = subprocess.run([exe] + common + ["--ranks", "1", "--sharding", mode, scene_file, str(out)], capture_output=True, timeout=120)
        assert r.returncode == 0, r.stderr.decode()
        assert out.read_bytes() == whole.read_bytes()
    r = subprocess.run([exe] + common + ["--ranks", "1", "--rank", "1", "--id-file", str(tmp_path / "id"), scene_file, str(whole)], capture_output=True)
    assert r.returncode == 1 and b"--ranks" in r.stderr
