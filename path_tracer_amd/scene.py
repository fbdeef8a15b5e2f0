"""Host-side scene container and loader for the reference's JSON scene format.

The loader restates what `JsonParser::parseScene` builds (parser/JsonParser.cpp:447-515):
camera (:69-84), material inheritance (:113-131), lights (:133-171), sphere / box /
square / tri_mesh geometry (:179-311) and the transform stack (:59-67, :384-445) with
glm's rotate/scale/translate/make_mat4 semantics.  It produces the flat description
`include/mipt.h` takes (`mipt_scene_desc`); no rendering happens here.

Texture-mapped material parameters ({"mapped": file}) are supported for the two colours
shadeBRDF reads through a texture, diffuse and ambient; mapped specular / reflective /
emissive / normal are parsed and dropped like the constants of those keys (shadeBRDF never
reads them), a mapped transmissive is refused.

Out of scope (raises): cylinder / cone (no path-tracer scene uses them).
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import abi

_PI_CAMERA = 3.14159265359  # scene/camera.cpp:4


@dataclass
class MaterialRec:
    kd: np.ndarray = field(default_factory=lambda: np.zeros(3))
    ka: np.ndarray = field(default_factory=lambda: np.zeros(3))
    kt: np.ndarray = field(default_factory=lambda: np.zeros(3))
    roughness: float = 1.0  # material.h:152-158
    metallic: float = 0.0
    index: float = 1.0
    kd_texture: int = 0  # 0 = constant, k > 0 = Scene.textures[k - 1]
    ka_texture: int = 0

    def copy(self) -> "MaterialRec":
        return MaterialRec(self.kd.copy(), self.ka.copy(), self.kt.copy(), self.roughness, self.metallic, self.index,
                           self.kd_texture, self.ka_texture)


@dataclass
class ObjectRec:
    type: int
    material: int
    mesh: int
    xform: np.ndarray  # 4x4, row-major math convention (xform[r, c])


@dataclass
class MeshRec:
    vertices: np.ndarray  # (n, 3) float64
    faces: np.ndarray  # (m, 3) int32
    normals: Optional[np.ndarray] = None  # (n, 3) float64
    uvs: Optional[np.ndarray] = None  # (n, 2) float64
    colors: Optional[np.ndarray] = None  # (n, 3) float64, read only when uvs is None (trimesh.cpp:170-189)


@dataclass
class LightRec:
    type: int
    color: np.ndarray
    position: np.ndarray = field(default_factory=lambda: np.zeros(3))
    direction: np.ndarray = field(default_factory=lambda: np.zeros(3))
    u: np.ndarray = field(default_factory=lambda: np.zeros(3))
    v: np.ndarray = field(default_factory=lambda: np.zeros(3))
    uL: float = 0.0
    vL: float = 0.0
    atten: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, 1.0], dtype=np.float32))


class Camera:
    """scene/camera.{h,cpp}: the host keeps the camera; the library only receives
    eye / look / u / v after update() (camera.cpp:84-88)."""

    def __init__(self):
        self.aspect = 1.0
        self.normalized_height = 1.0
        self.eye = np.zeros(3)
        self.m = np.eye(3)
        self.update()

    def set_eye(self, eye):
        self.eye = np.asarray(eye, dtype=np.float64)

    def set_look(self, viewdir, updir):  # camera.cpp:57-67 — no normalisation
        z = -np.asarray(viewdir, dtype=np.float64)
        y = np.asarray(updir, dtype=np.float64)
        x = np.cross(y, z)
        self.m = np.stack([x, y, z], axis=1)  # columns x, y, z
        self.update()

    def set_fov(self, fov_deg):  # camera.cpp:69-75
        fov = fov_deg / (180.0 / _PI_CAMERA)
        self.normalized_height = 2 * math.tan(fov / 2)
        self.update()

    def set_aspect(self, ar):
        self.aspect = float(ar)
        self.update()

    def update(self):  # camera.cpp:84-88
        self.u = self.m @ np.array([1.0, 0, 0]) * self.normalized_height * self.aspect
        self.v = self.m @ np.array([0, 1.0, 0]) * self.normalized_height
        self.look = self.m @ np.array([0, 0, -1.0])


@dataclass
class Scene:
    camera: Camera = field(default_factory=Camera)
    ambient: np.ndarray = field(default_factory=lambda: np.zeros(3))
    materials: List[MaterialRec] = field(default_factory=list)
    objects: List[ObjectRec] = field(default_factory=list)
    meshes: List[MeshRec] = field(default_factory=list)
    lights: List[LightRec] = field(default_factory=list)
    cubemap: Optional[List[np.ndarray]] = None  # six (h, w, 3) uint8, row 0 = bottom
    textures: List[np.ndarray] = field(default_factory=list)  # (h, w, 3) uint8, row 0 = bottom

    # ------------------------------------------------------------------ C view
    def to_c(self):
        """Return (mipt_scene_desc, keepalive) — keepalive owns every buffer the
        descriptor points into."""
        keep = []
        d = abi.SceneDesc()
        for name in ("eye", "look", "u", "v"):
            setattr(d.camera, name, abi.c_double3(*[float(x) for x in getattr(self.camera, name)]))
        d.ambient = abi.c_double3(*[float(x) for x in self.ambient])
        mats = (abi.Material * max(1, len(self.materials)))()
        for k, m in enumerate(self.materials):
            mats[k].kd = abi.c_double3(*map(float, m.kd))
            mats[k].ka = abi.c_double3(*map(float, m.ka))
            mats[k].kt = abi.c_double3(*map(float, m.kt))
            mats[k].roughness, mats[k].metallic, mats[k].index = float(m.roughness), float(m.metallic), float(m.index)
            mats[k].kd_texture, mats[k].ka_texture = int(m.kd_texture), int(m.ka_texture)
        objs = (abi.Object * max(1, len(self.objects)))()
        for k, o in enumerate(self.objects):
            objs[k].type, objs[k].material, objs[k].mesh = int(o.type), int(o.material), int(o.mesh)
            col_major = np.asarray(o.xform, dtype=np.float64).T.reshape(16)  # glm memory order
            objs[k].xform = (C.c_double * 16)(*[float(x) for x in col_major])
        meshes = (abi.Mesh * max(1, len(self.meshes)))()
        for k, m in enumerate(self.meshes):
            v = np.ascontiguousarray(m.vertices, dtype=np.float64)
            f = np.ascontiguousarray(m.faces, dtype=np.int32)
            keep += [v, f]
            meshes[k].vertices = v.ctypes.data_as(C.POINTER(C.c_double))
            meshes[k].n_vertices = v.shape[0]
            meshes[k].faces = f.ctypes.data_as(C.POINTER(C.c_int32))
            meshes[k].n_faces = f.shape[0]
            if m.normals is not None:
                n = np.ascontiguousarray(m.normals, dtype=np.float64)
                keep.append(n)
                meshes[k].normals = n.ctypes.data_as(C.POINTER(C.c_double))
            if m.uvs is not None:
                uv = np.ascontiguousarray(m.uvs, dtype=np.float64)
                keep.append(uv)
                meshes[k].uvs = uv.ctypes.data_as(C.POINTER(C.c_double))
            if m.colors is not None:
                col = np.ascontiguousarray(m.colors, dtype=np.float64)
                keep.append(col)
                meshes[k].colors = col.ctypes.data_as(C.POINTER(C.c_double))
        lights = (abi.Light * max(1, len(self.lights)))()
        for k, L in enumerate(self.lights):
            lights[k].type = int(L.type)
            for name in ("color", "position", "direction", "u", "v"):
                setattr(lights[k], name, abi.c_double3(*map(float, getattr(L, name))))
            lights[k].uL, lights[k].vL = float(L.uL), float(L.vL)
            lights[k].atten = abi.c_float3(*[float(np.float32(x)) for x in L.atten])
        d.materials, d.n_materials = mats, len(self.materials)
        d.objects, d.n_objects = objs, len(self.objects)
        d.meshes, d.n_meshes = meshes, len(self.meshes)
        d.lights, d.n_lights = lights, len(self.lights)
        keep += [mats, objs, meshes, lights]
        if self.cubemap is not None:
            faces = (abi.Image * 6)()
            for k in range(6):
                img = np.ascontiguousarray(self.cubemap[k], dtype=np.uint8)
                keep.append(img)
                faces[k].rgb = img.ctypes.data_as(C.POINTER(C.c_uint8))
                faces[k].height, faces[k].width = img.shape[0], img.shape[1]
            d.cubemap = faces
            keep.append(faces)
        if self.textures:
            tex = (abi.Image * len(self.textures))()
            for k, t in enumerate(self.textures):
                img = np.ascontiguousarray(t, dtype=np.uint8)
                keep.append(img)
                tex[k].rgb = img.ctypes.data_as(C.POINTER(C.c_uint8))
                tex[k].height, tex[k].width = img.shape[0], img.shape[1]
            d.textures, d.n_textures = tex, len(self.textures)
            keep.append(tex)
        return d, keep

    # ------------------------------------------------------------ npz fixtures
    def save_npz(self, path):
        arrs = {
            "camera": np.stack([self.camera.eye, self.camera.look, self.camera.u, self.camera.v]),
            "ambient": np.asarray(self.ambient, dtype=np.float64),
            "materials": np.array(
                [np.concatenate([m.kd, m.ka, m.kt, [m.roughness, m.metallic, m.index]]) for m in self.materials],
                dtype=np.float64,
            ).reshape(len(self.materials), 12),
            "obj_meta": np.array([[o.type, o.material, o.mesh] for o in self.objects], dtype=np.int32).reshape(-1, 3),
            "obj_xform": np.array([o.xform for o in self.objects], dtype=np.float64).reshape(-1, 4, 4),
            "light_meta": np.array([L.type for L in self.lights], dtype=np.int32),
            "light_vec": np.array(
                [np.concatenate([L.color, L.position, L.direction, L.u, L.v, [L.uL, L.vL]]) for L in self.lights],
                dtype=np.float64,
            ).reshape(len(self.lights), 17),
            "light_atten": np.array([L.atten for L in self.lights], dtype=np.float32).reshape(-1, 3),
            "n_meshes": np.array([len(self.meshes)], dtype=np.int32),
            "material_textures": np.array([[m.kd_texture, m.ka_texture] for m in self.materials], dtype=np.int32).reshape(-1, 2),
        }
        for k, m in enumerate(self.meshes):
            arrs[f"mesh{k}_v"] = m.vertices
            arrs[f"mesh{k}_f"] = m.faces
            if m.normals is not None:
                arrs[f"mesh{k}_n"] = m.normals
            if m.uvs is not None:
                arrs[f"mesh{k}_uv"] = m.uvs
            if m.colors is not None:
                arrs[f"mesh{k}_c"] = m.colors
        for k, t in enumerate(self.textures):
            arrs[f"texture{k}"] = t
        if self.cubemap is not None:
            for k in range(6):
                arrs[f"cube{k}"] = self.cubemap[k]
        np.savez_compressed(path, **arrs)

    @staticmethod
    def load_npz(path) -> "Scene":
        z = np.load(path, allow_pickle=False)
        s = Scene()
        cam = z["camera"]
        s.camera.eye, s.camera.look, s.camera.u, s.camera.v = cam[0], cam[1], cam[2], cam[3]
        s.ambient = z["ambient"]
        mtex = z["material_textures"] if "material_textures" in z.files else np.zeros((len(z["materials"]), 2), dtype=np.int32)
        for row, tx in zip(z["materials"], mtex):
            s.materials.append(MaterialRec(row[0:3].copy(), row[3:6].copy(), row[6:9].copy(), float(row[9]), float(row[10]), float(row[11]),
                                           int(tx[0]), int(tx[1])))
        for meta, xf in zip(z["obj_meta"], z["obj_xform"]):
            s.objects.append(ObjectRec(int(meta[0]), int(meta[1]), int(meta[2]), xf.copy()))
        for t, vec, att in zip(z["light_meta"], z["light_vec"], z["light_atten"]):
            s.lights.append(LightRec(int(t), vec[0:3].copy(), vec[3:6].copy(), vec[6:9].copy(), vec[9:12].copy(), vec[12:15].copy(), float(vec[15]), float(vec[16]), att.copy()))
        for k in range(int(z["n_meshes"][0])):
            n = z[f"mesh{k}_n"] if f"mesh{k}_n" in z.files else None
            uv = z[f"mesh{k}_uv"] if f"mesh{k}_uv" in z.files else None
            col = z[f"mesh{k}_c"] if f"mesh{k}_c" in z.files else None
            s.meshes.append(MeshRec(z[f"mesh{k}_v"], z[f"mesh{k}_f"], n, uv, col))
        k = 0
        while f"texture{k}" in z.files:
            s.textures.append(z[f"texture{k}"])
            k += 1
        if "cube0" in z.files:
            s.cubemap = [z[f"cube{k}"] for k in range(6)]
        return s

    def n_triangles(self) -> int:
        return int(sum(m.faces.shape[0] for m in self.meshes))


# ----------------------------------------------------------------- glm helpers
def _translate(v):
    m = np.eye(4)
    m[:3, 3] = v
    return m


def _scale(v):
    return np.diag([v[0], v[1], v[2], 1.0])


def _rotate(angle, axis):
    """glm::rotate(dmat4(1), angle, axis) — radians, axis normalised inside."""
    a = np.asarray(axis, dtype=np.float64)
    a = a * (1.0 / math.sqrt(float(a @ a)))
    c, s = math.cos(angle), math.sin(angle)
    t = (1.0 - c) * a
    R = np.eye(4)
    # glm stores Rotate[col][row]; R[row, col] here.
    R[0, 0] = c + t[0] * a[0]
    R[1, 0] = t[0] * a[1] + s * a[2]
    R[2, 0] = t[0] * a[2] - s * a[1]
    R[0, 1] = t[1] * a[0] - s * a[2]
    R[1, 1] = c + t[1] * a[1]
    R[2, 1] = t[1] * a[2] + s * a[0]
    R[0, 2] = t[2] * a[0] + s * a[1]
    R[1, 2] = t[2] * a[1] - s * a[0]
    R[2, 2] = c + t[2] * a[2]
    return R


_TRANSFORM_KEYS = ("rotate", "scale", "translate", "transform")
_GEOM_KEYS = ("sphere", "box", "square", "cylinder", "cone", "tri_mesh", "obj_mesh")


class SceneFormatError(ValueError):
    pass


def _vec3(j):
    if not isinstance(j, (list, tuple)) or len(j) < 3:
        raise TypeError("not a vec3")
    return np.array([float(j[0]), float(j[1]), float(j[2])])


def load_image(path: str) -> np.ndarray:
    """readImage as TextureMap uses it (material.cpp:172-184, fileio/pngimage.cpp:199-221):
    8-bit RGB, rows bottom-up.  Palette and grey images are expanded to RGB like libpng is
    told to; images with an alpha channel are refused (the reference keeps four channels
    and then indexes them as three, material.cpp:212-222: not reproduced), and so are files
    whose gamma chunk would make libpng rescale the samples (gamma * 2.2 != 1)."""
    from PIL import Image

    im = Image.open(path)
    g = im.info.get("gamma")
    if g is not None and abs(float(g) * 2.2 - 1.0) > 0.011:  # libpng skips gamma within 0.05 / ...; 0.45455 * 2.2 = 1.00001
        raise NotImplementedError(f"{path}: gamma {g} would be corrected by libpng, not supported")
    if im.mode in ("RGBA", "LA", "PA") or "transparency" in im.info:
        raise NotImplementedError(f"{path}: textures with an alpha channel are not supported")
    a = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return np.ascontiguousarray(a[::-1])


class _Parser:
    def __init__(self, scene_dir: str = "."):
        self.scene = Scene()
        self.cur_mat = MaterialRec()
        self.stack: List[np.ndarray] = []
        self.scene_dir = scene_dir
        self.texture_cache = {}

    def texture_index(self, name: str) -> int:  # Scene::getTexture, scene.cpp:140-147: one TextureMap per path
        path = os.path.join(self.scene_dir, name)
        if path not in self.texture_cache:
            self.scene.textures.append(load_image(path))
            self.texture_cache[path] = len(self.scene.textures)
        return self.texture_cache[path]

    def current_transform(self):  # JsonParser.cpp:59-67
        m = np.eye(4)
        for t in self.stack:
            m = m @ t
        return m

    # JsonParser.cpp:86-131.  A missing or ill-typed key leaves the value unchanged
    # (IGNORE_MISSING swallows json exceptions).
    def parse_material(self, j) -> MaterialRec:
        m = self.cur_mat.copy()

        def param(key):
            if not isinstance(j, dict) or key not in j:
                return None
            p = j[key]
            if isinstance(p, dict) and "constant" in p:
                try:
                    return _vec3(p["constant"])
                except (TypeError, ValueError):
                    return None
            if isinstance(p, dict) and "mapped" in p:
                return str(p["mapped"])
            raise SceneFormatError("Material parameter must be either constant or mapped")

        for key, attr in (("ambient", "ka"), ("diffuse", "kd"), ("transmissive", "kt")):
            v = param(key)
            if isinstance(v, str):
                if attr == "kt":
                    raise NotImplementedError("a texture-mapped transmissive colour is not supported")
                setattr(m, attr + "_texture", self.texture_index(v))
            elif v is not None:
                setattr(m, attr, v)
                if attr != "kt":
                    setattr(m, attr + "_texture", 0)
        for key in ("specular", "reflective", "emissive", "normal"):
            param(key)  # parsed by the reference, never read by shadeBRDF (a mapped one would also load the file)
        for key, attr in (("index", "index"), ("metallic", "metallic"), ("roughness", "roughness")):
            if isinstance(j, dict) and isinstance(j.get(key), (int, float)) and not isinstance(j.get(key), bool):
                setattr(m, attr, float(j[key]))
        return m

    def material_index(self, body) -> int:  # GET_MAT_W_CUR, JsonParser.cpp:40-41
        m = self.parse_material(body["material"]) if isinstance(body, dict) and "material" in body else self.cur_mat.copy()
        self.scene.materials.append(m)
        return len(self.scene.materials) - 1

    def parse_geometry(self, obj):
        if len(obj) != 1:
            raise SceneFormatError("Geometry must be a one-element object")
        key, val = next(iter(obj.items()))
        xf = self.current_transform()
        if key in ("sphere", "box", "square"):
            t = {"sphere": abi.OBJ_SPHERE, "box": abi.OBJ_BOX, "square": abi.OBJ_SQUARE}[key]
            self.scene.objects.append(ObjectRec(t, self.material_index(val), -1, xf))
        elif key == "tri_mesh":
            self.parse_trimesh(val, xf)
        elif key == "obj_mesh":
            self.parse_objmesh(val, xf)
        elif key in ("cylinder", "cone"):
            raise NotImplementedError(f"{key} is out of scope of the path-tracing hot path")
        else:
            raise SceneFormatError("Unknown geometry type: " + key)

    def parse_trimesh(self, j, xf):  # JsonParser.cpp:231-311
        mat = self.material_index(j)
        verts = np.array([[float(p[0]), float(p[1]), float(p[2])] for p in j["points"]], dtype=np.float64).reshape(-1, 3)
        faces = []
        for f in j["faces"]:
            if len(f) == 3:
                faces.append([f[0], f[1], f[2]])
            elif len(f) == 4:
                faces.append([f[0], f[1], f[2]])
                faces.append([f[0], f[2], f[3]])
            else:
                raise SceneFormatError(f"Got {len(f)} indices in a face: must be 3 or 4 indices")
        faces = np.array(faces, dtype=np.int32).reshape(-1, 3)
        if faces.size and (faces.max() >= len(verts)):
            raise SceneFormatError("Error while adding face. Maybe the point doesn't exist?")
        normals = None
        if "normals" in j:
            normals = np.array([[float(n[0]), float(n[1]), float(n[2])] for n in j["normals"]], dtype=np.float64).reshape(-1, 3)
            if len(normals) != len(verts):
                raise SceneFormatError("Bad Trimesh: Wrong number of normals.")
        uvs = None
        if "UV" in j:  # JsonParser.cpp:271-279
            uvs = np.array([[float(t[0]), float(t[1])] for t in j["UV"]], dtype=np.float64).reshape(-1, 2)
            if len(uvs) == 0:
                uvs = None
            elif len(uvs) != len(verts):
                raise SceneFormatError("Error in mesh: Bad Trimesh: Wrong number of UV coordinates.")
        if "materials" in j:
            raise SceneFormatError('"materials" key is no longer supported: use obj_mesh instead')
        if j.get("gennormals") is True:
            normals = generate_normals(verts, faces, normals)
        self.scene.meshes.append(MeshRec(verts, faces, normals, uvs))
        self.scene.objects.append(ObjectRec(abi.OBJ_TRIMESH, mat, len(self.scene.meshes) - 1, xf))

    def parse_objmesh(self, j, xf):  # JsonParser.cpp:692-754 and loadObjToTrimesh, :556-690
        from . import objfile

        path = os.path.join(self.scene_dir, str(j["objfile"]))
        try:
            data = objfile.read_obj(path, self.scene_dir)
        except (OSError, objfile.ObjFormatError) as e:
            raise SceneFormatError(f"Error while parsing OBJ file: {e}") from e
        for shape in data.shapes:
            try:
                verts, faces, normals, uvs, colors = objfile.shape_to_mesh(data, shape)
            except objfile.ObjFormatError as e:
                raise SceneFormatError(str(e)) from e
            # a fresh Material, then the FIRST material of the library (:650-674); the scene's
            # current material is not inherited
            m = MaterialRec()
            if data.materials:
                mtl = data.materials[0]
                m.kd, m.ka, m.kt = mtl.diffuse.copy(), mtl.ambient.copy(), mtl.transmittance.copy()
                m.index = float(mtl.ior)
                if mtl.diffuse_texname:
                    m.kd_texture = self.texture_index(mtl.diffuse_texname)
                if mtl.specular_texname:
                    load_image(os.path.join(self.scene_dir, mtl.specular_texname))  # loaded, never read by shadeBRDF
            self.scene.materials.append(m)
            if j.get("gennormals") is True:
                normals = generate_normals(verts, faces, normals)
            self.scene.meshes.append(MeshRec(verts, faces, normals, uvs, colors))
            self.scene.objects.append(ObjectRec(abi.OBJ_TRIMESH, len(self.scene.materials) - 1, len(self.scene.meshes) - 1, xf))

    def parse_transform(self, obj):  # JsonParser.cpp:384-445
        key, val = next(iter(obj.items()))
        children = val[1]
        if key == "rotate":
            self.stack.append(_rotate(float(val[1]), _vec3(val[0])))
            children = val[2]
        elif key == "scale":
            self.stack.append(_scale(_vec3(val[0])))
        elif key == "translate":
            self.stack.append(_translate(_vec3(val[0])))
        elif key == "transform":
            t = [float(x) for x in val[0]]
            self.stack.append(np.array(t, dtype=np.float64).reshape(4, 4).T)  # make_mat4: column-major
        else:
            raise SceneFormatError("Unknown transform type: " + key)
        for child in children:
            ckey = next(iter(child.keys()))
            if ckey in _TRANSFORM_KEYS:
                self.parse_transform(child)
            else:
                self.parse_geometry(child)
        self.stack.pop()

    def parse_camera(self, j):  # JsonParser.cpp:69-84
        c = Camera()
        try:
            c.set_eye(_vec3(j["position"]))
        except (KeyError, TypeError, ValueError):
            pass
        if "updir" in j or "viewdir" in j:
            c.set_look(_vec3(j["viewdir"]), _vec3(j["updir"]))
        if isinstance(j.get("fov"), (int, float)):
            c.set_fov(float(j["fov"]))
        if isinstance(j.get("aspectRatio"), (int, float)):  # note the key: "aspectratio" is ignored
            c.set_aspect(float(j["aspectRatio"]))
        return c

    def parse_light(self, key, j):  # JsonParser.cpp:133-171
        color = _vec3(j["color"])
        if key == "directional_light":
            return LightRec(abi.LIGHT_DIRECTIONAL, color, direction=_vec3(j["direction"]))
        att = np.array([0.0, 0.0, 1.0], dtype=np.float32)
        for k, name in enumerate(("constant_attenuation_coeff", "linear_attenuation_coeff", "quadratic_attenuation_coeff")):
            if isinstance(j.get(name), (int, float)):
                att[k] = np.float32(j[name])
        if key == "point_light":
            return LightRec(abi.LIGHT_POINT, color, position=_vec3(j["position"]), atten=att)
        return LightRec(
            abi.LIGHT_RECT_AREA, color, position=_vec3(j["position"]), u=_vec3(j["u"]), v=_vec3(j["v"]),
            uL=float(j["uL"]), vL=float(j["vL"]), atten=att,
        )

    def parse_scene(self, j) -> Scene:  # JsonParser.cpp:447-515
        for obj in j:
            key, val = next(iter(obj.items()))
            if key == "camera":
                self.scene.camera = self.parse_camera(val)
            elif key == "material":
                self.cur_mat = MaterialRec()
                self.cur_mat = self.parse_material(val)
            elif key == "ambient_light":
                self.scene.ambient = self.scene.ambient + _vec3(val["color"])
            elif key in ("directional_light", "point_light", "rectangle_area_light"):
                self.scene.lights.append(self.parse_light(key, val))
            elif key in _TRANSFORM_KEYS:
                self.parse_transform(obj)
            elif key in _GEOM_KEYS:
                self.parse_geometry(obj)
            else:
                raise SceneFormatError("Unknown scene object type: " + key)
        return self.scene


def generate_normals(verts, faces, normals=None):
    """Trimesh::generateNormals (trimesh.cpp:204-231): per-vertex mean of the unit face
    normals of the non-degenerate faces (not re-normalised)."""
    n = np.zeros_like(verts) if normals is None else normals.copy()
    cnt = np.zeros(len(verts), dtype=np.int64)
    a, b, c = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    ok = (np.linalg.norm(b - a, axis=1) != 0) & (np.linalg.norm(c - a, axis=1) != 0) & (np.linalg.norm(b - c, axis=1) != 0)
    fn = np.cross(b - a, c - a)
    with np.errstate(invalid="ignore", divide="ignore"):
        fn = fn * (1.0 / np.sqrt(np.sum(fn * fn, axis=1)))[:, None]
    for k in range(3):
        np.add.at(n, faces[ok, k], fn[ok])
        np.add.at(cnt, faces[ok, k], 1)
    nz = cnt > 0
    n[nz] /= cnt[nz, None]
    return n


def load_json_scene(path: str) -> Scene:
    """RayTracer::loadScene for a .json scene (RayTracer.cpp:333-415)."""
    with open(path, "r") as fh:
        j = json.load(fh)
    return _Parser(os.path.dirname(os.path.abspath(path))).parse_scene(j)


def parse_json_scene(text: str, scene_dir: str = ".") -> Scene:
    return _Parser(scene_dir).parse_scene(json.loads(text))
