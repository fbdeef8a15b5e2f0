"""Wavefront OBJ / MTL ingest for `obj_mesh` geometry (parser/JsonParser.cpp:529-754).

The reference reads OBJ files through tinyobjloader 2.0 (vendored as
ray/src/libs/tiny_obj_loader.h, built with TINYOBJLOADER_USE_DOUBLE and triangulation on,
vertex_color off: JsonParser.cpp:4-5,701-705) and then re-indexes every shape into one
Trimesh (loadObjToTrimesh, JsonParser.cpp:556-690).  This module restates the part of that
pipeline a scene can observe:

* numbers are assembled the way tiny_obj_loader.h:915-1083 does it (digit by digit, decimal
  digits scaled by a table of negative powers of ten, exponent applied as 5^e then ldexp), so
  a coordinate such as "0.35" becomes the same double as in the reference, which is not
  always the correctly rounded one;
* `v x y z [r g b]`, `vn`, `vt`, `f` with i, i/j, i//k, i/j/k corners and negative (relative)
  indices, `g` / `o` start a new shape, `usemtl` does not (tiny_obj_loader.h:2910-3314);
  colours are kept only if every vertex has them (:2917-2928, :3424-3427);
* quads are cut along the shorter diagonal (:1642-1743), larger polygons by the loader's
  built-in ear clipping (:1886-2132), quirks included;
* MTL: newmtl, Ka, Kd, Ks, Kt/Tf, Ke, Ns, Ni, map_Kd, map_Ks (:2326-2575); the mesh takes
  the FIRST material of the library, whatever `usemtl` says (JsonParser.cpp:650-674).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np


class ObjFormatError(ValueError):
    pass


_NEG_POW10 = (1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001)


def parse_double(tok: str) -> Tuple[bool, float]:
    """(ok, value) for one blank-delimited token, tiny_obj_loader.h:915-1083.  Trailing
    characters after a valid prefix are ignored, as there."""
    n = len(tok)
    if n == 0:
        return False, 0.0
    k = 0
    negative = False
    dot_first = False
    if tok[0] in "+-":
        negative = tok[0] == "-"
        k = 1
        dot_first = k < n and tok[k] == "."
    elif tok[0] == ".":
        dot_first = True
    elif not tok[0].isdigit() or not tok[0].isascii():
        return False, 0.0
    mant = 0.0
    if not dot_first:
        start = k
        while k < n and "0" <= tok[k] <= "9":
            mant = mant * 10 + (ord(tok[k]) - 48)
            k += 1
        if k == start:
            return False, 0.0
    exp10 = 0
    if k < n and tok[k] == ".":
        k += 1
        place = 1
        while k < n and "0" <= tok[k] <= "9":
            scale = _NEG_POW10[place] if place < len(_NEG_POW10) else math.pow(10.0, -place)
            mant += (ord(tok[k]) - 48) * scale
            place += 1
            k += 1
    elif k < n and tok[k] not in "eE":
        k = n  # any other character ends the number
    if k < n and tok[k] in "eE":
        k += 1
        exp_neg = False
        if k < n and tok[k] in "+-":
            exp_neg = tok[k] == "-"
            k += 1
        elif not (k < n and "0" <= tok[k] <= "9"):
            return False, 0.0
        start = k
        while k < n and "0" <= tok[k] <= "9":
            if exp10 > 2147483647 // 10:
                return False, 0.0
            exp10 = exp10 * 10 + (ord(tok[k]) - 48)
            k += 1
        if k == start:
            return False, 0.0
        if exp_neg:
            exp10 = -exp10
    if exp10:
        try:
            value = math.ldexp(mant * math.pow(5.0, exp10), exp10)
        except OverflowError:
            value = math.inf
    else:
        value = mant
    return True, -value if negative else value


def _reals(tokens: List[str], count: int, default: float = 0.0) -> List[float]:
    out = []
    for k in range(count):
        ok, v = parse_double(tokens[k]) if k < len(tokens) else (False, 0.0)
        out.append(v if ok else default)
    return out


def _atoi(text: str) -> int:
    """C atoi on the start of `text`: optional sign, then digits; 0 if there are none."""
    k = 0
    sign = 1
    if k < len(text) and text[k] in "+-":
        sign = -1 if text[k] == "-" else 1
        k += 1
    start = k
    while k < len(text) and "0" <= text[k] <= "9":
        k += 1
    return sign * int(text[start:k]) if k > start else 0


@dataclass
class ObjMaterial:  # tinyobj::material_t as InitMaterial leaves it (tiny_obj_loader.h:1501-1540)
    name: str = ""
    ambient: np.ndarray = field(default_factory=lambda: np.zeros(3))
    diffuse: np.ndarray = field(default_factory=lambda: np.zeros(3))
    specular: np.ndarray = field(default_factory=lambda: np.zeros(3))
    transmittance: np.ndarray = field(default_factory=lambda: np.zeros(3))
    emission: np.ndarray = field(default_factory=lambda: np.zeros(3))
    shininess: float = 1.0
    ior: float = 1.0
    diffuse_texname: str = ""
    specular_texname: str = ""


@dataclass
class ObjShape:
    name: str
    corners: List[Tuple[int, int, int]] = field(default_factory=list)  # (v, vt, vn) zero-based, -1 = absent; 3 per triangle


@dataclass
class ObjData:
    vertices: List[Tuple[float, float, float]] = field(default_factory=list)
    normals: List[Tuple[float, float, float]] = field(default_factory=list)
    texcoords: List[Tuple[float, float]] = field(default_factory=list)
    colors: List[Tuple[float, float, float]] = field(default_factory=list)  # empty unless every vertex has one
    shapes: List[ObjShape] = field(default_factory=list)
    materials: List[ObjMaterial] = field(default_factory=list)


_MAP_OPTIONS = {"-blendu": 1, "-blendv": 1, "-clamp": 1, "-boost": 1, "-bm": 1, "-o": 3, "-s": 3, "-t": 3, "-type": 1,
                "-texres": 1, "-imfchan": 1, "-mm": 2, "-colorspace": 1}


def _texture_name(rest: str) -> str:
    """File name of a map_* statement (tiny_obj_loader.h:1352-1469): options and their
    arguments are skipped, the first other token starts the name, which runs to the end of
    the line (names may contain blanks)."""
    pos = 0
    n = len(rest)
    while pos < n:
        while pos < n and rest[pos] in " \t":
            pos += 1
        end = pos
        while end < n and rest[end] not in " \t":
            end += 1
        tok = rest[pos:end]
        if tok in _MAP_OPTIONS and end < n:
            pos = end
            for _ in range(_MAP_OPTIONS[tok]):
                while pos < n and rest[pos] in " \t":
                    pos += 1
                while pos < n and rest[pos] not in " \t":
                    pos += 1
            continue
        return rest[pos:]
    return ""


def read_mtl(path: str) -> List[ObjMaterial]:
    mats: List[ObjMaterial] = []
    cur = ObjMaterial()
    with open(path, "r", errors="replace") as fh:
        for raw in fh:
            line = raw.rstrip("\n").rstrip("\r").rstrip(" \t").lstrip(" \t")
            if not line or line[0] == "#":
                continue
            parts = line.split(None, 1)
            key, rest = parts[0], (parts[1] if len(parts) > 1 else "")
            toks = rest.split()
            if key == "newmtl":
                if cur.name:
                    mats.append(cur)
                cur = ObjMaterial(name=line[7:])
            elif key == "Ka":
                cur.ambient = np.array(_reals(toks, 3))
            elif key == "Kd":
                cur.diffuse = np.array(_reals(toks, 3))
            elif key == "Ks":
                cur.specular = np.array(_reals(toks, 3))
            elif key in ("Kt", "Tf"):
                cur.transmittance = np.array(_reals(toks, 3))
            elif key == "Ke":
                cur.emission = np.array(_reals(toks, 3))
            elif key == "Ni":
                cur.ior = _reals(toks, 1)[0]
            elif key == "Ns":
                cur.shininess = _reals(toks, 1)[0]
            elif key == "map_Kd":
                cur.diffuse_texname = _texture_name(rest)
                # (a map_Kd without Kd also sets Kd to 0.6, :2548-2555; the texture replaces kd anyway)
            elif key == "map_Ks":
                cur.specular_texname = _texture_name(rest)
    if cur.name:  # the last material is flushed at the end of the file (tiny_obj_loader.h:2697-2700)
        mats.append(cur)
    return mats


def _corner(tok: str, nv: int, nvn: int, nvt: int) -> Tuple[int, int, int]:
    def fix(text: str, count: int) -> int:
        i = _atoi(text)
        if i == 0:
            raise ObjFormatError("Failed parse `f' line(e.g. zero value for face index.")
        return i - 1 if i > 0 else count + i

    parts = tok.split("/")
    v = fix(parts[0], nv)
    vt = vn = -1
    if len(parts) == 2:
        vt = fix(parts[1], nvt)
    elif len(parts) >= 3:
        if parts[1] != "":
            vt = fix(parts[1], nvt)
        vn = fix(parts[2], nvn)
    return v, vt, vn


def _inside(vx, vy, tx, ty) -> bool:
    """Crossing-number test of (tx, ty) against the triangle (vx, vy), tiny_obj_loader.h:1557-1569."""
    c = False
    j = 2
    for i in range(3):
        if (vy[i] > ty) != (vy[j] > ty) and tx < (vx[j] - vx[i]) * (ty - vy[i]) / (vy[j] - vy[i]) + vx[i]:
            c = not c
        j = i
    return c


def ear_clip(face: List[Tuple[int, int, int]], vertices) -> List[Tuple[int, int, int]]:
    """Triangles of a polygon with five or more corners the way the reference's loader cuts
    them (its built-in ear clipping, tiny_obj_loader.h:1886-2132), quirks included: the
    polygon is projected on the two axes chosen from its first non-degenerate corner, the
    convexity test multiplies the corner's cross product by (x0 * y1 - y0 * x1) / 2 of the
    corner's first two points, and the walk gives up after a full turn without an ear (what
    is left is dropped unless it is a triangle)."""
    n = len(face)
    eps = 2.220446049250313e-16
    axes = [1, 2]
    for k in range(n):
        p0, p1, p2 = (vertices[face[(k + d) % n][0]] for d in range(3))
        e0 = [p1[a] - p0[a] for a in range(3)]
        e1 = [p2[a] - p1[a] for a in range(3)]
        cx = abs(e0[1] * e1[2] - e0[2] * e1[1])
        cy = abs(e0[2] * e1[0] - e0[0] * e1[2])
        cz = abs(e0[0] * e1[1] - e0[1] * e1[0])
        if cx > eps or cy > eps or cz > eps:
            if not (cx > cy and cx > cz):
                axes[0] = 0
                if cz > cx and cz > cy:
                    axes[1] = 1
            break
    rest = list(face)
    out: List[Tuple[int, int, int]] = []
    guess = 0
    budget = n
    before = n
    while len(rest) > 3 and budget > 0:
        m = len(rest)
        if guess >= m:
            guess -= m
        if before != m:
            before = m
            budget = m
        else:
            budget -= 1
        tri = [rest[(guess + k) % m] for k in range(3)]
        vx = [vertices[c[0]][axes[0]] for c in tri]
        vy = [vertices[c[0]][axes[1]] for c in tri]
        cross = (vx[1] - vx[0]) * (vy[2] - vy[1]) - (vy[1] - vy[0]) * (vx[2] - vx[1])
        area = (vx[0] * vy[1] - vy[0] * vx[1]) * 0.5
        if cross * area < 0.0:
            guess += 1
            continue
        overlap = False
        for other in range(3, m):
            q = vertices[rest[(guess + other) % m][0]]
            if _inside(vx, vy, q[axes[0]], q[axes[1]]):
                overlap = True
                break
        if overlap:
            guess += 1
            continue
        out.extend(tri)
        del rest[(guess + 1) % m]
    if len(rest) == 3:
        out.extend(rest)
    return out


def read_obj(path: str, mtl_dir: Optional[str] = None) -> ObjData:
    """Parse `path`; material libraries are looked up in `mtl_dir` (the scene's directory,
    JsonParser.cpp:702), falling back to default materials when none loads."""
    data = ObjData()
    mtl_dir = os.path.dirname(path) if mtl_dir is None else mtl_dir
    loaded_libs = set()
    all_colored = True
    shape = ObjShape("")
    faces: List[List[Tuple[int, int, int]]] = []
    material = -1
    material_ids = {}

    def flush_faces():
        for f in faces:
            if len(f) < 3:
                continue
            if len(f) == 3:
                shape.corners.extend(f)
            elif len(f) == 4:
                for c in f:
                    if not 0 <= c[0] < len(data.vertices):
                        break
                else:
                    p = [np.array(data.vertices[c[0]]) for c in f]
                    d02 = p[2] - p[0]
                    d13 = p[3] - p[1]
                    s02 = d02[0] * d02[0] + d02[1] * d02[1] + d02[2] * d02[2]
                    s13 = d13[0] * d13[0] + d13[1] * d13[1] + d13[2] * d13[2]
                    order = (0, 1, 2, 0, 2, 3) if s02 < s13 else (0, 1, 3, 1, 2, 3)
                    shape.corners.extend(f[k] for k in order)
            else:
                for c in f:
                    if not 0 <= c[0] < len(data.vertices):
                        raise ObjFormatError(f"{path}: vertex index out of range in a polygon")
                shape.corners.extend(ear_clip(f, data.vertices))
        faces.clear()

    def end_shape(name: str):
        nonlocal shape
        flush_faces()
        if shape.corners:
            data.shapes.append(shape)
        shape = ObjShape(name)

    with open(path, "r", errors="replace") as fh:
        for raw in fh:
            line = raw.rstrip("\n").rstrip("\r").lstrip(" \t")
            if not line or line[0] == "#":
                continue
            head = line[:2]
            if head in ("v ", "v\t"):
                toks = line[2:].split()
                data.vertices.append(tuple(_reals(toks, 3)))
                rgb = [parse_double(toks[k]) if k < len(toks) else (False, 0.0) for k in (3, 4, 5)]
                # all three must parse, tried in order (tiny_obj_loader.h:1153-1154)
                found = rgb[0][0] and rgb[1][0] and rgb[2][0]
                all_colored = all_colored and found
                if all_colored:
                    data.colors.append((rgb[0][1], rgb[1][1], rgb[2][1]))
            elif line[:3] in ("vn ", "vn\t"):
                data.normals.append(tuple(_reals(line[3:].split(), 3)))
            elif line[:3] in ("vt ", "vt\t"):
                data.texcoords.append(tuple(_reals(line[3:].split(), 2)))
            elif head in ("f ", "f\t"):
                nv, nvn, nvt = len(data.vertices), len(data.normals), len(data.texcoords)
                faces.append([_corner(t, nv, nvn, nvt) for t in line[2:].split()])
            elif line.startswith("usemtl"):
                name = line[6:].split()[0] if line[6:].split() else ""
                new_id = material_ids.get(name, -1)
                if new_id != material:
                    flush_faces()
                    material = new_id
            elif line.startswith("mtllib") and len(line) > 6 and line[6] in " \t":
                for name in line[7:].split(" "):
                    if not name:
                        continue
                    if name in loaded_libs:
                        break
                    lib = os.path.join(mtl_dir, name)
                    if os.path.isfile(lib):
                        for m in read_mtl(lib):
                            material_ids.setdefault(m.name, len(data.materials))
                            data.materials.append(m)
                        loaded_libs.add(name)
                        break
            elif head in ("g ", "g\t"):
                names = line[2:].split()
                end_shape(" ".join(names))
            elif head in ("o ", "o\t"):
                end_shape(line[2:])
    end_shape("")
    if not all_colored:
        data.colors = []
    return data


def shape_to_mesh(data: ObjData, shape: ObjShape):
    """loadObjToTrimesh, JsonParser.cpp:556-625: one mesh vertex per distinct (v, vt, vn)
    corner, in order of first use; normals normalised; UVs and colours where present.
    Returns (vertices, faces, normals | None, uvs | None, colors | None)."""
    index = {}
    verts, norms, uvs, cols = [], [], [], []
    faces = []
    tri = []
    for c in shape.corners:
        k = index.get(c)
        if k is None:
            k = len(index)
            index[c] = k
            v, vt, vn = c
            if not 0 <= v < len(data.vertices):
                raise ObjFormatError("Error while parsing OBJ file: vertex index out of range")
            verts.append(data.vertices[v])
            if vn != -1:
                if not 0 <= vn < len(data.normals):
                    raise ObjFormatError("Error while parsing OBJ file: normal index out of range")
                nx, ny, nz = data.normals[vn]
                inv = 1.0 / math.sqrt(nx * nx + ny * ny + nz * nz) if (nx or ny or nz) else math.inf
                norms.append((nx * inv, ny * inv, nz * inv))  # glm::normalize: v * inversesqrt(dot(v, v))
            if vt != -1:
                if not 0 <= vt < len(data.texcoords):
                    raise ObjFormatError("Error while parsing OBJ file: texture coordinate index out of range")
                uvs.append(data.texcoords[vt])
            if data.colors:
                cols.append(data.colors[v])
        tri.append(k)
        if len(tri) == 3:
            faces.append(tri)
            tri = []
    n = len(verts)
    # Trimesh::doubleCheck, trimesh.cpp:50-60
    if cols and len(cols) != n:
        raise ObjFormatError("Error while parsing OBJ file: Bad Trimesh: Wrong number of vertex colors.")
    if uvs and len(uvs) != n:
        raise ObjFormatError("Error while parsing OBJ file: Bad Trimesh: Wrong number of UV coordinates.")
    if norms and len(norms) != n:
        raise ObjFormatError("Error while parsing OBJ file: Bad Trimesh: Wrong number of normals.")
    arr = lambda rows, w: np.array(rows, dtype=np.float64).reshape(-1, w)
    return (arr(verts, 3), np.array(faces, dtype=np.int32).reshape(-1, 3), arr(norms, 3) if norms else None,
            arr(uvs, 2) if uvs else None, arr(cols, 3) if cols else None)
