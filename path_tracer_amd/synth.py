"""Seeded procedural scenes for the large benchmark configurations.

The reference ships no triangle-heavy scene (its only tri_mesh is a single
triangle), so the 1 M- and 10 M-triangle workloads are generated here, in the
reference's own scene vocabulary: `tri_mesh` objects (one material each, as a
Trimesh has), a floor `square`, `rectangle_area_light`s and a `directional_light`.

Layout: a displaced-grid terrain in four strips plus clusters of displaced
icospheres, 16 meshes = 16 materials with roughness in {0, .2, .., 1} and metallic in
{0, .5, 1}; every second mesh carries per-vertex normals.  Vertex coordinates lie in
[-20, 20] x [-3, 8] x [-60, -10].
"""
from __future__ import annotations

import math

import numpy as np

from . import abi
from .scene import Camera, LightRec, MaterialRec, MeshRec, ObjectRec, Scene, generate_normals


def _icosphere(level: int):
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = np.array(
        [[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]],
        dtype=np.float64,
    )
    v /= np.linalg.norm(v, axis=1)[:, None]
    f = np.array(
        [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
         [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]],
        dtype=np.int64,
    )
    for _ in range(level):
        edges = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        key = np.sort(edges, axis=1)
        uniq, inv = np.unique(key, axis=0, return_inverse=True)
        mid = v[uniq[:, 0]] + v[uniq[:, 1]]
        mid /= np.linalg.norm(mid, axis=1)[:, None]
        base = len(v)
        v = np.concatenate([v, mid])
        n = len(f)
        inv = np.asarray(inv).reshape(-1)
        m01, m12, m20 = base + inv[:n], base + inv[n : 2 * n], base + inv[2 * n :]
        f = np.concatenate(
            [np.stack([f[:, 0], m01, m20], 1), np.stack([f[:, 1], m12, m01], 1), np.stack([f[:, 2], m20, m12], 1), np.stack([m01, m12, m20], 1)]
        )
    return v, f.astype(np.int32)


def _terrain_height(x, z):
    return -2.2 + 0.9 * np.sin(0.35 * x) * np.cos(0.27 * z) + 0.35 * np.sin(1.3 * x + 0.7 * z) + 0.15 * np.cos(2.9 * z - 1.1 * x)


def make_scene(n_triangles: int = 1_000_000, seed: int = 0xC4, n_area_lights: int = 1, cubemap: bool = False, rough_terrain: bool = True) -> Scene:
    """Build the synthetic benchmark scene with about `n_triangles` triangles.

    rough_terrain (the benchmark's choice): the four terrain strips are rough dielectrics
    (roughness 0.8 / 1.0, metallic 0) whatever the material table says.  With False they keep
    the table's entries (roughness 0, 0.2, 0.4, 0.6, two of them metallic): every hit on a
    roughness-0 strip then spawns a mirror ray as well as a bounce ray, and the reference's
    estimator multiplies rays on the ground (1.8 children per surviving hit)."""
    rng = np.random.RandomState(seed)
    sc = Scene()
    cam = Camera()
    # off-axis on purpose: a camera in the plane x = 0 looking down -z would send a whole
    # pixel column exactly along the terrain grid's edges
    cam.set_eye([0.37, 6.0, 6.0])
    cam.set_look([0.03, -0.28, -1.0], [0.0, 1.0, 0.0])
    cam.set_fov(45.0)
    cam.set_aspect(16.0 / 9.0)
    sc.camera = cam
    sc.ambient = np.array([0.05, 0.05, 0.06])

    rough = [0.0, 0.2, 0.4, 0.6, 0.8, 1.0]
    metal = [0.0, 0.5, 1.0]
    mrng = np.random.RandomState(seed + 1)
    for k in range(16):
        m = MaterialRec()
        m.kd = mrng.uniform(0.1, 0.9, size=3)
        m.ka = 0.3 * m.kd
        m.roughness = rough[k % 6]
        m.metallic = metal[(k // 6 + k) % 3]
        m.index = 1.47
        # terrain strips stay rough dielectrics so that paths do not multiply on the ground
        if k < 4 and rough_terrain:
            m.roughness, m.metallic = (0.8, 0.0) if k % 2 == 0 else (1.0, 0.0)
        sc.materials.append(m)

    eye4 = np.eye(4)
    # ---- terrain: G x G cells in four z-strips ---------------------------------------
    G = max(4, int(math.sqrt(n_triangles / 4.0)))
    G -= G % 4
    xs = np.linspace(-20.0, 20.0, G + 1)
    zs = np.linspace(-60.0, -10.0, G + 1)
    rows_per_strip = G // 4
    for strip in range(4):
        z0, z1 = strip * rows_per_strip, (strip + 1) * rows_per_strip
        X, Z = np.meshgrid(xs, zs[z0 : z1 + 1])
        Y = _terrain_height(X, Z)
        verts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1)
        nx = G + 1
        r, c = np.meshgrid(np.arange(z1 - z0), np.arange(G), indexing="ij")
        i00 = (r * nx + c).ravel()
        i01, i10, i11 = i00 + 1, i00 + nx, i00 + nx + 1
        faces = np.concatenate([np.stack([i00, i10, i01], 1), np.stack([i01, i10, i11], 1)]).astype(np.int32)
        normals = generate_normals(verts, faces) if strip % 2 == 0 else None
        sc.meshes.append(MeshRec(verts, faces, normals))
        sc.objects.append(ObjectRec(abi.OBJ_TRIMESH, strip, len(sc.meshes) - 1, eye4.copy()))
    terrain_tris = 2 * G * G

    # ---- blobs: displaced icospheres in 12 meshes ---------------------------------------
    remaining = max(0, n_triangles - terrain_tris)
    # Subdivision level: as coarse as keeps the blob count at or below ~150, so that the
    # blobs can sit on a jittered grid without touching each other or the ground.  (The
    # reference's estimator multiplies rays in narrow gaps between glossy surfaces — each
    # hit spawns up to two children — so near-contact geometry makes path trees explode.)
    level = 1
    while level < 8 and remaining / (20 * 4 ** level) > 150:
        level += 1
    bv, bf = _icosphere(level)
    per_blob = len(bf)
    n_blobs = max(12, int(round(remaining / per_blob)))
    cols = int(math.ceil(math.sqrt(n_blobs * 36.0 / 46.0)))
    rows_ = int(math.ceil(n_blobs / cols))
    cell = min(36.0 / cols, 46.0 / rows_)
    rmax = min(1.5, 0.36 * cell)
    order = rng.permutation(cols * rows_)[:n_blobs]
    blobs = [[] for _ in range(12)]
    for b in range(n_blobs):
        blobs[b % 12].append(int(order[b]))
    for mi in range(12):
        vs, fs, ns = [], [], []
        off = 0
        smooth = mi % 2 == 0
        for cell_id in blobs[mi]:
            gx, gz = cell_id % cols, cell_id // cols
            radius = rng.uniform(0.55, 1.0) * rmax
            jitter = 0.5 * cell - 1.15 * radius
            cx = -18.0 + (gx + 0.5) * (36.0 / cols) + rng.uniform(-1.0, 1.0) * max(0.0, min(jitter, 0.5 * 36.0 / cols - 1.15 * radius))
            cz = -58.0 + (gz + 0.5) * (46.0 / rows_) + rng.uniform(-1.0, 1.0) * max(0.0, min(jitter, 0.5 * 46.0 / rows_ - 1.15 * radius))
            # float well clear of the terrain (whose height varies by < 1.5 over a blob's footprint)
            cy = float(_terrain_height(np.array(cx), np.array(cz))) + 1.5 + radius * rng.uniform(2.0, 3.5)
            cy = min(cy, 8.0 - radius * 1.2)
            freq = rng.uniform(2.0, 5.0)
            disp = 1.0 + 0.12 * np.sin(freq * bv[:, 0] + 1.7) * np.cos(freq * bv[:, 1]) * np.sin(freq * bv[:, 2] + 0.3)
            p = bv * (radius * disp)[:, None] + np.array([cx, cy, cz])
            vs.append(p)
            fs.append(bf + off)
            if smooth:
                ns.append(bv.copy())
            off += len(bv)
        verts = np.concatenate(vs)
        faces = np.concatenate(fs).astype(np.int32)
        normals = np.concatenate(ns) if smooth else None
        sc.meshes.append(MeshRec(verts, faces, normals))
        sc.objects.append(ObjectRec(abi.OBJ_TRIMESH, 4 + mi, len(sc.meshes) - 1, eye4.copy()))

    # ---- floor square under everything -----------------------------------------------
    floor = MaterialRec()
    floor.kd = np.array([0.55, 0.55, 0.55])
    floor.ka = np.array([0.1, 0.1, 0.1])
    sc.materials.append(floor)
    T = np.eye(4)
    T[:3, 3] = [0.0, -4.2, -35.0]
    Sc = np.diag([220.0, 220.0, 220.0, 1.0])
    a = -math.pi / 2.0
    Rx = np.array([[1, 0, 0, 0], [0, math.cos(a), -math.sin(a), 0], [0, math.sin(a), math.cos(a), 0], [0, 0, 0, 1.0]])
    sc.objects.append(ObjectRec(abi.OBJ_SQUARE, len(sc.materials) - 1, -1, T @ Sc @ Rx))

    # ---- lights -------------------------------------------------------------------------
    att = np.array([1.0, 0.0, 0.0], dtype=np.float32)
    if n_area_lights <= 1:
        sc.lights.append(LightRec(abi.LIGHT_RECT_AREA, np.array([2.4, 2.3, 2.1]), position=np.array([-6.0, 16.0, -40.0]),
                                  u=np.array([1.0, 0.0, 0.0]), v=np.array([0.0, 0.0, 1.0]), uL=10.0, vL=10.0, atten=att))
    else:
        side = int(round(math.sqrt(n_area_lights)))
        inten = 6.0 / n_area_lights
        for a_ in range(side):
            for b_ in range(side):
                pos = np.array([-18.0 + 36.0 * a_ / max(1, side - 1), 16.0, -58.0 + 46.0 * b_ / max(1, side - 1)])
                sc.lights.append(LightRec(abi.LIGHT_RECT_AREA, np.array([inten, inten, inten]), position=pos, u=np.array([1.0, 0.0, 0.0]),
                                          v=np.array([0.0, 0.0, 1.0]), uL=1.0, vL=1.0, atten=att))
    sc.lights.append(LightRec(abi.LIGHT_DIRECTIONAL, np.array([1.3, 1.3, 1.4]), direction=np.array([-0.4, -1.0, -0.3])))

    if cubemap:
        sc.cubemap = make_sky_cubemap(64)
    return sc


def make_sky_cubemap(size: int = 64):
    """Six procedural RGB8 faces (+x, -x, +y, -y, +z, -z): a vertical sky gradient with a
    per-face tint, rows stored bottom-up as the reference's image reader returns them."""
    faces = []
    tints = [(1.0, 0.95, 0.9), (0.9, 0.95, 1.0), (0.7, 0.8, 1.0), (0.45, 0.4, 0.35), (0.95, 1.0, 0.95), (1.0, 0.9, 0.95)]
    yy, xx = np.meshgrid(np.linspace(0.0, 1.0, size), np.linspace(0.0, 1.0, size), indexing="ij")
    for k in range(6):
        base = 0.35 + 0.5 * yy + 0.08 * np.sin(9.0 * xx + k)
        img = np.stack([base * tints[k][0], base * tints[k][1], base * tints[k][2]], axis=2)
        faces.append(np.clip(img * 255.0, 0, 255).astype(np.uint8))
    return faces


def small_mesh_scene(seed: int = 7, n_triangles: int = 20000, smooth: bool = True) -> Scene:
    """A small version of the same layout for parity tests (a CPU render finishes in seconds)."""
    return make_scene(n_triangles=n_triangles, seed=seed)


# ------------------------------------------------------------------ variants of the demo scenes
_CUBE_VERTS = np.array([[x, y, z] for z in (-0.5, 0.5) for y in (-0.5, 0.5) for x in (-0.5, 0.5)], dtype=np.float64)
# two triangles per face of the unit cube (Box.cpp:9-13 spans [-0.5, 0.5]^3), outward winding
_CUBE_FACES = np.array(
    [[0, 2, 3], [0, 3, 1],  # z = -0.5
     [4, 5, 7], [4, 7, 6],  # z = +0.5
     [0, 1, 5], [0, 5, 4],  # y = -0.5
     [2, 6, 7], [2, 7, 3],  # y = +0.5
     [0, 4, 6], [0, 6, 2],  # x = -0.5
     [1, 3, 7], [1, 7, 5]],  # x = +0.5
    dtype=np.int32)


def with_rect_area_light(scene: Scene, position, u, v, uL: float, vL: float, color=(1.0, 1.0, 1.0),
                         atten=(0.25, 0.0, 0.0), replace_point_lights: bool = True) -> Scene:
    """A copy of `scene` lit by one rectangular area light (light.cpp:92-140), which costs
    `area_samples` shadow rays per shaded hit where a point light costs one.  The demo Cornell
    box ships with a point light only; this is the area-light variant of it."""
    import copy

    out = copy.deepcopy(scene)
    if replace_point_lights:
        out.lights = [L for L in out.lights if L.type != abi.LIGHT_POINT]
    out.lights.append(LightRec(
        abi.LIGHT_RECT_AREA, np.asarray(color, dtype=np.float64), position=np.asarray(position, dtype=np.float64),
        u=np.asarray(u, dtype=np.float64), v=np.asarray(v, dtype=np.float64), uL=float(uL), vL=float(vL),
        atten=np.asarray(atten, dtype=np.float32)))
    return out


def cornell_area_light(scene: Scene) -> Scene:
    """cornellBoxes with its point light (0, 9, -8) swapped for a strip light at the same place.
    The sampled points are corner + uVec * uI + vVec + (vI, vI, vI) (light.cpp:105-111), so the
    strip is laid out for that: 6 long in x, and vL small enough that the diagonal term stays
    under the ceiling (y = 9.6)."""
    return with_rect_area_light(scene, position=(-3.0, 8.5, -9.0), u=(1.0, 0.0, 0.0), v=(0.0, 0.0, 1.0), uL=6.0, vL=0.5)


def tessellate_boxes(scene: Scene) -> Scene:
    """A copy of `scene` with every Box replaced by a 12-triangle mesh of the same surface:
    vertices in world space under an identity transform.  Mirroring transforms get their
    winding flipped so the face normals (trimesh.h:145-146) keep pointing out of the box."""
    import copy

    out = copy.deepcopy(scene)
    objects = []
    for o in out.objects:
        if o.type != abi.OBJ_BOX:
            objects.append(o)
            continue
        m = np.asarray(o.xform, dtype=np.float64)
        verts = _CUBE_VERTS @ m[:3, :3].T + m[:3, 3]
        faces = _CUBE_FACES.copy()
        if np.linalg.det(m[:3, :3]) < 0:
            faces = faces[:, ::-1].copy()
        out.meshes.append(MeshRec(np.ascontiguousarray(verts), np.ascontiguousarray(faces)))
        objects.append(ObjectRec(abi.OBJ_TRIMESH, o.material, len(out.meshes) - 1, np.eye(4)))
    out.objects = objects
    return out
