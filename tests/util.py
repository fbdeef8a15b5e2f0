"""Small scene builders shared by the tests."""
import math

import numpy as np

from path_tracer_amd import abi
from path_tracer_amd.scene import Camera, LightRec, MaterialRec, MeshRec, ObjectRec, Scene


def translate(x, y, z):
    m = np.eye(4)
    m[:3, 3] = [x, y, z]
    return m


def scale(x, y=None, z=None):
    y = x if y is None else y
    z = x if z is None else z
    return np.diag([x, y, z, 1.0])


def rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1.0]])


def rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1.0]])


def material(kd=(0.8, 0.2, 0.4), ka=(0.0, 0.0, 0.0), kt=(0.0, 0.0, 0.0), roughness=1.0, metallic=0.0, index=1.0):
    return MaterialRec(np.array(kd, float), np.array(ka, float), np.array(kt, float), roughness, metallic, index)


def camera(eye=(0, 0, 5), viewdir=(0, 0, -1), updir=(0, 1, 0), fov=None, aspect=None):
    c = Camera()
    c.set_eye(eye)
    c.set_look(viewdir, updir)
    if fov is not None:
        c.set_fov(fov)
    if aspect is not None:
        c.set_aspect(aspect)
    return c


def dir_light(direction, color=(1, 1, 1)):
    return LightRec(abi.LIGHT_DIRECTIONAL, np.array(color, float), direction=np.array(direction, float))


def point_light(position, color=(1, 1, 1), atten=(0.0, 0.0, 1.0)):
    return LightRec(abi.LIGHT_POINT, np.array(color, float), position=np.array(position, float), atten=np.array(atten, np.float32))


def area_light(corner, u, v, uL, vL, color=(1, 1, 1), atten=(1.0, 0.0, 0.0)):
    return LightRec(abi.LIGHT_RECT_AREA, np.array(color, float), position=np.array(corner, float), u=np.array(u, float),
                    v=np.array(v, float), uL=uL, vL=vL, atten=np.array(atten, np.float32))


def scene(objects=(), lights=(), cam=None, ambient=(0, 0, 0), meshes=(), cubemap=None):
    """objects: iterable of (type, xform 4x4, MaterialRec[, mesh index])."""
    sc = Scene()
    sc.camera = cam or camera()
    sc.ambient = np.array(ambient, float)
    sc.meshes = list(meshes)
    for o in objects:
        sc.materials.append(o[2])
        sc.objects.append(ObjectRec(o[0], len(sc.materials) - 1, o[3] if len(o) > 3 else -1, np.asarray(o[1], float)))
    sc.lights = list(lights)
    sc.cubemap = cubemap
    return sc


def quad_mesh(normals=None):
    """Two triangles covering [-1,1]^2 in the plane z = 0, facing +z."""
    v = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], float)
    f = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    return MeshRec(v, f, normals)


def camera_rays(sc, n, seed=0):
    rng = np.random.RandomState(seed)
    eye, look, u, v = (np.asarray(getattr(sc.camera, k), float) for k in ("eye", "look", "u", "v"))
    xy = rng.uniform(-0.5, 0.5, size=(n, 2))
    d = look[None] + xy[:, :1] * u[None] + xy[:, 1:] * v[None]
    d /= np.linalg.norm(d, axis=1)[:, None]
    return np.repeat(eye[None], n, 0), d


def random_dirs(n, seed=0):
    rng = np.random.RandomState(seed)
    d = rng.normal(size=(n, 3))
    return d / np.linalg.norm(d, axis=1)[:, None]


def rmsd(a, b):
    """raycheck.py:316-324: ||a - b||_2 / sqrt(size) on float [0,1] RGB."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm((a - b).ravel()) / math.sqrt(a.size))


def grid_mesh(n, size, height, seed=0, smooth=False):
    """An n x n displaced grid of 2 n^2 triangles over [-size, size]^2 in the local z = 0 plane,
    heights uniform in [-height, height]."""
    rng = np.random.RandomState(seed)
    xs = np.linspace(-size, size, n + 1)
    gx, gy = np.meshgrid(xs, xs, indexing="ij")
    v = np.stack([gx.ravel(), gy.ravel(), rng.uniform(-height, height, gx.size)], 1)
    idx = np.arange((n + 1) * (n + 1)).reshape(n + 1, n + 1)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    f = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)]).astype(np.int32)
    normals = None
    if smooth:
        normals = np.tile(np.array([[0.0, 0.0, 1.0]]), (len(v), 1)) + rng.normal(scale=0.2, size=(len(v), 3))
        normals /= np.linalg.norm(normals, axis=1)[:, None]
    return MeshRec(v, f, normals)
