"""Randomised scenes: every primitive kind under random (non-uniform, rotated) transforms,
random meshes with and without vertex normals, all three light types, opaque and
translucent materials.  The HIP path must agree with the oracle per ray (closest hit and
normal), per shading point (shadow attenuation of every light) and per pixel (a small
deterministic frame and a small frame with the reference's settings).

The reference ships no test of this kind (it has no tests at all): parity unpinned, the
oracle is the checker.
"""
import math

import numpy as np
import pytest

from path_tracer_amd import abi
from path_tracer_amd.scene import MeshRec
from tests import util as U

pytestmark = pytest.mark.gpu

FRAME_FRAC = 0.998  # share of a frame's pixels that must agree to 1e-5 (tests/test_gpu_parity.py states the same bound)


def _rot(rng):
    a, b, c = rng.uniform(0, 2 * math.pi, 3)
    rz = np.array([[math.cos(c), -math.sin(c), 0, 0], [math.sin(c), math.cos(c), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])
    return U.rot_x(a) @ U.rot_y(b) @ rz


def _xform(rng, spread):
    s = rng.uniform(0.3, 1.6, 3)
    return U.translate(*rng.uniform(-spread, spread, 3)) @ _rot(rng) @ U.scale(*s)


def _material(rng, translucent):
    kt = rng.uniform(0.1, 0.9, 3) if translucent and rng.rand() < 0.5 else (0, 0, 0)
    return U.material(kd=rng.uniform(0.05, 0.95, 3), ka=rng.uniform(0, 0.3, 3), kt=kt, roughness=float(rng.choice([0.0, 0.15, 0.5, 1.0])),
                      metallic=float(rng.choice([0.0, 0.5, 1.0])), index=float(rng.uniform(1.0, 2.0)))


def _mesh(rng, n_faces, smooth):
    """A random triangle soup around the origin (some faces share vertices)."""
    n_verts = max(3, n_faces + 2)
    v = rng.uniform(-1.0, 1.0, size=(n_verts, 3))
    f = np.stack([np.arange(n_faces), np.arange(n_faces) + 1, np.arange(n_faces) + 2], 1).astype(np.int32) % n_verts
    normals = None
    if smooth:
        normals = rng.normal(size=(n_verts, 3))
        normals /= np.linalg.norm(normals, axis=1)[:, None]
    return MeshRec(v, f, normals)


def random_scene(seed, big_mesh=0, n_lights=0):
    """big_mesh > 0: one more mesh of that many faces (enough primitives for the device builder);
    n_lights > 0: that many lights of random kinds instead of one to three."""
    rng = np.random.RandomState(seed)
    translucent = seed % 3 == 0
    spread = 2.5
    objs, meshes = [], []
    for _ in range(rng.randint(1, 6)):
        objs.append((int(rng.choice([abi.OBJ_SPHERE, abi.OBJ_BOX, abi.OBJ_SQUARE])), _xform(rng, spread), _material(rng, translucent)))
    for _ in range(rng.randint(0, 3)):
        meshes.append(_mesh(rng, int(rng.choice([1, 2, 7, 40, 300])), smooth=bool(rng.rand() < 0.5)))
        objs.append((abi.OBJ_TRIMESH, _xform(rng, spread), _material(rng, translucent), len(meshes) - 1))
    if big_mesh:
        meshes.append(_mesh(rng, big_mesh, smooth=bool(seed % 2)))
        objs.append((abi.OBJ_TRIMESH, U.scale(2.0), _material(rng, False), len(meshes) - 1))
    # a floor so that most camera rays end somewhere
    objs.append((abi.OBJ_SQUARE, U.translate(0, -3.5, 0) @ U.rot_x(-math.pi / 2) @ U.scale(40), _material(rng, False)))
    lights = []
    kinds = rng.permutation(3)[: rng.randint(1, 4)] if not n_lights else rng.randint(0, 3, size=n_lights)
    for k in kinds:
        if k == 0:
            lights.append(U.dir_light(rng.normal(size=3) + np.array([0, -2.0, 0]), color=rng.uniform(0.3, 1.0, 3)))
        elif k == 1:
            lights.append(U.point_light(rng.uniform(-4, 4, 3) + np.array([0, 6.0, 0]), color=rng.uniform(0.3, 1.0, 3),
                                        atten=(float(rng.uniform(0, 1)), float(rng.uniform(0, 0.2)), float(rng.uniform(0, 0.05)))))
        else:
            lights.append(U.area_light(rng.uniform(-2, 2, 3) + np.array([0, 6.0, 0]), (1, 0, 0), (0, 0, 1), float(rng.uniform(0.5, 3)),
                                       float(rng.uniform(0.5, 3)), color=rng.uniform(0.3, 1.0, 3), atten=(1.0, 0.0, float(rng.uniform(0, 0.02)))))
    cam = U.camera(eye=(0.3, 1.0, 9.0), viewdir=(0.0, -0.15, -1.0), fov=50.0)
    cubemap = None
    if seed % 4 == 1:
        from path_tracer_amd import synth

        cubemap = synth.make_sky_cubemap(16)
    return U.scene(objs, lights=lights, cam=cam, ambient=rng.uniform(0, 0.2, 3), meshes=meshes, cubemap=cubemap)


@pytest.mark.parametrize("seed", list(range(1, 33)))
def test_random_scene_matches_the_oracle(seed, gpu, oracle_lib):
    check_scene(random_scene(seed), seed, gpu, oracle_lib)


@pytest.mark.parametrize("builder", ["host", "gpu"])
@pytest.mark.parametrize("seed", [101, 102, 103])
def test_random_scene_with_a_large_triangle_soup(seed, builder, gpu, oracle_lib, monkeypatch):
    """6000 random triangles piercing one another (long thin overlapping boxes: the worst case for
    a BVH), through the host builder and through the device builder."""
    if builder == "gpu":
        monkeypatch.setenv("MIPT_BUILDER", "gpu")
    check_scene(random_scene(seed, big_mesh=6000), seed, gpu, oracle_lib, n_rays=3000, n_fire=10, frame=None)


@pytest.mark.parametrize("seed", [201, 202, 203])
def test_random_scene_with_many_lights(seed, gpu, oracle_lib):
    """Seven and twelve lights of mixed kinds: with more than four the kernels find a shadow ray's
    light in a table instead of scanning the light list."""
    check_scene(random_scene(seed, n_lights=7 if seed != 203 else 12), seed, gpu, oracle_lib, n_rays=3000, n_fire=600, frame=(48, 36))


def check_scene(sc, seed, gpu, oracle_lib, n_rays=6000, n_fire=100000, frame=(80, 60)):
    orc = oracle_lib.Oracle(sc)
    gpu.upload_scene(sc)
    # --- closest hits: camera rays, then bounce-style rays from the hit points
    org, d = U.camera_rays(sc, n_rays, seed=seed)
    tg, pg, ng = gpu.intersect(org, d)
    to, po, no = orc.intersect(org, d)
    same = pg == po
    assert same.mean() >= 0.9995, f"{(~same).sum()} of {len(same)} camera rays found a different primitive"
    hit = same & (po >= 0)
    assert hit.mean() > 0.3
    assert np.all(np.abs(tg[hit] - to[hit]) <= 1e-9 * np.maximum(1.0, to[hit]))
    assert np.all(np.abs(ng[hit] - no[hit]) <= 1e-9)
    P = org[hit] + to[hit, None] * d[hit]
    d2 = U.random_dirs(len(P), seed=seed + 100)
    tg2, pg2, _ = gpu.intersect(P + d2 * 1e-8, d2)
    to2, po2, _ = orc.intersect(P + d2 * 1e-8, d2)
    same2 = pg2 == po2
    assert same2.mean() >= 0.999, f"{(~same2).sum()} of {len(same2)} bounce rays found a different primitive"
    h2 = same2 & (po2 >= 0)
    assert np.all(np.abs(tg2[h2] - to2[h2]) <= 1e-9 * np.maximum(1.0, to2[h2]))
    # --- shadow attenuation of every light from the hit points (any-hit or the occluder walk)
    fire = (P + no[hit] * 3e-8)[:n_fire]
    keys = np.arange(len(fire), dtype=np.uint64) * np.uint64(2654435761) + np.uint64(seed)
    p = abi.default_params(8, 8)
    for li in range(len(sc.lights)):
        a = gpu.shadow_attenuation(p, li, fire, keys)
        b = orc.shadow_attenuation(p, li, fire, keys)
        bad = np.abs(a - b).max(axis=1) > 1e-9
        assert bad.mean() <= 3e-3, f"light {li}: {bad.sum()} of {len(fire)} shading points differ"
    # --- frames: deterministic 1 spp, then the reference's settings with shared random streams (depth capped: random
    # mirrors among random triangles make path trees that grow by half per level, in the reference without end)
    if frame is None:  # the oracle's literal occluder walks through a dense soup take seconds per ray
        return
    for prm in (abi.default_params(frame[0], frame[1], spp=1, rr_q=0.0, max_depth=4, flags=abi.FLAG_FIXED_AREA_PATTERN),
                abi.default_params(frame[0] // 2, frame[1] // 2, spp=4, max_depth=6)):
        f32, _ = gpu.render(prm, want_rgb8=False)
        ref = orc.render(prm, threads=8)
        diff = np.abs(f32.astype(np.float64) - ref).max(axis=2)
        assert (diff <= 1e-5).mean() >= FRAME_FRAC, f"only {(diff <= 1e-5).mean():.4f} of the pixels within 1e-5 (max {diff.max():.3e})"
