"""The shadow kernels a render launches, ray by ray.

mipt_shadow_attenuation walks one hit's rays in a plain loop (one thread per hit); a frame instead goes through
the persistent any-hit kernel (work feed, ray replacement, the own-triangle shortcut) in the instantiation the
scene selects — one-byte or fp64 records, with or without the analytic primitives' tests (and then the pass
that clears what those block), with or without triangles, with the local-unit thresholds — or through the
translucent walk.  mipt_shadow_rays runs exactly those launches on caller-given shaded hits and returns every
ray's value; the oracle gives the same per ray (one pass of the loop of light.cpp:118-144 per sample).

Bar: every ray equal to 1e-12 relative, except a handful of grazing rays that may flip (<= 0.1 %).
"""
import math

import numpy as np
import pytest

from path_tracer_amd import abi, synth
from tests import util as U
from tests.test_gpu_local_units import XFORMS, _scaled_scene
from tests.test_gpu_parity import _mesh_scene_with_analytic_occluders, upload

pytestmark = pytest.mark.gpu


def shaded_hits(sc, orc, n, seed):
    """Hit points as k_shade lists them: camera hits and the hits of rays bounced off those, with the primitive
    each lies on."""
    org, d = U.camera_rays(sc, n, seed=seed)
    t, prim, nrm = orc.intersect(org, d)
    hit = prim >= 0
    P, N, prims = org[hit] + t[hit, None] * d[hit], nrm[hit], prim[hit]
    d2 = U.random_dirs(len(P), seed=seed + 1)
    d2 *= np.sign(np.sum(d2 * N, axis=1))[:, None]                       # into the hemisphere of the normal, like a bounce
    o2 = P + d2 * 1e-8
    t2, prim2, n2 = orc.intersect(o2, d2)
    h2 = prim2 >= 0
    P = np.concatenate([P, o2[h2] + t2[h2, None] * d2[h2]])
    N = np.concatenate([N, n2[h2]])
    prims = np.concatenate([prims, prim2[h2]])
    keys = (np.arange(len(P), dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(seed)
    return P, N, prims.astype(np.int32), keys


def oracle_rays(sc, orc, p, P, N, keys):
    fire = P + (N * 1e-8) * 3.0                                          # material.cpp:117, the same roundings on both sides
    per_light = []
    for li, L in enumerate(sc.lights):
        S = p.area_samples if L.type == abi.LIGHT_RECT_AREA else 1
        per_light.append(orc.shadow_samples(p, li, fire, keys, samples=S))
    return np.concatenate(per_light, axis=1)


def compare(gpu, sc, orc, n=4000, seed=5, area_samples=10, max_flips=1e-3, min_blocked=0.02):
    p = abi.default_params(8, 8)
    p.area_samples = area_samples
    P, N, prims, keys = shaded_hits(sc, orc, n, seed)
    assert len(P) > n // 4
    want = oracle_rays(sc, orc, p, P, N, keys)
    got = gpu.shadow_rays(p, P, N, keys, prims, want.shape[1])
    assert not np.isnan(got).any(), "a ray the launch never wrote"
    bad = np.abs(got - want).max(axis=2) > 1e-12 * (1.0 + np.abs(want).max(axis=2))
    assert bad.mean() <= max_flips, f"{bad.sum()} of {bad.size} shadow rays differ"
    blocked = (want.max(axis=2) == 0)
    assert min_blocked < blocked.mean() < 0.98, blocked.mean()            # both outcomes occur
    return got, want


@pytest.mark.parametrize("lean,vis_bytes", [("1", "1"), ("1", "0"), ("0", "1"), ("0", "0")])
def test_mesh_scene_kernels(gpu, oracle_lib, monkeypatch, lean, vis_bytes):
    """20 000 triangles over a floor square, an area light and a directional one: the four instantiations a mesh
    scene can run (analytic tests in the walk or in the pass after it; one-byte or fp64 records)."""
    monkeypatch.setenv("MIPT_LEAN_ANY", lean)
    monkeypatch.setenv("MIPT_VIS_BYTES", vis_bytes)
    sc = synth.make_scene(20000, seed=3)
    upload(gpu, sc)
    compare(gpu, sc, oracle_lib.Oracle(sc))


@pytest.mark.parametrize("lean", ["1", "0"])
def test_analytic_primitives_beside_meshes(gpu, oracle_lib, monkeypatch, lean):
    """A sphere, a box and a square over the terrain: hits on them and rays they block, found by the pass over the
    analytic primitives (lean) or inside the walk."""
    monkeypatch.setenv("MIPT_LEAN_ANY", lean)
    sc = _mesh_scene_with_analytic_occluders()
    orc = oracle_lib.Oracle(sc)
    upload(gpu, sc)
    compare(gpu, sc, orc, n=6000, seed=9)


@pytest.mark.parametrize("name", ["spheres", "cornellBoxes", "hitchcockBRDF", "areaLight"])
def test_scenes_without_triangles(gpu, scenes, oracles, name):
    sc = scenes(name)
    upload(gpu, sc)
    compare(gpu, sc, oracles(name), n=3000, seed=13)


@pytest.mark.parametrize("name", ["scale_0.01", "nonuniform"])
def test_meshes_with_local_unit_thresholds(gpu, oracle_lib, name):
    floor = (abi.OBJ_SQUARE, U.translate(0, -3, -9) @ U.rot_x(-math.pi / 2) @ U.scale(60), U.material(kd=(0.8, 0.8, 0.8)))
    lights = [U.area_light((-2, 9, -11), (4, 0, 0), (0, 0, 4), 1, 1), U.dir_light((0.2, -1, -0.1)), U.point_light((3, 6, -4), atten=(1, 0, 0))]
    sc = _scaled_scene(XFORMS[name], n=24, lights=lights, extra=[floor])
    upload(gpu, sc)
    compare(gpu, sc, oracle_lib.Oracle(sc), n=6000, seed=21, max_flips=2e-3, min_blocked=0.02 if name == "scale_0.01" else 0.002)


def test_translucent_walk(gpu, oracle_lib):
    glass = U.material(kd=(0.2, 0.2, 0.2), kt=(0.5, 0.25, 0.9))
    floor = U.material(kd=(0.8, 0.8, 0.8))
    sc = U.scene([(abi.OBJ_BOX, U.translate(0, 2, 0) @ U.scale(3, 1, 3), glass), (abi.OBJ_SPHERE, U.translate(2.5, 2, 0.5), glass),
                  (abi.OBJ_SQUARE, U.rot_x(-math.pi / 2) @ U.scale(30), floor)],
                 lights=[U.point_light((0.5, 9, 0.5), atten=(1, 0, 0)), U.dir_light((0.1, -1, 0.2)), U.area_light((-1, 8, -1), (1, 0, 0), (0, 0, 1), 2, 2)],
                 cam=U.camera(eye=(0, 6, 12), viewdir=(0, -0.4, -1)))
    orc = oracle_lib.Oracle(sc)
    upload(gpu, sc)
    p = abi.default_params(8, 8)
    P, N, prims, keys = shaded_hits(sc, orc, 3000, 31)
    want = oracle_rays(sc, orc, p, P, N, keys)
    got = gpu.shadow_rays(p, P, N, keys, prims, want.shape[1])
    assert ((want > 0) & (want < 0.99)).any()                            # some rays really pass through glass
    bad = np.abs(got - want).max(axis=2) > 1e-9
    assert bad.mean() <= 5e-3, f"{bad.sum()} of {bad.size}"


def test_many_lights(gpu, oracle_lib):
    """64 area lights and a directional one (641 rays per hit: the ray -> light table instead of the search)."""
    sc = synth.make_scene(20000, seed=0xC5, n_area_lights=64)
    upload(gpu, sc)
    compare(gpu, sc, oracle_lib.Oracle(sc), n=600, seed=41, area_samples=10)
