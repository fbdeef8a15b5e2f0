"""Closed-form checks of the HIP path that do not involve the oracle: texture lookup, cubemap face
choice and bilinear fetch, and the rectangular area light's sample geometry, each against a value
worked out by hand from the reference's source (the same cases tests/test_oracle_kat.py and
tests/test_textures.py hold for the oracle).  The reference ships inputs for these paths
(customNormalMap.json, the cubemaps) but no outputs: parity unpinned beyond these and the
lowroughnessHighMetallic screenshot."""
import numpy as np
import pytest

from path_tracer_amd import abi
from tests import util as U
from tests.test_oracle_kat import py_mix, py_u01
from tests.test_textures import bilinear, checker

pytestmark = pytest.mark.gpu


def test_square_texture_lookup_closed_form(gpu):
    # Square.cpp:41: uv = hit + 0.5; TextureMap::getMappedValue (material.cpp:186-222): bilinear over (w-1, h-1)
    tex = checker()
    sc = U.scene([(abi.OBJ_SQUARE, np.eye(4), U.material())], cam=U.camera())
    sc.textures = [tex]
    sc.materials[0].kd_texture = 1
    gpu.upload_scene(sc)
    pts = np.array([[-0.5, -0.5], [0.5, 0.5], [0.123, -0.321], [0.0, 0.0], [0.499, -0.499], [-0.37, 0.41]])
    org = np.column_stack([pts, np.full(len(pts), 3.0)])
    d = np.repeat([[0.0, 0.0, -1.0]], len(pts), 0)
    uv, kd, ka = gpu.hit_material(org, d)
    assert np.allclose(uv, pts + 0.5, atol=1e-15)
    for k in range(len(pts)):
        assert np.allclose(kd[k], bilinear(tex, uv[k, 0], uv[k, 1]), atol=1e-14)
    assert np.allclose(kd[1], tex[-1, -1] / 255.0, atol=1e-14)   # u = v = 1: the texels beyond the edge read black with weight 0
    assert np.all(ka == 0)


def test_cubemap_faces_closed_form(gpu):
    # cubeMap.cpp:7-64: the face is chosen by the dominant axis after dir.z is negated; an empty scene shows the cubemap alone
    faces = []
    for k in range(6):
        img = np.zeros((4, 4, 3), np.uint8)
        img[..., 0] = 40 * k + 10
        img[..., 1] = (np.arange(4) * 60)[None, :]   # varies with x
        img[..., 2] = (np.arange(4) * 50)[:, None]   # varies with y (row 0 = bottom)
        faces.append(img)
    for viewdir, up, face in (((1, 0, 0), (0, 1, 0), 0), ((-1, 0, 0), (0, 1, 0), 1), ((0, 1, 0), (0, 0, 1), 2), ((0, -1, 0), (0, 0, 1), 3),
                              ((0, 0, -1), (0, 1, 0), 4), ((0, 0, 1), (0, 1, 0), 5)):
        sc = U.scene([], cam=U.camera(eye=(0, 0, 0), viewdir=viewdir, updir=up), cubemap=faces)
        gpu.upload_scene(sc)
        # a 2 x 2 frame: pixel (1, 1) is the ray through x = y = 0.5, the view direction itself (RayTracer.cpp:70-71)
        f32, _ = gpu.render(abi.default_params(2, 2, spp=1), want_rgb8=False)
        centre = f32[1, 1].astype(np.float64) * 255
        assert np.allclose(centre[0], 40 * face + 10, atol=1e-4), (viewdir, centre)
        # the centre of a face is coord (0.5, 0.5) -> x = y = 1.5: halfway between texels 1 and 2
        assert np.allclose(centre[1], 90.0, atol=1e-4) and np.allclose(centre[2], 75.0, atol=1e-4)


def test_area_light_samples_closed_form(gpu):
    # samplePoint (light.cpp:105-111): corner + uVec*uI + vVec + vI, vI on every component; attenuation
    # min(1, 1/(a + b d + c d^2)) with float coefficients (light.cpp:139-142); nothing in the way
    L = U.area_light((0, 1, -15), (0, 10, 0), (0, 15, 10), 5.0, 5.0, color=(0.5, 1.0, 0.25), atten=(0.0, 0.0, 0.0008))
    sc = U.scene([(abi.OBJ_SPHERE, U.translate(500, 0, 0), U.material())], lights=[L])
    gpu.upload_scene(sc)
    fire = np.array([[1.0, 2.0, 3.0], [-4.0, 0.5, 2.0]])
    keys = np.array([99, 12345678901234567], dtype=np.uint64)
    got = gpu.shadow_attenuation(abi.default_params(8, 8), 0, fire, keys)
    for k in range(2):
        acc = 0.0
        for s in range(10):
            uI = 5.0 * py_u01(py_mix(int(keys[k]), 16 + 2 * s))
            vI = 5.0 * py_u01(py_mix(int(keys[k]), 16 + 2 * s + 1))
            pos = np.array([0, 1, -15.0]) + np.array([0, 10, 0.0]) * uI + np.array([0, 15, 10.0]) + vI
            dist = np.linalg.norm(pos - fire[k])
            acc += min(1.0, 1 / (np.float32(0.0008) * dist * dist))
        assert np.allclose(got[k], np.array([0.5, 1.0, 0.25]) * acc / 10, rtol=1e-12)
