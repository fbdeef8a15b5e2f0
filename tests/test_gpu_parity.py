"""GPU parity: the HIP path through the C ABI (libmipt.so) against the CPU oracle on the
same seeded inputs, plus size-independent properties at the benchmark's full frame size.

Stated tolerances.  Everything on the GPU path that decides a hit is fp64, so per-ray
results agree to rounding: |dt| <= 1e-9 * max(1, t), normals to 1e-9, the same primitive
on every ray.  Images (mean of per-sample-clamped radiance, float32): |d| <= 1e-5 per
channel on >= 99.8 % of the pixels (a ray that grazes an edge may fall either way after a
different rounding, which changes that pixel), RGB8 within 1 LSB on those pixels.
"""
import ctypes as C
import math

import numpy as np
import pytest

from path_tracer_amd import abi, synth
from tests import util as U
from tests.conftest import SCENE_NAMES

pytestmark = pytest.mark.gpu

PIX_TOL = 1e-5
PIX_FRAC = 0.998


def upload(gpu, sc):
    gpu.upload_scene(sc)
    return gpu


def assert_images_match(f32, ref, rgb8=None, ref8=None, frac=PIX_FRAC):
    diff = np.abs(f32.astype(np.float64) - ref).max(axis=2)
    ok = diff <= PIX_TOL
    assert ok.mean() >= frac, f"only {ok.mean():.5f} of the pixels within {PIX_TOL} (max {diff.max():.3e})"
    if rgb8 is not None:
        d8 = np.abs(rgb8.astype(np.int32) - ref8.astype(np.int32)).max(axis=2)
        assert (d8[ok] <= 1).all()


def assert_rays_match(gpu_res, orc_res, min_same=1.0):
    tg, pg, ng = gpu_res
    to, po, no = orc_res
    same = pg == po
    assert same.mean() >= min_same, f"{(~same).sum()} of {len(same)} rays found a different primitive"
    hit = same & (po >= 0)
    assert np.all(np.abs(tg[hit] - to[hit]) <= 1e-9 * np.maximum(1.0, to[hit]))
    assert np.all(np.abs(ng[hit] - no[hit]) <= 1e-9)
    assert np.all(tg[same & (po < 0)] == to[same & (po < 0)])


# -------------------------------------------------------------------- per ray
@pytest.mark.parametrize("name", SCENE_NAMES)
def test_closest_hit_matches_the_oracle_per_ray(name, gpu, scenes, oracles):
    sc, orc = scenes(name), oracles(name)
    upload(gpu, sc)
    org, d = U.camera_rays(sc, 20000, seed=1)
    res = orc.intersect(org, d)
    assert_rays_match(gpu.intersect(org, d), res)
    t, prim, n = res
    hit = prim >= 0
    P = org[hit] + t[hit, None] * d[hit]
    # bounce-style rays: origin eps along the direction (RayTracer.cpp:259)
    d2 = U.random_dirs(len(P), seed=2)
    assert_rays_match(gpu.intersect(P + d2 * 1e-8, d2), orc.intersect(P + d2 * 1e-8, d2))
    # mirror-style rays: origin eps along the normal (RayTracer.cpp:270-271)
    r = d[hit] - n[hit] * (2.0 * np.sum(n[hit] * d[hit], axis=1))[:, None]
    assert_rays_match(gpu.intersect(P + n[hit] * 1e-8, r), orc.intersect(P + n[hit] * 1e-8, r))


def test_closest_hit_on_a_triangle_mesh_scene(gpu, oracle_lib):
    from path_tracer_amd import synth

    sc = synth.make_scene(20000, seed=7)
    orc = oracle_lib.Oracle(sc)
    upload(gpu, sc)
    org, d = U.camera_rays(sc, 20000, seed=3)
    res = orc.intersect(org, d)
    assert_rays_match(gpu.intersect(org, d), res, min_same=0.9999)
    t, prim, n = res
    hit = prim >= 0
    assert hit.mean() > 0.5
    P = org[hit] + t[hit, None] * d[hit]
    d2 = U.random_dirs(len(P), seed=4)
    assert_rays_match(gpu.intersect(P + d2 * 1e-8, d2), orc.intersect(P + d2 * 1e-8, d2), min_same=0.9999)
    r = d[hit] - n[hit] * (2.0 * np.sum(n[hit] * d[hit], axis=1))[:, None]
    assert_rays_match(gpu.intersect(P + n[hit] * 1e-8, r), orc.intersect(P + n[hit] * 1e-8, r), min_same=0.9999)


def test_traversal_stack_overflow_area(gpu, oracle_lib, monkeypatch):
    # with two LDS entries per lane nearly every walk keeps the rest of its stack in the
    # global-memory overflow area: closest hits, any-hit shadow rays and images must not change
    from path_tracer_amd import synth

    sc = synth.make_scene(20000, seed=11)
    orc = oracle_lib.Oracle(sc)
    monkeypatch.setenv("MIPT_LDS_STACK", "2")
    upload(gpu, sc)
    monkeypatch.delenv("MIPT_LDS_STACK")
    org, d = U.camera_rays(sc, 20000, seed=5)
    res = orc.intersect(org, d)
    assert_rays_match(gpu.intersect(org, d), res, min_same=0.9999)
    p = abi.default_params(96, 54, spp=2, seed=3)
    f32, rgb8 = gpu.render(p)
    ref, ref8 = orc.render(p, threads=8, want_rgb8=True)
    assert_images_match(f32, ref, rgb8, ref8)


def test_gpu_built_bvh_gives_the_same_results(gpu, oracle_lib, monkeypatch):
    # MIPT_BUILDER=gpu builds the tree on the device (Morton order, nearest-neighbour clustering, four-wide
    # collapse); the box tests only cull, so hits and images must not depend on which builder made the tree
    from path_tracer_amd import synth

    sc = synth.make_scene(30000, seed=21)
    orc = oracle_lib.Oracle(sc)
    monkeypatch.setenv("MIPT_BUILDER", "gpu")
    upload(gpu, sc)
    monkeypatch.delenv("MIPT_BUILDER")
    stats = gpu.scene_stats()
    assert stats["nodes"] > 1000
    org, d = U.camera_rays(sc, 30000, seed=9)
    res = orc.intersect(org, d)
    assert_rays_match(gpu.intersect(org, d), res, min_same=0.9999)
    t, prim, n = res
    hit = prim >= 0
    P = org[hit] + t[hit, None] * d[hit]
    d2 = U.random_dirs(len(P), seed=10)
    assert_rays_match(gpu.intersect(P + d2 * 1e-8, d2), orc.intersect(P + d2 * 1e-8, d2), min_same=0.9999)
    p = abi.default_params(96, 54, spp=2, seed=4)
    f32, rgb8 = gpu.render(p)
    ref, ref8 = orc.render(p, threads=8, want_rgb8=True)
    assert_images_match(f32, ref, rgb8, ref8)


def test_gpu_built_bvh_is_nearly_as_good_as_the_host_built_one(gpu, monkeypatch):
    """The device builder's trees must stay close to the host's binned-SAH trees in what a render
    costs: node visits and primitive tests of the same frame within 1.5x (measured 1.3x / 1.0x at
    200 k triangles; a radix tree over the Morton codes, the earlier device builder, needed 2.7x)."""
    from path_tracer_amd import synth

    sc = synth.make_scene(200_000)
    work = {}
    for builder in ("host", "gpu"):
        monkeypatch.setenv("MIPT_BUILDER", builder)
        upload(gpu, sc)
        gpu.reset_counters()
        gpu.render(abi.default_params(480, 270, spp=2, flags=abi.FLAG_COUNTERS), want_f32=False, want_rgb8=False)
        work[builder] = gpu.counters()
    monkeypatch.delenv("MIPT_BUILDER")
    assert work["gpu"]["path_rays"] == work["host"]["path_rays"] and work["gpu"]["shadow_rays"] == work["host"]["shadow_rays"]
    assert work["gpu"]["node_visits"] <= 1.5 * work["host"]["node_visits"]
    assert work["gpu"]["prim_tests"] <= 1.5 * work["host"]["prim_tests"]


def test_transformed_mesh_matches_local_space_reference(gpu, oracle_lib):
    # the library pre-transforms triangles to world space; the reference intersects in the
    # mesh's local space (scene.cpp:13-40).  Non-uniform scale + rotation + mirroring.
    nrm = np.array([[1, 0, 1], [0, 0, 1], [0, 0, 1], [0, 1, 1]], float)
    xf = U.translate(0.3, -0.2, -4) @ U.rot_y(0.7) @ U.rot_x(-0.4) @ U.scale(2.0, -0.7, 1.3)
    sc = U.scene([(abi.OBJ_TRIMESH, xf, U.material(), 0), (abi.OBJ_TRIMESH, U.translate(0, 0, -6) @ U.scale(3), U.material(), 1)],
                 meshes=[U.quad_mesh(nrm), U.quad_mesh()], cam=U.camera(eye=(0, 0, 3)))
    orc = oracle_lib.Oracle(sc)
    upload(gpu, sc)
    org, d = U.camera_rays(sc, 20000, seed=8)
    res = orc.intersect(org, d)
    assert (res[1] >= 0).mean() > 0.3
    assert_rays_match(gpu.intersect(org, d), res, min_same=0.9999)


def test_coincident_spheres_resolve_like_the_reference(gpu, scenes, oracles):
    # the six coincident pairs of spheres.json differ in roughness: picking the other one changes the image
    sc, orc = scenes("spheres"), oracles("spheres")
    upload(gpu, sc)
    centres = np.array([o.xform[:3, 3] for o in sc.objects if o.type == abi.OBJ_SPHERE])
    uniq, counts = np.unique(np.round(centres, 9), axis=0, return_counts=True)
    pairs = uniq[counts == 2]
    assert len(pairs) == 6
    eye = np.asarray(sc.camera.eye)
    d = pairs - eye
    d /= np.linalg.norm(d, axis=1)[:, None]
    tg, pg, _ = gpu.intersect(np.repeat(eye[None], 6, 0), d)
    to, po, _ = orc.intersect(np.repeat(eye[None], 6, 0), d)
    assert np.array_equal(pg, po) and np.allclose(tg, to, rtol=1e-12, atol=0)


@pytest.mark.parametrize("name", ["spheres", "cornellBoxes", "hitchcockBRDF", "singleSphere", "spheresDirectional"])
def test_shadow_attenuation_matches_per_light(name, gpu, scenes, oracles):
    sc, orc = scenes(name), oracles(name)
    upload(gpu, sc)
    org, d = U.camera_rays(sc, 4000, seed=5)
    t, prim, n = orc.intersect(org, d)
    hit = prim >= 0
    fire = org[hit] + t[hit, None] * d[hit] + n[hit] * 1e-8 * 3.0      # material.cpp:117
    keys = np.arange(len(fire), dtype=np.uint64) * np.uint64(7919) + np.uint64(13)
    for flags in (0, abi.FLAG_FIXED_AREA_PATTERN):
        p = abi.default_params(8, 8, flags=flags)
        for li in range(len(sc.lights)):
            a = gpu.shadow_attenuation(p, li, fire, keys)
            b = orc.shadow_attenuation(p, li, fire, keys)
            bad = np.abs(a - b).max(axis=1) > 1e-9
            assert bad.mean() <= 2e-3, f"light {li}: {bad.sum()} of {len(fire)} shading points differ"


def test_translucent_shadow_walk(gpu, oracle_lib):
    glass = U.material(kd=(0.2, 0.2, 0.2), kt=(0.5, 0.25, 0.9))
    floor = U.material(kd=(0.8, 0.8, 0.8))
    sc = U.scene([(abi.OBJ_BOX, U.translate(0, 2, 0) @ U.scale(3, 1, 3), glass), (abi.OBJ_SPHERE, U.translate(2.5, 2, 0.5), glass),
                  (abi.OBJ_SQUARE, U.rot_x(-math.pi / 2) @ U.scale(30), floor)],
                 lights=[U.point_light((0.5, 9, 0.5), atten=(1, 0, 0)), U.dir_light((0.1, -1, 0.2)), U.area_light((-1, 8, -1), (1, 0, 0), (0, 0, 1), 2, 2)],
                 cam=U.camera(eye=(0, 6, 12), viewdir=(0, -0.4, -1)))
    orc = oracle_lib.Oracle(sc)
    upload(gpu, sc)
    xs, zs = np.meshgrid(np.linspace(-5, 5, 40), np.linspace(-5, 5, 40))
    fire = np.stack([xs.ravel(), np.full(xs.size, 3e-8), zs.ravel()], 1)
    keys = np.arange(len(fire), dtype=np.uint64)
    p = abi.default_params(8, 8)
    for li in range(3):
        a, b = gpu.shadow_attenuation(p, li, fire, keys), orc.shadow_attenuation(p, li, fire, keys)
        assert ((b > 0) & (b < 0.99)).any()                       # some rays really pass through glass
        bad = np.abs(a - b).max(axis=1) > 1e-9
        assert bad.mean() <= 5e-3
    pr = abi.default_params(48, 48, spp=2)
    f32, rgb8 = gpu.render(pr)
    ref, ref8 = orc.render(pr, threads=4, want_rgb8=True)
    assert_images_match(f32, ref, rgb8, ref8)


# ---------------------------------------------------------------------- images
@pytest.mark.parametrize("name", SCENE_NAMES)
def test_deterministic_one_spp_image(name, gpu, scenes, oracles):
    """Fixed seeds, Russian roulette off, depth cap 4, fixed area-light pattern."""
    sc, orc = scenes(name), oracles(name)
    upload(gpu, sc)
    w, h = (256, 144) if name == "hitchcockBRDF" else (160, 160)
    p = abi.default_params(w, h, spp=1, rr_q=0.0, max_depth=4, flags=abi.FLAG_FIXED_AREA_PATTERN)
    f32, rgb8 = gpu.render(p)
    ref, ref8 = orc.render(p, threads=8, want_rgb8=True)
    assert_images_match(f32, ref, rgb8, ref8)


@pytest.mark.parametrize("name", ["spheres", "cornellBoxes", "hitchcockBRDF", "areaLight"])
def test_stochastic_image_same_random_streams(name, gpu, scenes, oracles):
    """Reference settings (q = 0.1, random area samples): both sides draw the same counter-based
    numbers, so the images agree pixel by pixel, not just statistically."""
    sc, orc = scenes(name), oracles(name)
    upload(gpu, sc)
    p = abi.default_params(96, 96, spp=8)
    f32, rgb8 = gpu.render(p)
    ref, ref8, cnt = orc.render(p, threads=8, want_rgb8=True, want_counters=True)
    assert_images_match(f32, ref, rgb8, ref8)
    # the same path trees: identical ray and hit counts
    gpu.reset_counters()
    pc = abi.default_params(96, 96, spp=8, flags=abi.FLAG_COUNTERS)
    gpu.render(pc, want_f32=False, want_rgb8=False)
    c = gpu.counters()
    assert c["samples"] == cnt["samples"] == 96 * 96 * 8
    for k in ("path_rays", "shaded_hits"):
        assert abs(c[k] - cnt[k]) <= 2e-4 * cnt[k] + 2, (k, c[k], cnt[k])


def test_mesh_scene_images(gpu, oracle_lib):
    from path_tracer_amd import synth

    sc = synth.make_scene(20000, seed=7)
    orc = oracle_lib.Oracle(sc)
    upload(gpu, sc)
    p = abi.default_params(128, 72, spp=1, rr_q=0.0, max_depth=4, flags=abi.FLAG_FIXED_AREA_PATTERN)
    f32, rgb8 = gpu.render(p)
    ref, ref8 = orc.render(p, threads=8, want_rgb8=True)
    assert ref.mean() > 0.02
    assert_images_match(f32, ref, rgb8, ref8)
    p = abi.default_params(96, 54, spp=2)
    f32, rgb8 = gpu.render(p)
    ref, ref8 = orc.render(p, threads=8, want_rgb8=True)
    assert_images_match(f32, ref, rgb8, ref8)


def test_shadow_rays_into_their_own_surface(gpu, oracle_lib):
    """shadeBRDF fires a shadow ray toward every light, also one behind the surface (material.cpp:117-126);
    such a ray starts 3e-8 above the triangle it leaves and crosses it at once.  The shadow kernel ends it
    with a test of that triangle instead of a walk.  Lights below the terrain and inside the blob layer:
    most shadow rays of mesh hits are of that kind.  The frames must stay the oracle's (flat and smooth
    normals, hits next to edges included), and the walks must be gone from the counters."""
    import copy

    from path_tracer_amd import synth

    base = synth.make_scene(20000, seed=7)
    low = copy.deepcopy(base)
    for L in low.lights:
        if L.type == abi.LIGHT_RECT_AREA:
            L.position = np.array([-6.0, -30.0, -40.0])  # below terrain and floor: every mesh hit seen from above faces away
        else:
            L.direction = np.array([0.3, 1.0, 0.2])      # shines upward
    mid = copy.deepcopy(base)
    for L in mid.lights:
        if L.type == abi.LIGHT_RECT_AREA:
            L.position = np.array([-6.0, 1.0, -40.0])    # among the blobs: grazing and back-facing hits mixed
    per_ray = {}
    for name, sc in (("base", base), ("low", low), ("mid", mid)):
        upload(gpu, sc)
        orc = oracle_lib.Oracle(sc)
        p = abi.default_params(128, 72, spp=2)
        f32, rgb8 = gpu.render(p)
        ref, ref8 = orc.render(p, threads=8, want_rgb8=True)
        assert_images_match(f32, ref, rgb8, ref8)
        gpu.reset_counters()
        pc = abi.default_params(128, 72, spp=2, flags=abi.FLAG_COUNTERS)
        f32c, _ = gpu.render(pc, want_rgb8=False)
        assert np.array_equal(f32c, f32)
        c = gpu.counters()
        assert c["shadow_rays"] > 0
        per_ray[name] = c["shadow_node_visits"] / c["shadow_rays"]
    # lights behind every mesh surface: what is left are the walks from the analytic floor and the few
    # rays their own triangle does not block
    assert per_ray["low"] < 0.25 * per_ray["base"], per_ray


def test_cull_negligible_flag_skips_rays_without_changing_the_image(gpu, oracle_lib):
    """MIPT_FLAG_CULL_NEGLIGIBLE drops the shadow rays of hits on "metallic 1" materials (diffuse
    weight 1.1e-16 after material.cpp:232-241) and nothing else: fewer shadow rays, the same
    path trees, an image within 1e-14 of the full one and still on the oracle's."""
    from path_tracer_amd import synth

    sc = synth.make_scene(20000, seed=7)
    upload(gpu, sc)
    imgs, cnts = [], []
    for extra in (0, abi.FLAG_CULL_NEGLIGIBLE):
        gpu.reset_counters()
        p = abi.default_params(128, 72, spp=4, flags=abi.FLAG_COUNTERS | extra)
        f32, _ = gpu.render(p, want_rgb8=False)
        imgs.append(f32.astype(np.float64))
        cnts.append(gpu.counters())
    full, culled = cnts
    assert culled["shadow_rays"] < full["shadow_rays"]
    assert culled["shadow_rays"] > 0.5 * full["shadow_rays"]
    for k in ("samples", "path_rays", "shaded_hits"):
        assert culled[k] == full[k]
    assert np.abs(imgs[0] - imgs[1]).max() <= 1e-7  # fp32 output: at most one rounding apart
    assert (imgs[0] != imgs[1]).mean() < 1e-3
    ref = oracle_lib.Oracle(sc).render(abi.default_params(128, 72, spp=4), threads=8)
    assert np.abs(imgs[1] - ref).max() <= 1e-5


@pytest.mark.parametrize("variant", ["cornell_area_light", "hitchcock_tessellated"])
def test_authored_variants_of_the_demo_scenes(variant, gpu, oracle_lib, scenes):
    """cornellBoxes under a rectangular area light (10 shadow rays per hit) and hitchcockBRDF
    with every box as 12 triangles: per-ray hits, a deterministic frame and a frame with the
    reference's settings."""
    from path_tracer_amd import synth

    sc = synth.cornell_area_light(scenes("cornellBoxes")) if variant == "cornell_area_light" else synth.tessellate_boxes(scenes("hitchcockBRDF"))
    orc = oracle_lib.Oracle(sc)
    upload(gpu, sc)
    org, d = U.camera_rays(sc, 20000, seed=5)
    assert_rays_match(gpu.intersect(org, d), orc.intersect(org, d))
    p = abi.default_params(128, 96, spp=1, rr_q=0.0, max_depth=4, flags=abi.FLAG_FIXED_AREA_PATTERN)
    f32, rgb8 = gpu.render(p)
    ref, ref8 = orc.render(p, threads=8, want_rgb8=True)
    assert ref.mean() > 0.01
    assert_images_match(f32, ref, rgb8, ref8)
    p = abi.default_params(96, 96, spp=8)
    f32, rgb8 = gpu.render(p)
    ref, ref8, cnt = orc.render(p, threads=8, want_rgb8=True, want_counters=True)
    assert_images_match(f32, ref, rgb8, ref8)
    if variant == "cornell_area_light":
        assert cnt["shadow_rays"] == 10 * cnt["shaded_hits"]


def test_cubemap_environment(gpu, oracle_lib, scenes):
    from path_tracer_amd import synth
    import copy

    sc = copy.copy(scenes("singleSphere"))
    sc.cubemap = synth.make_sky_cubemap(32)
    orc = oracle_lib.Oracle(sc)
    upload(gpu, sc)
    p = abi.default_params(96, 96, spp=4)
    f32, rgb8 = gpu.render(p)
    ref, ref8 = orc.render(p, threads=8, want_rgb8=True)
    assert ref.mean() > 0.2                                           # the sky is visible
    assert_images_match(f32, ref, rgb8, ref8)


def test_converged_render_is_within_the_noise_floor(gpu, scenes, oracles):
    """Different seeds on the two sides: agreement must be statistical.  RMSD is raycheck.py's
    (||a-b||_2 / sqrt(size) on float [0,1] RGB); the bound is 1.5x the RMSD between two oracle
    renders with different seeds, and 0.02 absolute at 256 spp."""
    sc, orc = scenes("areaLight"), oracles("areaLight")
    upload(gpu, sc)
    spp = 256
    f32, _ = gpu.render(abi.default_params(48, 48, spp=spp, seed=5))
    a = orc.render(abi.default_params(48, 48, spp=spp, seed=6), threads=8)
    b = orc.render(abi.default_params(48, 48, spp=spp, seed=7), threads=8)
    floor = U.rmsd(a, b)
    got = U.rmsd(f32, a)
    assert got <= 1.5 * floor + 1e-4 and got <= 0.02, (got, floor)


@pytest.mark.parametrize("name,w,h", [("spheres", 64, 64), ("cornellBoxes", 64, 64), ("hitchcockBRDF", 64, 36)])
def test_converged_render_at_the_stated_tolerance(gpu, scenes, oracles, name, w, h):
    """The converged criterion on the three demo-scene configurations: 1024 samples per pixel, another seed on each
    side, RMSD (raycheck.py:316-324: ||a - b||_2 / sqrt(size) on float [0, 1] RGB) between the GPU frame and an
    oracle frame at most 1.5 x the RMSD between two oracle frames of different seeds, and at most 0.01."""
    sc, orc = scenes(name), oracles(name)
    upload(gpu, sc)
    spp = 1024
    f32, _ = gpu.render(abi.default_params(w, h, spp=spp, seed=303), want_rgb8=False)
    a = orc.render(abi.default_params(w, h, spp=spp, seed=101), threads=16)
    b = orc.render(abi.default_params(w, h, spp=spp, seed=202), threads=16)
    floor = U.rmsd(a, b)
    got, got_b = U.rmsd(f32, a), U.rmsd(f32, b)
    assert 0 < floor <= 0.01, floor
    assert max(got, got_b) <= 1.5 * floor and max(got, got_b) <= 0.01, (got, got_b, floor)
    assert abs(float(f32.mean()) - float(a.mean())) <= 0.01 * float(a.mean()) + 1e-4   # no brightness offset hiding inside the noise


# ------------------------------------------------------------- host interface
def test_trace_pixel_is_the_rendered_pixel(gpu, scenes):
    upload(gpu, scenes("cornellBoxes"))
    p = abi.default_params(40, 40, spp=6)
    f32, _ = gpu.render(p)
    for (i, j) in ((0, 0), (13, 27), (39, 39)):
        px = gpu.trace_pixel(p, i, j)
        assert np.allclose(px, f32[j, i], rtol=0, atol=2e-7)


def test_raytracer_shim_follows_the_reference_call_order(scenes):
    from path_tracer_amd.raytracer import RayTracer

    rt = RayTracer(0)
    assert not rt.sceneLoaded() and np.array_equal(rt.tracePixel(0, 0), np.zeros(3))     # RayTracer.cpp:63-64
    assert not rt.loadScene("/nonexistent/scene.json") and rt.last_alert
    assert rt.loadScene(scenes("spheres"))
    rt.overrides = {"spp": 4}
    rt.traceSetup(32, 32)
    assert rt.getBuffer()[0].sum() == 0                                                   # zero-filled (:427)
    rt.traceImage(32, 32)
    rt.waitRender()
    buf, w, h = rt.getBuffer()
    assert rt.checkRender() and (w, h) == (32, 32) and buf.shape == (32, 32, 3) and buf.dtype == np.uint8
    assert buf[-4:].sum() == 0 and buf[:16].sum() > 0                                     # row 0 = bottom; sky on top
    c = rt.tracePixel(10, 5)
    assert np.array_equal(buf[5, 10], (255.0 * c).astype(np.int64).astype(np.uint8))      # truncation (:97-99)


def test_sharding_is_exact_on_the_gpu(gpu, scenes):
    import torch

    upload(gpu, scenes("spheres"))
    w, h = 64, 48
    full = torch.zeros(3 * w * h, dtype=torch.float64, device="cuda")
    gpu.render_accumulate(abi.default_params(w, h, spp=4, band_rows=4), full.data_ptr())
    tiles = torch.zeros_like(full)
    for r in range(3):
        gpu.render_accumulate(abi.default_params(w, h, spp=4, band_rows=4, shard_index=r, shard_count=3), tiles.data_ptr())
    samples = torch.zeros_like(full)
    for (a, b) in ((0, 1), (1, 4)):
        gpu.render_accumulate(abi.default_params(w, h, spp=4, band_rows=4, sample_begin=a, sample_end=b), samples.data_ptr())
    batched = torch.zeros_like(full)
    gpu.render_accumulate(abi.default_params(w, h, spp=4, band_rows=4, batch_samples=1000), batched.data_ptr())
    torch.cuda.synchronize()
    for other in (tiles, samples, batched):
        assert float((other - full).abs().max()) <= 1e-12   # same samples, same streams; only fp64 add order differs
    f = torch.zeros(3 * w * h, dtype=torch.float32, device="cuda")
    b8 = torch.zeros(3 * w * h, dtype=torch.uint8, device="cuda")
    gpu.resolve(full.data_ptr(), w, h, 4, f.data_ptr(), b8.data_ptr())
    torch.cuda.synchronize()
    mean = (full / 4).cpu().numpy()
    assert np.array_equal(f.cpu().numpy(), mean.astype(np.float32))
    assert np.array_equal(b8.cpu().numpy(), (255.0 * mean).astype(np.int64).astype(np.uint8))


def test_growing_path_trees_spill_to_the_overflow_stack(gpu, oracle_lib):
    # two facing mirrors: every hit spawns a bounce and a mirror ray (RayTracer.cpp:262,273),
    # the tree doubles per level until the depth cap
    mirror = U.material(kd=(0.3, 0.3, 0.3), ka=(0.1, 0.1, 0.1), roughness=0.0)
    sc = U.scene([(abi.OBJ_SQUARE, U.translate(0, 0, -2) @ U.scale(400), mirror), (abi.OBJ_SQUARE, U.translate(0, 0, 2) @ U.scale(400), mirror)],
                 lights=[U.point_light((0, 0.5, 0), atten=(1, 0, 0))], cam=U.camera(eye=(0, 0, 0)), ambient=(1, 1, 1))
    orc = oracle_lib.Oracle(sc)
    upload(gpu, sc)
    p = abi.default_params(16, 16, spp=2, max_depth=12, batch_samples=128)
    f32, _ = gpu.render(p, want_rgb8=False)
    ref, cnt = orc.render(p, threads=8, want_counters=True)
    assert cnt["path_rays"] > 40 * 16 * 16 * 2                        # the tree really grew
    assert_images_match(f32, ref)


def test_options_choose_code_paths_not_results(gpu):
    """mipt_set_option: the device builder, a two-entry LDS stack, fp64 shadow records and the analytic tests inside the
    walk give the frame the defaults give (to the rounding of sums made in another order)."""
    from path_tracer_amd.raytracer import MiptError

    sc = _mesh_scene_with_analytic_occluders()
    p = abi.default_params(80, 60, spp=3, seed=9)
    upload(gpu, sc)
    base = gpu.render(p, want_rgb8=False)[0].astype(np.float64)
    assert base.mean() > 0.02
    choices = {"builder": 1, "lds_stack": 2, "vis_bytes": 0, "lean_any": 0, "fuse_rays": 0, "overlap_rays": 0}
    try:
        for name, value in choices.items():
            gpu.set_option(name, value)
        upload(gpu, sc)
        other = gpu.render(p, want_rgb8=False)[0].astype(np.float64)
        assert np.abs(other - base).max() <= 1e-6
        with pytest.raises(MiptError):
            gpu.set_option("no_such_option", 1)
        with pytest.raises(MiptError):
            gpu.set_option("lds_stack", 1000)
    finally:
        for name, value in {"builder": 0, "lds_stack": 24, "vis_bytes": -1, "lean_any": -1, "fuse_rays": -1, "overlap_rays": 1 << 18}.items():
            gpu.set_option(name, value)


def test_tail_launch_output_larger_than_a_batch_is_parked(gpu, oracle_lib):
    # The launch that finishes small levels depth first hands back what its waves could not finish: close to five rays
    # per input ray when every hit spawns two (facing mirrors, no Russian roulette).  With 6300 camera rays in the last
    # batch of 16384 that is more than one launch takes; the excess goes to the overflow stack like any other level's.
    import torch

    mirror = U.material(kd=(0.3, 0.3, 0.3), ka=(0.1, 0.1, 0.1), roughness=0.0)
    sc = U.scene([(abi.OBJ_SQUARE, U.translate(0, 0, -2) @ U.scale(400), mirror), (abi.OBJ_SQUARE, U.translate(0, 0, 2) @ U.scale(400), mirror)],
                 lights=[U.point_light((0, 0.5, 0), atten=(1, 0, 0))], cam=U.camera(eye=(0, 0, 0)), ambient=(1, 1, 1))
    orc = oracle_lib.Oracle(sc)
    upload(gpu, sc)
    w, h = 107, 212                                                    # 22 684 pixels = one batch of 16384 and one of 6300
    p = abi.default_params(w, h, spp=1, max_depth=10, batch_samples=16384)
    p.rr_q = 0.0
    f32, _ = gpu.render(p, want_rgb8=False)
    ref, cnt = orc.render(p, threads=16, want_counters=True)
    assert cnt["path_rays"] > 500 * w * h                              # every tree doubled down to the depth cap
    assert_images_match(f32, ref)
    # the entry point that cannot allocate runs batches of that size by itself: the same frame, or MIPT_ERR_LIMIT when its
    # fixed overflow stack is too small for the tree, never a queue overflow
    from path_tracer_amd.raytracer import MiptError
    d = torch.zeros(3 * w * h, dtype=torch.float64, device="cuda")
    try:
        gpu.render_accumulate_static(p, d.data_ptr())
        torch.cuda.synchronize()
        assert np.abs(d.cpu().numpy().reshape(h, w, 3) - ref).max() <= 1e-5
    except MiptError as e:
        assert e.code == abi.ERR_LIMIT, e


def test_finishing_launch_whose_waves_hand_rays_back(oracle_lib, monkeypatch):
    """The launch that finishes a mesh scene's narrow levels hands its input out in chunks, lets its lanes walk on their
    own and exports its shaded hits; a wave whose share of the list is used up stops and hands back its stack, its lanes'
    rays (walked or not) and, once every wave has stopped, the input nobody took.  With the share cut to 64 and to 256
    records the launch is repeated many times over what comes back; the frame is the one the plain level loop gives
    (the same samples), which is the oracle's."""
    from path_tracer_amd.raytracer import Context

    sc = synth.make_scene(20000, seed=3)
    orc = oracle_lib.Oracle(sc)
    p = abi.default_params(96, 64, spp=4, seed=33)
    ref = orc.render(p, threads=16)
    frames = []
    for fuse, budget in (("0", None), ("1048576", "1"), ("1048576", "4"), ("1048576", None)):
        monkeypatch.setenv("MIPT_FUSE_RAYS", fuse)
        if budget is None:
            monkeypatch.delenv("MIPT_FUSE_BUDGET", raising=False)
        else:
            monkeypatch.setenv("MIPT_FUSE_BUDGET", budget)
        ctx = Context(0)
        try:
            ctx.upload_scene(sc)
            ctx.reset_counters()
            pk = abi.default_params(96, 64, spp=4, seed=33, flags=abi.FLAG_PROFILE)
            frames.append(ctx.render(pk, want_rgb8=False)[0].astype(np.float64))
            launches = ctx.kernel_times()["tail"]["launches"]
            if fuse == "0":
                assert launches == 0
            elif budget is not None:
                assert launches >= 3, launches                             # stopped waves: the launch ran again on what came back
            else:
                assert launches >= 1
        finally:
            ctx.close()
    assert_images_match(frames[0], ref)
    for f in frames[1:]:
        assert np.abs(f - frames[0]).max() <= 1e-6


def test_two_launch_streams_give_the_one_stream_frame(scenes, monkeypatch):
    # narrow levels put their shadow rays on a second stream beside the next level's closest-hit
    # launch; MIPT_OVERLAP=0 keeps everything on the caller's stream.  Same rays, same sums up to
    # the order of the atomic adds into a sample's accumulator.
    from path_tracer_amd.raytracer import Context
    frames = {}
    # one stream; every level that fits a half list on two; narrow levels only; the same without the fused tail launch of mesh scenes
    for knob, rays, fuse in (("0", "0", "0"), ("1", "1073741824", "1048576"), ("1", "4096", "4096"), ("1", "4096", "0")):
        monkeypatch.setenv("MIPT_OVERLAP", knob)
        monkeypatch.setenv("MIPT_OVERLAP_RAYS", rays)
        monkeypatch.setenv("MIPT_FUSE_RAYS", fuse)
        ctx = Context(0)
        try:
            for name in ("cornellBoxes", "areaLight", "mesh"):
                ctx.upload_scene(synth.make_scene(20000, seed=3) if name == "mesh" else scenes(name))
                for batch in (0, 3000):                                   # one batch; many small ones with spills and refills
                    p = abi.default_params(96, 96, spp=6, seed=21, batch_samples=batch)
                    frames.setdefault((name, batch), []).append(ctx.render(p, want_rgb8=False)[0].astype(np.float64))
        finally:
            ctx.close()
    for key, (one, *others) in frames.items():
        assert one.max() > 0.05, key
        for other in others:
            assert np.abs(other - one).max() <= 1e-6, key  # float32 frames: an ulp at most


def _mesh_scene_with_analytic_occluders(seed=11):
    """The small benchmark layout plus a sphere, a box and a square hovering over the terrain (they cast shadows
    and are hit themselves), beside the floor square: four analytic primitives among 20 000 triangles."""
    from path_tracer_amd.scene import ObjectRec

    sc = synth.make_scene(20000, seed=seed)
    def place(kind, mat, pos, scale, tilt=0.0):
        T = np.eye(4)
        T[:3, 3] = pos
        Sm = np.diag([scale[0], scale[1], scale[2], 1.0])
        c, s_ = math.cos(tilt), math.sin(tilt)
        Rx = np.array([[1, 0, 0, 0], [0, c, -s_, 0], [0, s_, c, 0], [0, 0, 0, 1.0]])
        sc.objects.append(ObjectRec(kind, mat, -1, T @ Rx @ Sm))
    place(abi.OBJ_SPHERE, 5, [-3.0, 2.5, -22.0], [1.6, 1.2, 1.6])
    place(abi.OBJ_BOX, 9, [3.5, 1.5, -26.0], [2.5, 1.0, 2.0], tilt=0.4)
    place(abi.OBJ_SQUARE, 2, [0.5, 4.0, -30.0], [6.0, 6.0, 1.0], tilt=-1.3)
    return sc


def test_analytic_primitives_beside_meshes_block_shadow_rays(oracle_lib, monkeypatch):
    """In a mesh scene with a few analytic primitives the shadow walk and the closest-hit walk pass those over: the
    shading kernel lets them improve on a closest hit, and what they block is dropped either by a pass per hit behind
    the shadow walk (shaft of the hit's rays against the primitive's box, then the rays themselves) or, with one-byte
    records (MIPT_VIS_BYTES=1 here; the default in big scenes), by the sum kernel asking the same questions itself;
    MIPT_LEAN_ANY=0 keeps their tests in both walks.  All three must give the oracle's frame, shadows of the
    analytic primitives on the meshes, on each other and on themselves included."""
    from path_tracer_amd.raytracer import Context

    sc = _mesh_scene_with_analytic_occluders()
    orc = oracle_lib.Oracle(sc)
    frames = []
    for knob, vis_bytes in (("1", "0"), ("0", "0"), ("1", "1")):
        monkeypatch.setenv("MIPT_LEAN_ANY", knob)
        monkeypatch.setenv("MIPT_VIS_BYTES", vis_bytes)
        ctx = Context(0)
        try:
            ctx.upload_scene(sc)
            p = abi.default_params(160, 90, spp=1, rr_q=0.0, max_depth=3, flags=abi.FLAG_FIXED_AREA_PATTERN)
            f32, rgb8 = ctx.render(p)
            ref, ref8 = orc.render(p, threads=8, want_rgb8=True)
            assert_images_match(f32, ref, rgb8, ref8)
            p = abi.default_params(128, 72, spp=3, seed=9)
            f32, rgb8 = ctx.render(p)
            ref, ref8 = orc.render(p, threads=8, want_rgb8=True)
            assert_images_match(f32, ref, rgb8, ref8)
            frames.append(f32.astype(np.float64))
        finally:
            ctx.close()
    assert np.abs(frames[0] - frames[1]).max() <= 1e-6 and np.abs(frames[0] - frames[2]).max() <= 1e-6


def test_one_byte_visibility_records_give_the_fp64_records_frame(scenes, monkeypatch):
    # Where every light's per-sample attenuation is a constant of the light (area lights with b = c = 0, light.cpp:139-142)
    # a shadow ray's record can be one byte, the sums putting the constant in (the default in scenes of more than 4096
    # primitives, MIPT_VIS_BYTES=1 everywhere); MIPT_VIS_BYTES=0 keeps the fp64 records.
    # The additions are the same ones in the same order: the frames agree up to the order of the atomic adds.
    # (An area light with b or c set uses the fp64 records either way: the second mesh scene.)
    import copy

    from path_tracer_amd.raytracer import Context
    curved = synth.make_scene(20000, seed=3)
    for L in curved.lights:
        if L.type == abi.LIGHT_RECT_AREA:
            L.atten = np.array([0.6, 0.02, 0.001], dtype=np.float32)
    frames = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("MIPT_VIS_BYTES", knob)
        ctx = Context(0)
        try:
            for name, sc in (("areaLight", scenes("areaLight")), ("cornell_area", synth.cornell_area_light(copy.deepcopy(scenes("cornellBoxes")))),
                             ("mesh", synth.make_scene(20000, seed=3)), ("mesh_curved_attenuation", curved)):
                ctx.upload_scene(sc)
                for batch in (0, 3000):
                    p = abi.default_params(96, 96, spp=6, seed=5, batch_samples=batch)
                    frames.setdefault((name, batch), []).append(ctx.render(p, want_rgb8=False)[0].astype(np.float64))
        finally:
            ctx.close()
    for key, (bytes_, doubles) in frames.items():
        assert bytes_.max() > 0.05, key
        assert np.abs(bytes_ - doubles).max() <= 1e-6, key  # float32 frames: an ulp at most


# ------------------------------------------------------------------ edge cases
def test_edge_cases(gpu, oracle_lib):
    empty = U.scene([], lights=[U.dir_light((0, -1, 0))])
    upload(gpu, empty)
    f32, rgb8 = gpu.render(abi.default_params(5, 3, spp=2))
    assert f32.shape == (3, 5, 3) and f32.sum() == 0 and rgb8.sum() == 0
    t, prim, _ = gpu.intersect([[0, 0, 0]], [[0, 0, 1]])
    assert prim[0] == -1 and t[0] == 1000.0
    nolight = U.scene([(abi.OBJ_SPHERE, np.eye(4), U.material(ka=(0.2, 0.4, 0.6)))], ambient=(1, 1, 1), cam=U.camera(eye=(0, 0, 4)))
    upload(gpu, nolight)
    p = abi.default_params(1, 1, spp=3, rr_q=0.0, max_depth=1)
    f32, _ = gpu.render(p)
    ref = oracle_lib.Oracle(nolight).render(p)
    assert np.allclose(f32, ref, atol=1e-7)
    onetri = U.scene([(abi.OBJ_TRIMESH, np.eye(4), U.material(), 0)],
                     meshes=[type(U.quad_mesh())(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 0.0]]), np.array([[0, 1, 2], [0, 0, 3]], np.int32))],
                     cam=U.camera(eye=(0.2, 0.2, 3)))
    upload(gpu, onetri)                                              # the second face is degenerate and dropped (trimesh.h:139-141)
    assert gpu.scene_stats()["prims"] == 1
    t, prim, n = gpu.intersect([[0.2, 0.2, 3]], [[0, 0, -1]])
    assert prim[0] == 0 and t[0] == pytest.approx(3.0) and np.allclose(n[0], [0, 0, 1])


def test_errors_are_reported_not_swallowed(scenes):
    from path_tracer_amd.raytracer import Context, MiptError

    ctx = Context(0)
    with pytest.raises(MiptError) as e:
        ctx.render(abi.default_params(8, 8))
    assert e.value.code == abi.ERR_NOSCENE
    ctx.upload_scene(scenes("room"))
    for bad in (dict(spp=0), dict(rr_q=1.0), dict(max_depth=0), dict(max_depth=300), dict(shard_index=2, shard_count=2), dict(area_samples=0)):
        with pytest.raises(MiptError) as e:
            ctx.render(abi.default_params(8, 8, **bad))
        assert e.value.code == abi.ERR_INVALID
    sc = scenes("room")
    import copy
    broken = copy.deepcopy(sc)
    broken.objects[0].material = 99
    with pytest.raises(MiptError):
        ctx.upload_scene(broken)
    # coordinates no tree can be built from are refused at upload, and the context stays usable
    from path_tracer_amd import synth
    mesh_scene = synth.make_scene(2000, seed=5)
    for bad in (float("nan"), float("inf"), 1e300):
        broken = copy.deepcopy(mesh_scene)
        broken.meshes[0].vertices[3, 1] = bad
        with pytest.raises(MiptError) as e:
            ctx.upload_scene(broken)
        assert e.value.code == abi.ERR_INVALID
    broken = copy.deepcopy(mesh_scene)
    broken.objects[0].xform[0, 3] = float("nan")
    with pytest.raises(MiptError):
        ctx.upload_scene(broken)
    ctx.upload_scene(mesh_scene)
    f32, _ = ctx.render(abi.default_params(16, 16, spp=2), want_rgb8=False)
    assert np.isfinite(f32).all() and f32.max() > 0
    with pytest.raises(MiptError):
        Context(4096)
    ctx.close()


# --------------------------------------- properties at the benchmark's full size
def test_full_size_frame_properties(gpu):
    """1920x1080 on a 200k-triangle version of the benchmark scene: tile shards are disjoint and
    add up to the unsharded frame, sample ranges add up, every mean lies in [0, 1], and the two
    independent sums agree to fp64 rounding."""
    import torch
    from path_tracer_amd import synth

    sc = synth.make_scene(200_000)
    upload(gpu, sc)
    w, h = 1920, 1080
    full = torch.zeros(3 * w * h, dtype=torch.float64, device="cuda")
    gpu.render_accumulate(abi.default_params(w, h, spp=2), full.data_ptr())
    parts = []
    for r in range(2):
        t = torch.zeros_like(full)
        gpu.render_accumulate(abi.default_params(w, h, spp=2, shard_index=r, shard_count=2, band_rows=8), t.data_ptr())
        parts.append(t)
    torch.cuda.synchronize()
    rows = torch.arange(h, device="cuda")
    own0 = ((rows // 8) % 2 == 0)
    p0 = parts[0].view(h, -1)
    p1 = parts[1].view(h, -1)
    assert float(p0[~own0].abs().sum()) == 0 and float(p1[own0].abs().sum()) == 0
    assert float((parts[0] + parts[1] - full).abs().max()) <= 1e-12
    s = torch.zeros_like(full)
    gpu.render_accumulate(abi.default_params(w, h, spp=2, sample_begin=0, sample_end=1), s.data_ptr())
    gpu.render_accumulate(abi.default_params(w, h, spp=2, sample_begin=1, sample_end=2), s.data_ptr())
    torch.cuda.synchronize()
    assert float((s - full).abs().max()) <= 1e-12
    assert float(full.min()) >= 0 and float(full.max()) <= 2.0 + 1e-12
    assert float(full.mean()) > 0.05
