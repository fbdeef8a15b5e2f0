"""The two synthetic benchmark configurations as tests.

  * config 5 (10 M triangles, 64 rectangle area lights, a cubemap, 3840 x 2160): the light set-up,
    the environment and the frame size at a triangle count the CPU oracle can follow (50 k):
    per-light shadow attenuation and a small frame against the oracle, then the full 3840 x 2160
    frame at 1 spp through size-independent properties (tile shards are disjoint and add up,
    sample ranges add up, every mean lies in [0, 1]);
  * config 4 (1 M triangles, 1920 x 1080) at its full triangle count: every fourth row band of the
    frame at 1 spp against the oracle, same random streams.

Tolerances as in tests/test_gpu_parity.py: 1e-5 per channel on >= 99.8 % of the pixels.
"""
import os

import numpy as np
import pytest

from path_tracer_amd import abi
from tests import util as U
from tests.conftest import GOLDEN
from tests.test_gpu_parity import assert_images_match, upload

pytestmark = pytest.mark.gpu


def config5_scene(n_triangles):
    from path_tracer_amd import synth

    sc = synth.make_scene(n_triangles, seed=0xC5, n_area_lights=64, cubemap=True)
    z = np.load(os.path.join(GOLDEN, "cubemaps", "autumn.npz"))  # the cubemap the configuration names (assets/cubemaps/cubemap_autumn)
    sc.cubemap = [z[f"cube{k}"] for k in range(6)]
    return sc


def test_config5_lights_and_cubemap_against_the_oracle(gpu, oracle_lib):
    sc = config5_scene(50_000)
    assert len(sc.lights) == 65
    orc = oracle_lib.Oracle(sc)
    upload(gpu, sc)
    org, d = U.camera_rays(sc, 1500, seed=31)
    t, prim, n = orc.intersect(org, d)
    hit = prim >= 0
    assert 0.3 < hit.mean() < 1.0                                      # some rays reach the sky
    fire = org[hit] + t[hit, None] * d[hit] + n[hit] * 3e-8
    keys = np.arange(len(fire), dtype=np.uint64) * np.uint64(6151) + np.uint64(3)
    p = abi.default_params(8, 8)
    for li in (0, 7, 27, 36, 63, 64):                                  # corners, middle, the directional light
        a, b = gpu.shadow_attenuation(p, li, fire, keys), orc.shadow_attenuation(p, li, fire, keys)
        bad = np.abs(a - b).max(axis=1) > 1e-9
        assert bad.mean() <= 2e-3, f"light {li}: {bad.sum()} of {len(fire)} shading points differ"
    pr = abi.default_params(96, 54, spp=2, seed=11)                    # 641 shadow rays per shaded hit
    f32, rgb8 = gpu.render(pr)
    ref, ref8, cnt = orc.render(pr, threads=8, want_rgb8=True, want_counters=True)
    assert cnt["shadow_rays"] == 641 * cnt["shaded_hits"]
    assert ref.mean() > 0.05
    assert_images_match(f32, ref, rgb8, ref8)
    # sky pixels are exactly the cubemap lookup of the camera ray
    sky = (orc.intersect(*camera_grid(sc, 96, 54))[1] < 0).reshape(54, 96)
    assert sky.sum() > 50
    assert np.abs(f32[sky].astype(np.float64) - ref[sky]).max() <= 1e-7


def camera_grid(sc, w, h):
    """The camera rays tracePixel fires (x = i / w, y = j / h, RayTracer.cpp:70-71), row j = 0 first."""
    eye, look, u, v = (np.asarray(getattr(sc.camera, k), float) for k in ("eye", "look", "u", "v"))
    i, j = np.meshgrid(np.arange(w), np.arange(h))
    x, y = i.ravel() / w - 0.5, j.ravel() / h - 0.5
    d = look[None] + x[:, None] * u[None] + y[:, None] * v[None]
    d /= np.linalg.norm(d, axis=1)[:, None]
    return np.repeat(eye[None], len(d), 0), d


def test_config5_full_frame_properties(gpu):
    import torch

    upload(gpu, config5_scene(50_000))
    w, h = 3840, 2160
    full = torch.zeros(3 * w * h, dtype=torch.float64, device="cuda")
    gpu.render_accumulate(abi.default_params(w, h, spp=2), full.data_ptr())
    tiles = torch.zeros_like(full)
    for r in range(8):                                                 # the eight tile shards of the configuration
        one = torch.zeros_like(full)
        gpu.render_accumulate(abi.default_params(w, h, spp=2, shard_index=r, shard_count=8, band_rows=8), one.data_ptr())
        rows = torch.arange(h, device="cuda")
        own = ((rows // 8) % 8 == r)
        assert float(one.view(h, -1)[~own].abs().sum()) == 0            # nothing outside the shard's bands
        tiles += one
    samples = torch.zeros_like(full)
    for (a, b) in ((0, 1), (1, 2)):
        gpu.render_accumulate(abi.default_params(w, h, spp=2, sample_begin=a, sample_end=b), samples.data_ptr())
    torch.cuda.synchronize()
    assert float((tiles - full).abs().max()) <= 1e-12
    assert float((samples - full).abs().max()) <= 1e-12
    assert float(full.min()) >= 0 and float(full.max()) <= 2.0 + 1e-12
    assert 0.1 < float(full.mean()) / 2 < 0.9


def test_config4_full_scene_rows_against_the_oracle(gpu, oracle_lib):
    from path_tracer_amd import synth

    sc = synth.make_scene(1_000_000, seed=0xC4)
    assert sc.n_triangles() > 990_000
    orc = oracle_lib.Oracle(sc)
    upload(gpu, sc)
    w, h = 1920, 1080
    p = abi.default_params(w, h, spp=1, band_rows=2, shard_count=8, shard_index=3)   # 136 rows spread over the frame
    f32, _ = gpu.render(p, want_rgb8=False)
    ref = orc.render(p, threads=16)
    rows = np.arange(h)[(np.arange(h) // 2) % 8 == 3]
    assert np.abs(ref[rows]).mean() > 0.02
    diff = np.abs(f32[rows].astype(np.float64) - ref[rows]).max(axis=2)
    assert (diff <= 1e-5).mean() >= 0.998, f"{(diff > 1e-5).sum()} of {diff.size} pixels differ (max {diff.max():.3e})"
    other = np.setdiff1d(np.arange(h), rows)
    assert f32[other].sum() == 0                                        # rows outside the shard stay 0 (traceSetup zero-fills)


def test_config5_at_ten_million_triangles_against_the_oracle(gpu, oracle_lib):
    """Config 5 at its own triangle count: 10 M triangles (a 2 GB tree, eight times the Infinity Cache, the deepest
    traversal stacks), 64 area lights + the directional one, the cubemap, 3840 x 2160.  20 000 closest hits ray by
    ray, then three rows of the frame at 1 spp (641 shadow rays per shaded hit) against the oracle."""
    sc = config5_scene(10_000_000)
    assert sc.n_triangles() > 9_900_000 and len(sc.lights) == 65
    orc = oracle_lib.Oracle(sc)
    upload(gpu, sc)
    assert gpu.scene_stats()["device_bytes"] > 1_500_000_000
    org, d = U.camera_rays(sc, 20000, seed=77)
    t, prim, n = orc.intersect(org, d)
    tg, pg, ng = gpu.intersect(org, d)
    hit = prim >= 0
    assert 0.3 < hit.mean() < 1.0
    same = pg == prim
    assert same.mean() >= 0.9995, f"{(~same).sum()} of {len(same)} rays end on another primitive"
    ok = same & hit
    assert np.abs(tg[ok] - t[ok]).max() <= 1e-9 and np.abs(ng[ok] - n[ok]).max() <= 1e-9
    w, h = 3840, 2160
    p = abi.default_params(w, h, spp=1, band_rows=1, shard_count=720, shard_index=300)   # rows 300, 1020, 1740
    f32, _ = gpu.render(p, want_rgb8=False)
    ref, cnt = orc.render(p, threads=16, want_counters=True)
    assert cnt["shadow_rays"] == 641 * cnt["shaded_hits"] and cnt["shaded_hits"] > 5000
    rows = np.array([300, 1020, 1740])
    assert np.abs(ref[rows]).mean() > 0.02
    diff = np.abs(f32[rows].astype(np.float64) - ref[rows]).max(axis=2)
    assert (diff <= 1e-5).mean() >= 0.998, f"{(diff > 1e-5).sum()} of {diff.size} pixels differ (max {diff.max():.3e})"
    other = np.setdiff1d(np.arange(h), rows)
    assert f32[other].sum() == 0
