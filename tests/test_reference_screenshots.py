"""The only outputs of its path tracer the reference publishes are GUI screenshots of four
demo scenes (ray/spheresImage.png, spheresDirectionalImage.png, cornellBoxImage.png,
hitchcockbrdf.png): stochastic renders at its fixed settings, seen through a window.
tools/make_screenshot_fixtures.py reduced each to a 32x32 grid of block-mean RGB
(tests/golden/screenshots/*.npy).  These tests pin the LOOK of our results to them — where
light and shadow fall, how bright and what colour every region is — which is as far as a
non-reproducible reference can pin anything:

  * luminance of the block grid correlates >= MIN_CORR with the screenshot's,
  * mean absolute block difference <= MAX_MAD (8-bit sRGB screenshot vs linear buffer:
    the reference writes its buffer to the window unconverted, so they are comparable),
  * overall mean brightness within 12 %.

Window scaling and the title-bar crop are only approximately known, hence blocks, not pixels.
"""
import os

import numpy as np
import pytest

from path_tracer_amd import abi
from tests.conftest import GOLDEN

GRID = 32
CASES = {  # scene: (oracle spp at 128x128, min corr, max mean-abs-diff)
    "spheres": (48, 0.975, 0.02),
    "spheresDirectional": (32, 0.99, 0.012),
    "cornellBoxes": (32, 0.99, 0.012),
    "hitchcockBRDF": (96, 0.93, 0.012),  # a dark image: the screenshot's own noise limits the correlation (0.94 at 100 spp)
}


def block_means(img, n=GRID):
    h, w, _ = img.shape
    ys = (np.arange(n + 1) * h / n).astype(int)
    xs = (np.arange(n + 1) * w / n).astype(int)
    return np.array([[img[ys[i]:ys[i + 1], xs[j]:xs[j + 1]].mean(axis=(0, 1)) for j in range(n)] for i in range(n)])


def compare(mean_image_bottom_up, name, min_corr, max_mad):
    shot = np.load(os.path.join(GOLDEN, "screenshots", name + ".npy")).astype(np.float64)
    ours = block_means(np.asarray(mean_image_bottom_up, dtype=np.float64)[::-1])  # buffer row 0 is the bottom row
    corr = np.corrcoef(shot.mean(axis=2).ravel(), ours.mean(axis=2).ravel())[0, 1]
    mad = np.abs(shot - ours).mean()
    assert corr >= min_corr, f"{name}: block luminance correlation {corr:.4f} < {min_corr}"
    assert mad <= max_mad, f"{name}: mean absolute block difference {mad:.4f} > {max_mad}"
    assert abs(ours.mean() - shot.mean()) <= 0.12 * shot.mean(), f"{name}: brightness {ours.mean():.4f} vs {shot.mean():.4f}"
    return corr, mad


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_the_published_renders(name, scenes, oracles):
    spp, min_corr, max_mad = CASES[name]
    p = abi.default_params(128, 128, spp=spp, seed=5)
    img = oracles(name).render(p, threads=8, want_rgb8=True)[0]
    compare(img, name, min_corr, max_mad)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_reproduces_the_published_renders(name, gpu, scenes):
    # the reference's own settings: 512 x 512, 100 spp (RayTracer.cpp:61)
    gpu.upload_scene(scenes(name))
    p = abi.default_params(512, 512, spp=100, seed=9)
    f32, _ = gpu.render(p, want_rgb8=False)
    _, min_corr, max_mad = CASES[name]
    compare(f32, name, min_corr, max_mad)
