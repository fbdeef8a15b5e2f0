#!/usr/bin/env python3
"""Reduces the renders the reference publishes (GUI screenshots of its own path tracer,
ray/*.png) to small statistical fixtures: the 32x32 grid of block-mean RGB of the rendered
area.  The screenshots are stochastic renders at the reference's fixed settings (100 spp,
q = 0.1, 10 area samples) seen through a window, so they pin the look of the result — where
the light falls, how bright and what colour every region is — not individual pixels.

usage (in the build container, where /root/reference exists):
    python tools/make_screenshot_fixtures.py
writes tests/golden/screenshots/<scene>.npy (float32, 32x32x3, row 0 = top of the image).
`top` is the height of the window's title bar in the screenshot (0 where it was cropped off).
"""
import os

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/ray"
SHOTS = {  # scene fixture name -> (screenshot, title-bar rows)
    "spheres": ("spheresImage.png", 54),
    "spheresDirectional": ("spheresDirectionalImage.png", 60),
    "cornellBoxes": ("cornellBoxImage.png", 6),
    "hitchcockBRDF": ("hitchcockbrdf.png", 8),
}
GRID = 32


def block_means(img, n=GRID):
    """img: (h, w, 3) floats, row 0 = top.  Returns (n, n, 3) block means."""
    h, w, _ = img.shape
    ys = (np.arange(n + 1) * h / n).astype(int)
    xs = (np.arange(n + 1) * w / n).astype(int)
    return np.array([[img[ys[i]:ys[i + 1], xs[j]:xs[j + 1]].mean(axis=(0, 1)) for j in range(n)] for i in range(n)])


def main():
    out = os.path.join(ROOT, "tests", "golden", "screenshots")
    os.makedirs(out, exist_ok=True)
    for name, (png, top) in SHOTS.items():
        a = np.asarray(Image.open(os.path.join(REF, png)).convert("RGB")).astype(np.float64) / 255.0
        np.save(os.path.join(out, name + ".npy"), block_means(a[top:]).astype(np.float32))
        print(name, png, a.shape, "top", top)


if __name__ == "__main__":
    main()
