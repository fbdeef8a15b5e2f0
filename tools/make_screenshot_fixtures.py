#!/usr/bin/env python3
"""Reduces the renders the reference publishes (GUI screenshots of its own path tracer,
ray/*.png) to small statistical fixtures: the 32x32 grid of block-mean RGB of the rendered
area.  The screenshots are stochastic renders at the reference's fixed settings (100 spp,
q = 0.1, 10 area samples) seen through a window, so they pin the look of the result — where
the light falls, how bright and what colour every region is — not individual pixels.

usage (in the build container, where /root/reference exists):
    python tools/make_screenshot_fixtures.py
writes tests/golden/screenshots/<scene>.npy (float32, 32x32x3, row 0 = top of the image) and
tests/golden/cubemaps/autumn.npz (the six faces of assets/cubemaps/cubemap_autumn, the
environment of the lowroughnessHighMetallic screenshot, box-filtered to 128 x 128, 8-bit RGB,
rows bottom-up, in CubeMap::tMap order +x -x +y -y +z -z).
The crop box of each screenshot (title bar, window border) is read off the row and column
means: the title bar is a uniform 0.23 grey that ends in the first row of the render, borders
are one to six pixels of a different brightness than the render inside them.  A scan of
boxes a few pixels either way against the restatement rendered at the reference's 512 x 512
picks these same boxes (best or within 0.0004 of the best mean absolute block difference), so
the boxes are not a fitted parameter.  Renders at a lower resolution sit up to 1.5 reference
pixels off (pixel i samples x = i / w, its left edge, RayTracer.cpp:70-71), which is why the
CPU test renders at 256 x 256 and the bounds leave room for that.

Not used: ray/softShadows.png.  README.md:33-35 names its scene "spheres.json (OLD SCENE)":
five spheres in a row under cubemap_terrain, a scene and camera that are no longer on disk
(neither spheres.json nor areaLight.json reproduces its sky region: block correlation -0.2 / 0.26).
"""
import os

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/ray"
SHOTS = {  # scene fixture name -> (screenshot, (top, bottom, left, right) pixels outside the rendered area)
    "spheres": ("spheresImage.png", (54, 0, 0, 0)),                   # 1022 x 1014
    "spheresDirectional": ("spheresDirectionalImage.png", (54, 4, 4, 0)),  # 1022 x 1020
    "cornellBoxes": ("cornellBoxImage.png", (2, 3, 1, 0)),            # 1023 x 1021
    "hitchcockBRDF": ("hitchcockbrdf.png", (2, 1, 4, 6)),             # 1205 x 1204
    "singleSphere": ("lowroughnessHighMetallic.png", (66, 2, 0, 2)),  # 1092 x 1084; singleSphere.json under cubemap_autumn (README.md:37-40)
}
CUBEMAP_DIR = "assets/cubemaps/cubemap_autumn"
CUBE_SIZE = 128
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
    for name, (png, (top, bottom, left, right)) in SHOTS.items():
        a = np.asarray(Image.open(os.path.join(REF, png)).convert("RGB")).astype(np.float64) / 255.0
        h, w, _ = a.shape
        crop = a[top:h - bottom, left:w - right]
        np.save(os.path.join(out, name + ".npy"), block_means(crop).astype(np.float32))
        print(name, png, a.shape, "->", crop.shape)
    # the cubemap faces, matched to their slots as TraceUI::matchCubemapFiles does (ui/TraceUI.cc:72-134)
    faces = {}
    cdir = os.path.join(REF, CUBEMAP_DIR)
    for k, (sign, axis) in enumerate((("pos", "x"), ("neg", "x"), ("pos", "y"), ("neg", "y"), ("pos", "z"), ("neg", "z"))):
        hits = [f for f in sorted(os.listdir(cdir)) if f.endswith(".png") and sign in f and axis in f[f.find(sign):]]
        assert len(hits) == 1, (sign, axis, hits)
        im = Image.open(os.path.join(cdir, hits[0]))
        assert im.info.get("gamma") is None  # no gAMA chunk: the reference's reader leaves the samples alone (fileio/pngimage.cpp:158-159)
        im = im.convert("RGB").resize((CUBE_SIZE, CUBE_SIZE), Image.BOX)
        faces[f"cube{k}"] = np.ascontiguousarray(np.asarray(im, dtype=np.uint8)[::-1])  # rows bottom-up (pngimage.cpp:217-221)
        print("cubemap face", k, hits[0])
    os.makedirs(os.path.join(ROOT, "tests", "golden", "cubemaps"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cubemaps", "autumn.npz"), **faces)


if __name__ == "__main__":
    main()
