#!/usr/bin/env python3
"""Flatten the reference's path-tracer demo scenes into scene-description fixtures.

Runs only where /root/reference is mounted.  Each JSON scene under
ray/assets/pathTracer (and the feature probes under ray/assets/custom that the hot
path can render) is parsed by path_tracer_amd.scene.load_json_scene and stored as
tests/golden/scenes/<name>.npz: numeric arrays only (camera vectors, 4x4 object
transforms, material and light parameters) — the input of mipt_upload_scene.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from path_tracer_amd.scene import load_json_scene  # noqa: E402

REF = "/root/reference/ray/assets"
SCENES = {
    "spheres": "pathTracer/spheres.json",
    "spheresDirectional": "pathTracer/spheresDirectional.json",
    "cornellBoxes": "pathTracer/cornellBoxes.json",
    "hitchcockBRDF": "pathTracer/hitchcockBRDF.json",
    "singleSphere": "pathTracer/singleSphere.json",
    "areaLight": "pathTracer/areaLight.json",
    "room": "custom/room.json",
    "russianRouletteTest": "custom/russianRouletteTest.json",
}

if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden", "scenes")
    os.makedirs(out, exist_ok=True)
    for name, rel in SCENES.items():
        sc = load_json_scene(os.path.join(REF, rel))
        sc.save_npz(os.path.join(out, name + ".npz"))
        print(f"{name}: {len(sc.objects)} objects, {len(sc.lights)} lights, {sc.n_triangles()} triangles")
