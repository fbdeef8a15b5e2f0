#!/usr/bin/env python3
"""Times mipt_render (host buffers in and out, as the reference's traceImage) on a fixture scene:
usage: time_render.py <scene> <width> <height> <spp> [batch_samples ...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from path_tracer_amd import abi
from path_tracer_amd.raytracer import Context
from path_tracer_amd.scene import Scene

name, w, h, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sc = Scene.load_npz(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "scenes", name + ".npz"))
ctx = Context(0)
ctx.upload_scene(sc)
for b in [int(x) for x in sys.argv[5:]] or [0]:
    p = abi.default_params(w, h, spp=spp, batch_samples=b)
    ctx.render(p)  # warm-up: allocates the workspace
    t = time.perf_counter()
    ctx.render(p)
    dt = time.perf_counter() - t
    print(f"{name} {w}x{h} {spp} spp batch {b}: {dt*1e3:.1f} ms, {w*h*spp/dt/1e6:.1f} Msamples/s (host buffers, D2H included)")
