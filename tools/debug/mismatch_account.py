"""Which pixels of the mesh-scene frames differ between the HIP path and the oracle, and why.
For every pixel over the tolerance the camera ray is traced on both sides; if the primary hits agree,
the first bounce and mirror rays are, and so on down the tree of the deterministic frame (RR off)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from path_tracer_amd import abi, synth
from path_tracer_amd.raytracer import Context
from oracle import binding
from tests.test_gpu_configs import camera_grid

gpu = Context(0)
for label, sc, (w, h), kw in (
    ("synth 20k, deterministic", synth.make_scene(20000, seed=7), (128, 72), dict(spp=1, rr_q=0.0, max_depth=4, flags=abi.FLAG_FIXED_AREA_PATTERN)),
    ("synth 20k, 2 spp", synth.make_scene(20000, seed=7), (96, 54), dict(spp=2)),
    ("synth 200k, deterministic", synth.make_scene(200000, seed=7), (256, 144), dict(spp=1, rr_q=0.0, max_depth=4, flags=abi.FLAG_FIXED_AREA_PATTERN)),
    ("synth 1M, deterministic", synth.make_scene(1000000), (256, 144), dict(spp=1, rr_q=0.0, max_depth=3, flags=abi.FLAG_FIXED_AREA_PATTERN)),
):
    orc = binding.Oracle(sc)
    gpu.upload_scene(sc)
    p = abi.default_params(w, h, **kw)
    f32, _ = gpu.render(p, want_rgb8=False)
    ref = orc.render(p, threads=16)
    diff = np.abs(f32.astype(np.float64) - ref).max(axis=2)
    bad = np.argwhere(diff > 1e-5)
    org, d = camera_grid(sc, w, h)
    tg, pg, ng = gpu.intersect(org, d)
    to, po, no = orc.intersect(org, d)
    prim_differs = (pg != po).reshape(h, w)
    t_close = (np.abs(tg - to) <= 1e-9 * np.maximum(1, to)).reshape(h, w)
    n_primary = sum(1 for (j, i) in bad if prim_differs[j, i])
    print(f"{label}: {len(bad)} of {w*h} pixels over 1e-5 ({len(bad)/(w*h):.5f}); {n_primary} of them have a different primary primitive "
          f"(same t: {sum(1 for (j,i) in bad if prim_differs[j,i] and t_close[j,i])}); all primary differences in the frame: {int(prim_differs.sum())}", flush=True)
    # the remaining ones: look at the hit points' bounce rays?  report their diff sizes
    rest = [(j, i) for (j, i) in bad if not prim_differs[j, i]]
    if rest:
        print("   deeper:", [(int(j), int(i), float(diff[j, i])) for (j, i) in rest[:8]])
