#!/usr/bin/env python3
"""Ad-hoc GPU-vs-oracle check used during bring-up (the real tests live in tests/)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from path_tracer_amd import abi, synth
from path_tracer_amd.scene import Scene
from path_tracer_amd.raytracer import Context
from oracle.binding import Oracle

def rays_for(sc, n, rng):
    eye = np.asarray(sc.camera.eye); look = np.asarray(sc.camera.look); u = np.asarray(sc.camera.u); v = np.asarray(sc.camera.v)
    xy = rng.uniform(-0.5, 0.5, size=(n, 2))
    d = look[None] + xy[:, :1] * u[None] + xy[:, 1:] * v[None]
    d /= np.linalg.norm(d, axis=1)[:, None]
    o = np.repeat(eye[None], n, 0)
    return o, d

def check_scene(name, sc, w=96, h=96):
    ctx = Context(0); t0 = time.time(); ctx.upload_scene(sc); print(name, 'upload', round(time.time()-t0, 3), ctx.scene_stats())
    orc = Oracle(sc)
    rng = np.random.RandomState(1)
    o, d = rays_for(sc, 20000, rng)
    tg, pg, ng = ctx.intersect(o, d)
    to, po, no = orc.intersect(o, d)
    hit = po >= 0
    print('  camera rays: hit frac', hit.mean(), 'prim mismatch', int((pg != po).sum()), 'max |dt|', np.abs(tg - to)[hit & (pg == po)].max() if hit.any() else 0,
          'max |dN|', np.abs(ng - no)[hit & (pg == po)].max() if hit.any() else 0)
    # secondary rays from hit points in random directions
    P = o[hit] + to[hit, None] * d[hit]
    d2 = rng.normal(size=P.shape); d2 /= np.linalg.norm(d2, axis=1)[:, None]
    o2 = P + d2 * 1e-8
    tg, pg, ng = ctx.intersect(o2, d2); to, po, no = orc.intersect(o2, d2)
    h2 = po >= 0
    print('  secondary rays: hit frac', h2.mean(), 'prim mismatch', int((pg != po).sum()), 'max |dt|', np.abs(tg - to)[h2 & (pg == po)].max() if h2.any() else 0)
    for flags, q, md, spp, label in ((abi.FLAG_FIXED_AREA_PATTERN, 0.0, 4, 1, 'deterministic 1spp'), (0, 0.1, 64, 4, 'stochastic 4spp')):
        p = abi.default_params(w, h, spp=spp, rr_q=q, max_depth=md, flags=flags)
        t0 = time.time(); f, b = ctx.render(p); tg_ = time.time() - t0
        t0 = time.time(); ref, ref8 = orc.render(p, threads=8, want_rgb8=True); to_ = time.time() - t0
        diff = np.abs(f.astype(np.float64) - ref)
        print(f'  {label}: gpu {tg_:.3f}s oracle {to_:.3f}s max|d| {diff.max():.3e} frac>1e-6 {(diff.max(axis=2) > 1e-6).mean():.5f} rgb8 mismatches {(b != ref8).sum()} mean {ref.mean():.4f}')
    ctx.close()

if __name__ == '__main__':
    gold = os.path.join(ROOT, 'tests', 'golden', 'scenes')
    for name in ('spheres', 'cornellBoxes', 'hitchcockBRDF', 'singleSphere', 'areaLight', 'spheresDirectional', 'room'):
        check_scene(name, Scene.load_npz(os.path.join(gold, name + '.npz')))
    check_scene('synth20k', synth.make_scene(20000, seed=7), 96, 54)
