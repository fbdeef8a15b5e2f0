import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from path_tracer_amd import abi, synth
from path_tracer_amd.raytracer import Context
tris = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sc = synth.make_scene(tris)
ctx = Context(0); t0=time.time(); ctx.upload_scene(sc); print('upload', time.time()-t0, ctx.scene_stats())
w,h = 480,270
d_sum = torch.zeros(3*w*h, dtype=torch.float64, device='cuda')
for md in (1,2,4,8,16,32,64):
    ctx.reset_counters()
    p = abi.default_params(w,h,spp=1,max_depth=md, flags=abi.FLAG_COUNTERS)
    t0=time.time(); ctx.render_accumulate(p, d_sum.data_ptr()); dt=time.time()-t0
    c = ctx.counters(); n = w*h
    print(f"max_depth {md:3d}: path rays/sample {c['path_rays']/n:8.2f} shaded/sample {c['shaded_hits']/n:8.2f} shadow/sample {c['shadow_rays']/n:8.1f} nodes/ray {c['node_visits']/(c['path_rays']+c['shadow_rays']):6.1f} prims/ray {c['prim_tests']/(c['path_rays']+c['shadow_rays']):5.1f} time {dt:.3f}")
