#!/usr/bin/env python3
"""Duration of one level's launches against the number of rays in the level: frames of n pixels of the
benchmark scene at 1 spp, depth 1 (one intersect, one shade, one shadow launch, one sum)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from path_tracer_amd import abi, synth
from path_tracer_amd.raytracer import Context

sc = synth.make_scene(int(os.environ.get("TRIS", "1000000")), seed=0xC4)
ctx = Context(0)
ctx.upload_scene(sc)
for (w, h) in ((16, 4), (32, 8), (64, 16), (128, 32), (256, 64), (512, 128), (1024, 256), (1920, 1080)):
    d = torch.zeros(3 * w * h, dtype=torch.float64, device="cuda:0")
    p = abi.default_params(w, h, spp=2, max_depth=2, flags=abi.FLAG_PROFILE)   # depth 1 rays are distinct (depth 0 traces one ray per pixel)
    for _ in range(3):
        ctx.render_accumulate(p, d.data_ptr(), 0)
    ctx.reset_counters()
    reps = 10
    for _ in range(reps):
        ctx.render_accumulate(p, d.data_ptr(), 0)
    torch.cuda.synchronize()
    kt = ctx.kernel_times()
    print(f"{w*h*2:8d} samples: per launch (us): " + "  ".join(f"{k} {v['ms']*1e3/max(1,v['launches']):.0f} ({v['items']//max(1,v['launches'])} items)" for k, v in kt.items() if v['launches']), flush=True)
