#!/usr/bin/env python3
"""One warm-up and one measured mipt_render_accumulate call of SPP samples per pixel on the benchmark scene (for kernel traces:
rocprofv3 --kernel-trace -- python3 tools/one_call.py 16).  usage: python tools/one_call.py SPP"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from path_tracer_amd import abi, synth
from path_tracer_amd.raytracer import Context
S = int(sys.argv[1])
sc = synth.make_scene(1000000, seed=0xC4)
ctx = Context(0); ctx.upload_scene(sc)
w, h = 1920, 1080
d = torch.zeros(3 * w * h, dtype=torch.float64, device="cuda:0")
p = abi.default_params(w, h, spp=S, batch_samples=1 << 28)
for _ in range(2):
    ctx.render_accumulate(p, d.data_ptr(), 0)
torch.cuda.synchronize()
