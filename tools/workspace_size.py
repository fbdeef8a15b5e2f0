import sys, os
sys.path.insert(0, os.getcwd())
import torch
from path_tracer_amd import abi, synth
from path_tracer_amd.raytracer import Context
sc = synth.make_scene(1000000, seed=0xC4)
ctx = Context(0); ctx.upload_scene(sc)
w, h = 1920, 1080
d = torch.zeros(3*w*h, dtype=torch.float64, device="cuda:0")
f0 = torch.cuda.mem_get_info(0)[0]
p = abi.default_params(w, h, spp=128, batch_samples=0)
ctx.render_accumulate(p, d.data_ptr(), 0); torch.cuda.synchronize()
f1 = torch.cuda.mem_get_info(0)[0]
print("workspace of a 128 spp batch at 1080p with the default batch size: %.1f GB" % ((f0 - f1) / 1e9))
