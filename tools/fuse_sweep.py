import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from path_tracer_amd import abi, synth
from path_tracer_amd.raytracer import Context
sc = synth.make_scene(1000000, seed=0xC4)
ctx = Context(0)
w, h = 1920, 1080
d = torch.zeros(3 * w * h, dtype=torch.float64, device="cuda:0")
for fuse in (0, 1<<12, 1<<14, 1<<15, 1<<16, 1<<17, 1<<18):
    ctx.set_option("fuse_rays", fuse)
    ctx.upload_scene(sc)
    for S in (16,):
        p = abi.default_params(w, h, spp=S, batch_samples=1 << 28, flags=abi.FLAG_PROFILE)
        ctx.render_accumulate(p, d.data_ptr(), 0)
        ctx.reset_counters()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.render_accumulate(p, d.data_ptr(), 0)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 3 * 1e3
        kt = ctx.kernel_times()
        print(f"fuse_rays {fuse:7d}: {S} spp wall {wall:6.1f} ms; " + ", ".join(f"{k} {v['launches'] // 3} ({v['ms'] / 3:.1f})" for k, v in kt.items() if v["launches"]), flush=True)
