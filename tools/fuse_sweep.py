import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from path_tracer_amd import abi, synth
from path_tracer_amd.raytracer import Context
sc = synth.make_scene(1000000, seed=0xC4)
ctx = Context(0)
w, h = 1920, 1080
d = torch.zeros(3 * w * h, dtype=torch.float64, device="cuda:0")
for fuse in [int(x) for x in os.environ.get('FUSE', '0,16384,65536,131072,262144,524288,1048576').split(',')]:
    ctx.set_option("fuse_rays", fuse)
    if os.environ.get("OVERLAP_FOLLOWS"):
        ctx.set_option("overlap_rays", max(fuse, 1 << 18))
    ctx.upload_scene(sc)
    for S in (16, 128):
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
