#!/usr/bin/env python3
"""Where a mipt_render_accumulate call's wall time goes on the benchmark scene, at a small and a large batch: the sum
of the kernels' own durations (HIP events around every launch) against the wall time of the call, and the launches
per kind.  What is left over is the host's turnaround between tree levels (a read-back of two counters each) and
the time the chip idles in launches of fewer rays than it has lanes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from path_tracer_amd import abi, synth
from path_tracer_amd.raytracer import Context

sc = synth.make_scene(int(os.environ.get("TRIS", "1000000")), seed=0xC4)
ctx = Context(0)
ctx.upload_scene(sc)
w, h = 1920, 1080
d = torch.zeros(3 * w * h, dtype=torch.float64, device="cuda:0")
for S in (4, 16, 128):
    for flags, tag in ((0, "plain"), (abi.FLAG_PROFILE, "events")):
        p = abi.default_params(w, h, spp=S, batch_samples=1 << 28, flags=flags)
        ctx.render_accumulate(p, d.data_ptr(), 0)
        ctx.reset_counters()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.render_accumulate(p, d.data_ptr(), 0)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 3 * 1e3
        if flags:
            kt = ctx.kernel_times()
            ksum = sum(v["ms"] for v in kt.values()) / 3
            print(f"{S:4d} spp {tag}: wall {wall:7.1f} ms, kernels {ksum:7.1f} ms, gap {wall - ksum:5.1f} ms; launches per call: " +
                  ", ".join(f"{k} {v['launches'] // 3} ({v['ms'] / 3:.1f} ms)" for k, v in kt.items() if v["launches"]))
        else:
            print(f"{S:4d} spp {tag}: wall {wall:7.1f} ms = {w * h * S / wall / 1e3:.1f} Msamples/s")
