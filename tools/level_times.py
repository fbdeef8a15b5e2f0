#!/usr/bin/env python3
"""Kernel time of the benchmark scene by path depth: renders with max_depth = 1, 2, 3, ... and
prints the increments (HIP-event kernel times of the library, MIPT_FLAG_PROFILE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from path_tracer_amd import abi, synth
from path_tracer_amd.raytracer import Context

sc = synth.make_scene(1_000_000, seed=0xC4)
ctx = Context(0)
ctx.upload_scene(sc)
w, h, S = 1920, 1080, 16
d = torch.zeros(3 * w * h, dtype=torch.float64, device="cuda:0")
prev = None
for md in (1, 2, 3, 4, 6, 64):
    p = abi.default_params(w, h, spp=S, batch_samples=1 << 25, max_depth=md, flags=abi.FLAG_PROFILE | abi.FLAG_COUNTERS)
    ctx.render_accumulate(p, d.data_ptr(), 0)  # warm-up
    ctx.reset_counters()
    ctx.render_accumulate(p, d.data_ptr(), 0)
    torch.cuda.synchronize()
    kt, c = ctx.kernel_times(), ctx.counters()
    row = dict(md=md, isect_ms=kt["intersect"]["ms"], shadow_ms=kt["shadow"]["ms"], path_rays=c["path_rays"], shadow_rays=c["shadow_rays"],
               snodes=c["shadow_node_visits"], sprims=c["shadow_prim_tests"])
    if prev:
        dr = row["shadow_rays"] - prev["shadow_rays"]
        print(f"depth {prev['md']}..{md - 1}: shadow {row['shadow_ms'] - prev['shadow_ms']:.1f} ms for {dr / 1e6:.0f} M rays = "
              f"{(row['shadow_ms'] - prev['shadow_ms']) * 1e6 / max(dr, 1):.3f} ns/ray, nodes/ray {(row['snodes'] - prev['snodes']) / max(dr, 1):.1f}, "
              f"prims/ray {(row['sprims'] - prev['sprims']) / max(dr, 1):.1f}; intersect {row['isect_ms'] - prev['isect_ms']:.1f} ms for "
              f"{(row['path_rays'] - prev['path_rays']) / 1e6:.0f} M rays")
    else:
        print(f"depth 0: shadow {row['shadow_ms']:.1f} ms for {row['shadow_rays'] / 1e6:.0f} M rays = {row['shadow_ms'] * 1e6 / row['shadow_rays']:.3f} ns/ray, "
              f"nodes/ray {row['snodes'] / row['shadow_rays']:.1f}, prims/ray {row['sprims'] / row['shadow_rays']:.1f}; intersect {row['isect_ms']:.1f} ms for {row['path_rays'] / 1e6:.0f} M rays")
    prev = row
