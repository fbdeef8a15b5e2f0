#!/usr/bin/env python3
"""The per-rank load of N ranks on one GPU, both ways of sharding the benchmark frame (128 spp of 1920x1080, 1 M triangles):
samples (every rank the whole frame at 128/N spp) and tiles (every rank all 128 spp of its bands of rows), rank by rank."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from path_tracer_amd import abi, synth
from path_tracer_amd.raytracer import Context

N = int(os.environ.get("RANKS", "8"))
sc = synth.make_scene(1000000, seed=0xC4)
ctx = Context(0)
ctx.upload_scene(sc)
w, h, S = 1920, 1080, 128
d = torch.zeros(3 * w * h, dtype=torch.float64, device="cuda:0")

def timed(p):
    ctx.render_accumulate(p, d.data_ptr(), 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.render_accumulate(p, d.data_ptr(), 0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 3 * 1e3

whole = timed(abi.default_params(w, h, spp=S, batch_samples=1 << 28))
print(f"one rank, {S} spp: {whole:.1f} ms; an {N}th: {whole / N:.1f} ms")
p = abi.default_params(w, h, spp=S, batch_samples=1 << 28)
p.sample_begin, p.sample_end = 0, S // N
print(f"samples: every rank {timed(p):.1f} ms")
for band in (1, 4, 16, 135):
    ts = []
    for r in range(N):
        p = abi.default_params(w, h, spp=S, batch_samples=1 << 28, band_rows=band, shard_count=N, shard_index=r)
        ts.append(timed(p))
    print(f"tiles, bands of {band} rows: slowest rank {max(ts):.1f} ms, fastest {min(ts):.1f}, mean {sum(ts) / N:.1f}")
