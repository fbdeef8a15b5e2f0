#!/usr/bin/env python3
"""Probe: does running K contexts concurrently (K host threads, K streams, each a slice of the
step's samples) beat one context on the whole step?  usage: overlap_probe.py [K ...]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from path_tracer_amd import abi, synth
from path_tracer_amd.raytracer import Context

sc = synth.make_scene(1_000_000, seed=0xC4)
w, h, S = 1920, 1080, 32
for K in [int(x) for x in sys.argv[1:]] or [1, 2, 3, 4]:
    ctxs = [Context(0) for _ in range(K)]
    for c in ctxs:
        c.upload_scene(sc)
    sums = [torch.zeros(3 * w * h, dtype=torch.float64, device="cuda:0") for _ in range(K)]
    streams = [torch.cuda.Stream() for _ in range(K)]
    def work(k, first):
        p = abi.default_params(w, h, spp=4096, batch_samples=(1 << 26) // K)
        p.sample_begin, p.sample_end = first + k * S // K, first + (k + 1) * S // K
        ctxs[k].render_accumulate(p, sums[k].data_ptr(), streams[k].cuda_stream)
    def step(first):
        ts = [threading.Thread(target=work, args=(k, first)) for k in range(K)]
        for t in ts: t.start()
        for t in ts: t.join()
        torch.cuda.synchronize()
    step(0)
    t0 = time.perf_counter()
    for i in range(1, 3):
        step(i * S)
    dt = (time.perf_counter() - t0) / 2
    print(f"K={K}: {dt*1e3:.1f} ms per 32-spp step, {w*h*S/dt/1e6:.1f} Msamples/s", flush=True)
    for c in ctxs:
        c.close()
    del sums
