#!/usr/bin/env python3
"""Soak run of the randomised GPU-vs-oracle comparison (tests/test_gpu_fuzz.py::check_scene) over a range
of seeds that the test suite does not use; prints every seed that fails and why, and a summary.
usage (on a GPU box): python tools/fuzz_soak.py FIRST LAST"""
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding  # the checker (this is a test tool)
from path_tracer_amd.raytracer import Context
from tests import test_gpu_fuzz as F


def main():
    first, last = int(sys.argv[1]), int(sys.argv[2])
    binding.build()
    gpu = Context(0)
    bad = []
    t0 = time.time()
    for seed in range(first, last + 1):
        try:
            kw = {}
            if seed % 7 == 0:
                kw = dict(n_lights=5 + seed % 9)
            F.check_scene(F.random_scene(seed, **kw), seed, gpu, binding, n_rays=4000, n_fire=20000, frame=(64, 48))
        except AssertionError as e:
            bad.append(seed)
            print(f"seed {seed}: {e}", flush=True)
        except Exception:
            bad.append(seed)
            print(f"seed {seed}: error\n{traceback.format_exc()}", flush=True)
        if (seed - first) % 20 == 19:
            print(f"... {seed - first + 1} seeds, {len(bad)} failed, {time.time() - t0:.0f} s", flush=True)
    print(f"{last - first + 1} seeds, failed: {bad}")
    gpu.close()


if __name__ == "__main__":
    main()
