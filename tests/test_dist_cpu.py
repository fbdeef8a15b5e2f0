"""Multi-rank path on the CPU: two gloo ranks shard one frame (by sample range and by row
bands), fill their radiance sums, and the single collective of the render brings them
together on rank 0.  The per-rank sums come from the CPU oracle here (no GPU in this
test); the sharding plan and the collective are the product code under test."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, mode, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    from oracle.binding import Oracle
    from path_tracer_amd import abi
    from path_tracer_amd.dist import combine, shard_params
    from tests.conftest import load_scene

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = load_scene("areaLight")
        base = abi.default_params(20, 14, spp=6, band_rows=3)
        p = shard_params(base, rank, world, mode)
        mean = Oracle(sc).render(p)                       # zero outside the shard; mean = sum / spp
        d_sum = torch.from_numpy((mean * base.spp).reshape(-1).copy())
        full = combine(d_sum, base, mode, dst=0)
        if rank == 0:
            np.save(out_path, full.numpy().reshape(14, 20, 3) / base.spp)
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["samples", "tiles"])
def test_two_ranks_reproduce_the_single_rank_frame(mode, tmp_path):
    import torch.multiprocessing as mp

    from oracle.binding import Oracle
    from path_tracer_amd import abi
    from tests.conftest import load_scene

    out = str(tmp_path / "frame.npy")
    port = 29500 + (os.getpid() % 2000) + (0 if mode == "samples" else 1)
    mp.spawn(_worker, args=(2, port, mode, out), nprocs=2, join=True)
    got = np.load(out)
    want = Oracle(load_scene("areaLight")).render(abi.default_params(20, 14, spp=6, band_rows=3))
    if mode == "tiles":
        assert np.array_equal(got, want)                  # disjoint rows: bit for bit
    else:
        assert np.allclose(got, want, rtol=0, atol=1e-15)  # fp64 summation order only


def test_shard_plans_partition_the_work():
    from path_tracer_amd import abi
    from path_tracer_amd.dist import owned_rows, shard_params

    base = abi.default_params(64, 37, spp=10, band_rows=4)
    for world in (1, 2, 3, 8):
        rows = np.concatenate([owned_rows(shard_params(base, r, world, "tiles")) for r in range(world)])
        assert sorted(rows.tolist()) == list(range(37))
        spans = [(p.sample_begin, p.sample_end) for p in (shard_params(base, r, world, "samples") for r in range(world))]
        assert spans[0][0] == 0 and spans[-1][1] == 10 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    with pytest.raises(ValueError):
        shard_params(base, 2, 2, "tiles")


def test_bench_starts_its_own_ranks():
    """`bench.py --gpus 2` started by itself (no launcher, WORLD_SIZE unset) runs two ranks that meet, and its
    stdout is exactly one JSON line that says n_gpus = 2; --rendezvous-only stops before any GPU work."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--rendezvous-only"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = res.stdout.splitlines()
    assert len(lines) == 1, res.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["warmup"] == 2


def test_bench_passes_a_failing_rank_on():
    """A rank that fails makes the self-launched bench fail too (here: no GPU in this container)."""
    import subprocess

    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload", "spheres"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0
    assert res.stdout.strip() == ""


def test_bench_call_plans_cover_the_job_exactly_once():
    """bench.py --gpus N: whatever the split, the ranks' render calls together cover every sample of the timed job
    exactly once (samples) or once per rank's rows (tiles), and no call is larger than one GPU's batch of S samples
    per pixel (tiles: N steps' samples for a N-th of the rows)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for K, W, S in ((20, 5, 128), (3, 1, 128), (5, 0, 16), (1, 1, 2), (7, 2, 3)):
        job = set(range(W * S, (W + K) * S))
        for world in (1, 2, 3, 4, 8):
            for split_each_step in (False, True):
                seen = []
                for rank in range(world):
                    calls = bench.plan_calls(K, W, S, world, rank, split_each_step=split_each_step)
                    assert all(0 < b - a <= S for a, b in calls)
                    seen += [s for a, b in calls for s in range(a, b)]
                assert len(seen) == len(job) and set(seen) == job, (K, W, S, world, split_each_step)
                if world > 1 and not split_each_step and K * S >= world:
                    # a rank's share comes in calls as large as the batch allows: at most one of them is short
                    calls = bench.plan_calls(K, W, S, world, 0)
                    assert sum(1 for a, b in calls if b - a < S) <= 1
            for rank in range(world):  # tiles: every rank renders the whole range for its rows
                calls = bench.plan_calls(K, W, S, world, rank, tiles=True)
                assert sorted(s for a, b in calls for s in range(a, b)) == sorted(job)
                assert all(b - a <= world * S for a, b in calls)
            # weak scaling: K * world * S samples, every rank K calls of S
            seen = [s for rank in range(world) for a, b in bench.plan_calls(K, W, S, world, rank, weak=True) for s in range(a, b)]
            assert sorted(seen) == list(range(W * S * world, (W + K) * S * world))
