#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One *step* = one pass of the wavefront pipeline (generate -> intersect -> shade ->
shadow -> ... -> resolve) over `--spp-per-step` samples of every pixel of the frame,
with scene, queues and the radiance sum resident in HBM.  Metric: path samples/s
(one path sample = one `trace(x, y)` call of the reference, RayTracer.cpp:38-56:
a camera ray plus its whole tree of bounce, mirror and shadow rays).

With N > 1 (one rank per GPU: launched by torch.distributed.run, or by bench.py itself when it is
started without a launcher) the SAME job — the K steps — is split N ways (strong scaling): every rank renders a
contiguous N-th of the job's sample range for the whole frame, in calls of at most S samples per pixel (or, with
--sharding tiles, all of the job's samples for its own bands of rows), and the per-pixel sums are brought to
rank 0 over RCCL once, inside the timed region, where rank 0 also resolves them to the 8-bit frame and copies
it to the host (what RayTracer::traceImage + getBuffer hand back).  --split-each-step splits every step N
ways instead (calls of S / N samples per pixel); --scaling weak keeps per-GPU work fixed (every rank renders
K whole steps).

Rank 0 prints one JSON line.  `roofline` describes the kernel with the most device time (roofline() below:
three fractions against measured ceilings, the largest names the bound); `cpu_baseline` is the fp64 CPU
restatement of the reference timed on sampled rows of the same frame, whose pixels also check the GPU's
(`parity`).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
INFINITY_CACHE_BYTES = 256 << 20
# Measured rulers of the chip (tools/micro/valu_issue.hip -> profiles/r03_valu_issue.json; tools/micro/gather_bench.hip ->
# profiles/r03_gather_bench.txt), at the shader clock those programs read inside their kernels under load:
N_SIMD, N_CU, SHADER_GHZ = 1024, 256, 2.39
VALU_CYC_ISSUE = 2.2   # cycles a wave64 vector instruction holds its SIMD's issue port (v_fma_f32, v_mov_b32, v_and_b32 ... alone: 2.2-2.4)
VALU_CYC_SLOW = 4.13   # cycles most opcodes (fp64, conversions, min/max, compares, selects ...) also hold the narrower pipe; the two overlap across waves
VMEM_CYC_LANE_LOAD = 1.05        # cycles of a CU's vector-L1 path per 16-byte lane-load when every lane reads a 64-byte record of its own (269 per 256)
VMEM_CYC_LANE_LOAD_RUNS10 = 0.52  # the same when runs of ten adjacent lanes read the same record (134 per 256)


def valu_peak_ginst(cycles_per_instruction):
    return N_SIMD * SHADER_GHZ / cycles_per_instruction


WORKLOADS = {
    # name: (triangles, width, height, area lights, cubemap, nominal spp)
    "synth1m": (1_000_000, 1920, 1080, 1, False, 4096),
    "synth10m": (10_000_000, 3840, 2160, 64, True, 8192),
    "synth100k": (100_000, 1920, 1080, 1, False, 4096),
    # the 1 M scene without the benchmark's rough terrain: glossy ground, path trees multiply (run with --max-depth 8)
    "synth1m_glossy_terrain": (1_000_000, 1920, 1080, 1, False, 4096),
}
FIXTURE_WORKLOADS = {
    # name: (fixture, width, height, nominal spp)
    "spheres": ("spheres", 512, 512, 100),
    "cornell": ("cornellBoxes", 1024, 1024, 1024),
    "hitchcock": ("hitchcockBRDF", 1920, 1080, 2048),
    # authored variants: area light instead of the point light; boxes as triangle meshes
    "cornell_area": ("cornellBoxes", 1024, 1024, 1024),
    "hitchcock_mesh": ("hitchcockBRDF", 1920, 1080, 2048),
}


def build_scene(name, triangles=None):
    from path_tracer_amd import synth
    from path_tracer_amd.scene import Scene

    if name in WORKLOADS:
        tris, w, h, nl, cube, spp = WORKLOADS[name]
        if triangles:
            tris = triangles
        sc = synth.make_scene(tris, seed=0xC4 if name != "synth10m" else 0xC5, n_area_lights=nl, cubemap=cube, rough_terrain=name != "synth1m_glossy_terrain")
        autumn = os.path.join(ROOT, "tests", "golden", "cubemaps", "autumn.npz")
        if cube and os.path.exists(autumn):  # the reference's cubemap_autumn (box-filtered to 128 x 128) instead of the procedural sky
            z = np.load(autumn)
            sc.cubemap = [z[f"cube{k}"] for k in range(6)]
        label = f"synthetic {sc.n_triangles()}-triangle mesh scene, {nl} area light(s) + 1 directional, {w}x{h}"
        return sc, w, h, spp, label
    fx, w, h, spp = FIXTURE_WORKLOADS[name]
    sc = Scene.load_npz(os.path.join(ROOT, "tests", "golden", "scenes", fx + ".npz"))
    if name == "cornell_area":
        return synth.cornell_area_light(sc), w, h, spp, f"{fx} with a rectangular area light in place of the point light, {w}x{h}"
    if name == "hitchcock_mesh":
        return synth.tessellate_boxes(sc), w, h, spp, f"{fx} with its 140 boxes as triangle meshes, {w}x{h}"
    return sc, w, h, spp, f"{fx} (reference demo scene), {w}x{h}"


def algorithmic_bytes(c, which):
    """Layout-independent accounting of the bytes a kernel's work needs: 64 B per BVH2
    node visit, 48 B per primitive test, 160 B per ray-state record handled by a stage."""
    if which == "shadow":
        return 64 * c["shadow_node_visits"] + 48 * c["shadow_prim_tests"] + 160 * c["shaded_records"]
    if which == "intersect":
        return 64 * (c["node_visits"] - c["shadow_node_visits"]) + 48 * (c["prim_tests"] - c["shadow_prim_tests"]) + 160 * c["path_rays"]
    raise KeyError(which)


def cpu_baseline(scene, base_params, budget_s=20.0):
    """Time the CPU oracle (a port of the reference's algorithm, un-pruned two-level BVH,
    fp64, std::thread column chunks) on a bounded sample of the same workload: every
    M-th row of the frame at 1 spp, M chosen so the run takes roughly `budget_s`."""
    import copy

    from oracle.binding import Oracle

    threads = min(16, os.cpu_count() or 1)
    orc = Oracle(scene)
    p = copy.copy(base_params)
    p.spp, p.sample_begin, p.sample_end = 1, 0, 1
    p.band_rows = 1
    # calibrate on one row
    p.shard_count, p.shard_index = p.height, p.height // 3
    t0 = time.time()
    orc.render(p, threads=threads)
    dt = max(time.time() - t0, 1e-3)
    rows = int(max(1, min(p.height, budget_s / dt)))
    p.shard_count = max(1, p.height // rows)
    p.shard_index = 0
    n_rows = len(range(0, p.height, p.shard_count))
    t0 = time.time()
    img = orc.render(p, threads=threads)
    dt = time.time() - t0
    samples = n_rows * p.width
    return {
        "value": samples / dt / 1e6,
        "unit": "Msamples/s",
        "cores": threads,
        "kind": "port",
        "sample": f"rows 0, {p.shard_count}, {2 * p.shard_count}, ... ({n_rows} rows x {p.width} px) of the frame at 1 spp, {dt:.1f} s",
    }, p, img


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="synth1m", choices=sorted(list(WORKLOADS) + list(FIXTURE_WORKLOADS)))
    ap.add_argument("--spp-per-step", type=int, default=0,
                    help="samples per pixel in one step (of the whole job, all ranks together); 0 = as many as the free HBM of one "
                         "GPU allows (128, 64 or 16 at 1080p)")
    ap.add_argument("--triangles", type=int, default=0)
    ap.add_argument("--batch-samples", type=int, default=0, help="samples per wavefront batch; 0 = one batch per step and rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="with --no-cpu-baseline: skip the oracle's check of sampled rows too (tuning sweeps)")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--cull-negligible", action="store_true",
                    help="set MIPT_FLAG_CULL_NEGLIGIBLE: skip shadow rays whose whole term is below 2^-52 (off: every shadow "
                         "ray the reference traces is traced)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: 'strong' = the N = 1 job split N ways (default); 'weak' = every rank renders a whole N = 1 step")
    ap.add_argument("--sharding", default="samples", choices=["samples", "tiles"],
                    help="N > 1: 'samples' = every rank renders the whole frame for its own samples (reduce); "
                         "'tiles' = every rank renders all samples of its own bands of rows (gather; strong scaling only)")
    ap.add_argument("--split-each-step", action="store_true",
                    help="N > 1, strong scaling: split every step N ways (calls of S / N samples per pixel) instead of the job")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo rehearses the N>1 path with several ranks on one GPU (RCCL needs one GPU per rank)")
    ap.add_argument("--max-depth", type=int, default=0, help="0 = the library's default (64)")
    ap.add_argument("--no-profile-events", action="store_true",
                    help="time the steps without the per-kernel HIP events (then no kernel_ms_per_step and no roofline)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launcher check that needs no GPU: the ranks meet over gloo and rank 0 prints the line's envelope (no render)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started by itself with N > 1: the split into ranks lives inside the call, as the reference's split into
        # threads lives inside traceImage (RayTracer.cpp:482-489).  The ranks are children of this process, started
        # before anything here has touched the GPU; rank 0's line goes to this process's stdout.
        raise SystemExit(launch_ranks(args.gpus))

    if args.rendezvous_only:
        import torch.distributed as dist

        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world > 1:
            dist.init_process_group(backend="gloo")
            dist.barrier()
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"metric": "path samples/s", "value": None, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "rendezvous_only": True}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    import torch
    import torch.distributed as dist

    from path_tracer_amd import abi
    from path_tracer_amd.dist import combine, shard_params
    from path_tracer_amd.raytracer import Context

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s): use --nproc-per-node {args.gpus}, or start bench.py by itself")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    weak = args.scaling == "weak" and world > 1
    tiles = args.sharding == "tiles" and world > 1
    if weak and tiles:
        raise SystemExit("--scaling weak shards the sample range; tiles are a strong-scaling split")
    mode = "tiles" if tiles else "samples"

    scene, w, h, nominal_spp, label = build_scene(args.workload, args.triangles)
    ctx = Context(local_rank)
    t0 = time.time()
    ctx.upload_scene(scene)
    upload_s = time.time() - t0
    stats = ctx.scene_stats()

    # One step = S samples of every pixel.  The deep, nearly empty levels of a batch cost the same
    # ~16 ms whatever its size, so S is as large as the free HBM of one GPU allows (about 700 B per
    # sample with the shadow-ray scratch); every rank takes the smallest choice of any rank.  S does
    # not depend on N: with N ranks the job is the same K steps of S samples (strong scaling).
    S = args.spp_per_step
    if S <= 0:
        free_b = torch.cuda.mem_get_info(local_rank)[0]
        if world > 1:
            t = torch.tensor([free_b], dtype=torch.int64, device=f"cuda:{local_rank}" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            free_b = int(t.item())
        S = 1
        for cand in (128, 64, 32, 16, 8, 4, 2):
            if cand * w * h * 700 + (40 << 30) <= free_b and cand * w * h <= (1 << 28):
                S = cand
                break
    if args.split_each_step and not weak and not tiles and S % world:
        S = max(world, S - S % world)  # every rank gets the same number of samples of every step
    step_spp = S * world if weak else S  # samples per pixel one step adds to the frame, all ranks together
    K, W = args.steps, args.warmup
    # The timed job: K steps = samples [W * step_spp, (W + K) * step_spp) of every pixel.  One rank renders it step by
    # step, one call per step.  N ranks split THE JOB, as mipt_dist_render and dist.shard_params split a frame: by samples,
    # rank r takes a contiguous N-th of the job's sample range and renders it in calls of at most S samples per pixel (the
    # batch one GPU holds); by tiles, it takes its bands of rows for the whole range, N steps' samples to a call (again one
    # full batch).  Splitting every step N ways instead (--split-each-step, what this script did before) hands each rank
    # calls of S / N samples per pixel, whose narrow tree levels cost what a full batch's do: 93.3 ms per 16 spp call
    # against 661 / 8 = 82.7 on the benchmark scene.  Weak scaling: every rank renders K whole steps.
    job_lo, job_hi = W * step_spp, (W + K) * step_spp
    shard_tiles = tiles
    calls = plan_calls(K, W, S, world, rank, tiles=tiles, weak=weak, split_each_step=args.split_each_step)
    call_spp = max([b - a for a, b in calls] + [1])
    pixels_mine = (w * h + world - 1) // world if shard_tiles else w * h
    per_call_samples = call_spp * pixels_mine
    batch = args.batch_samples if args.batch_samples > 0 else min(1 << 28, max(1 << 20, per_call_samples))
    total_spp = job_hi + 2 * max(call_spp, step_spp) + step_spp  # room behind the job for the untimed plain and counter calls
    base = abi.default_params(w, h, spp=total_spp, batch_samples=batch)
    if args.max_depth > 0:
        base.max_depth = args.max_depth
    dev = f"cuda:{local_rank}"
    d_sum = torch.zeros(3 * w * h, dtype=torch.float64, device=dev)
    d_rgb8 = torch.zeros(3 * w * h, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    extra_flags = abi.FLAG_CULL_NEGLIGIBLE if args.cull_negligible else 0

    def call_params(begin, end, flags):
        p = abi.default_params(w, h, spp=total_spp, batch_samples=batch, sample_begin=begin, sample_end=end)
        p.max_depth = base.max_depth
        if shard_tiles:
            p.shard_index, p.shard_count = rank, world
        p.flags = flags | extra_flags
        return p

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def finish(spp_done):
        """The collective, then on rank 0 what traceImage + getBuffer hand back: the 8-bit frame on the host."""
        full = combine(d_sum, base, mode, dst=0)
        if rank == 0:
            ctx.resolve(full.data_ptr(), w, h, spp_done, 0, d_rgb8.data_ptr(), stream)
            return d_rgb8.cpu()
        return None

    for s in range(W):  # untimed: calls of the size of the timed ones
        ctx.render_accumulate(call_params(0, call_spp, 0), d_sum.data_ptr(), stream)
    if W > 0:
        finish(W * step_spp)  # once, untimed: RCCL sets up its channels and buffers on first use
        d_sum.zero_()
    ctx.reset_counters()
    prof_flag = 0 if args.no_profile_events else abi.FLAG_PROFILE
    sync()
    t0 = time.perf_counter()
    for a, b in calls:
        ctx.render_accumulate(call_params(a, b, prof_flag), d_sum.data_ptr(), stream)
    frame8 = finish(K * step_spp)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ktimes = ctx.kernel_times()
    n_calls = max(1, len(calls))

    # one more call without the per-kernel events, timed by itself: what the events cost
    scratch = torch.zeros_like(d_sum)
    sync()
    t1 = time.perf_counter()
    ctx.render_accumulate(call_params(job_hi, job_hi + call_spp, 0), scratch.data_ptr(), stream)
    torch.cuda.synchronize()
    plain_step_ms = (time.perf_counter() - t1) * 1e3

    samples_total = K * step_spp * w * h
    if rank == 0:
        # one instrumented (untimed) step for the work counters; the work is deterministic
        ctx.reset_counters()
        scratch.zero_()
        ctx.render_accumulate(call_params(job_hi + call_spp, job_hi + 2 * call_spp, abi.FLAG_COUNTERS), scratch.data_ptr(), stream)
        torch.cuda.synchronize()
        cnt = ctx.counters()
        del scratch
        step_samples = max(1, cnt["samples"])  # one call of rank 0 (N = 1: one step)
        cnt["shaded_records"] = cnt["shadow_rays"] // max(1, scene_rays_per_hit(scene, base.area_samples))
        out = {
            "metric": "path samples/s",
            "value": samples_total / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if weak else "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": label,
                "spp_per_step": step_spp,
                "calls_per_rank": n_calls,
                "spp_per_call": call_spp,
                "job_split": ("none" if world == 1 else "every rank renders K whole steps" if weak else
                              "every step split N ways" if args.split_each_step else
                              "the job's sample range in contiguous shares, one per rank" if not tiles else
                              "every rank renders its bands of rows for the whole job, N steps' samples to a call"),
                "batch_samples": int(batch),
                "nominal_spp": nominal_spp,
                "rr_q": base.rr_q,
                "area_samples": base.area_samples,
                "max_depth": base.max_depth,
                "primitives": stats["prims"],
                "bvh_nodes": stats["nodes"],
                "scene_bytes": stats["device_bytes"],
                "sharding": mode if world > 1 else "none",
                "cull_negligible": bool(args.cull_negligible),
                "timed_region": "K x mipt_render_accumulate + the collective + mipt_resolve to RGB8 + its copy to the host",
            },
            "frame_checksum": int(frame8.to(torch.int64).sum().item()),
            "per_kernel_events_cost": {"ms_per_call_with_events": elapsed / n_calls * 1e3, "ms_of_one_call_without": plain_step_ms},
            "per_sample": {
                "path_rays": cnt["path_rays"] / step_samples,
                "shadow_rays": cnt["shadow_rays"] / step_samples,
                "shaded_hits": cnt["shaded_hits"] / step_samples,
                "node_visits": cnt["node_visits"] / step_samples,
                "prim_tests": cnt["prim_tests"] / step_samples,
            },
            "counters": cnt,
            "mrays_per_s": (cnt["path_rays"] + cnt["shadow_rays"]) / step_samples * samples_total / elapsed / 1e6,
            "scene_upload_s": upload_s,
            "kernel_source_hash": kernel_source_hash(),
        }
        if not args.no_profile_events:
            # per call of rank 0 (N = 1: per step); the counters above are one such call's
            out["roofline"] = roofline(args.workload, stats["device_bytes"], ktimes, n_calls, cnt, step_samples)
            out["kernel_ms_per_step"] = {k: v["ms"] / n_calls for k, v in ktimes.items()}
            out["kernel_launches_per_step"] = {k: v["launches"] / n_calls for k, v in ktimes.items()}
        if world == 1 and not (args.no_cpu_baseline and args.no_parity):
            # without the baseline leg the oracle still checks a few sampled rows of the workload (a shorter run of the same routine)
            cb, p_cpu, img_cpu = cpu_baseline(scene, base, args.cpu_budget if not args.no_cpu_baseline else min(args.cpu_budget, 4.0))
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cb
            # the same rows on the GPU, same random streams: parity of the bench workload itself
            p_gpu = abi.default_params(w, h, spp=1, band_rows=1, shard_count=p_cpu.shard_count, shard_index=0)
            p_gpu.max_depth = base.max_depth
            f32, _ = ctx.render(p_gpu, want_f32=True, want_rgb8=False)
            rows = np.arange(0, h, p_cpu.shard_count)
            diff = f32[rows].astype(np.float64) - img_cpu[rows]
            out["parity"] = {
                "rmsd": float(np.sqrt(np.mean(diff * diff))),
                "max_abs": float(np.abs(diff).max()),
                "pixels_over_1e-5": int((np.abs(diff).max(axis=2) > 1e-5).sum()),
                "pixels": int(diff.shape[0] * diff.shape[1]),
            }
            out["parity"]["sample"] = cb["sample"]
            if not args.no_cpu_baseline:
                out["vs_cpu_baseline"] = out["value"] / cb["value"] if cb["value"] > 0 else None
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def plan_calls(K, W, S, world, rank, tiles=False, weak=False, split_each_step=False):
    """The sample ranges [begin, end) of the render calls `rank` makes for the timed job of K steps of S samples per
    pixel (weak scaling: S per rank and step) that follows W warm-up steps.  Samples mode: the ranges of all ranks
    partition the job's range; tiles mode: every rank covers the whole range (for its own rows), N steps to a call."""
    step_spp = S * world if weak else S
    job_lo, job_hi = W * step_spp, (W + K) * step_spp
    if world == 1 or weak or split_each_step:
        if weak:
            calls = [(job_lo + k * step_spp + rank * S, job_lo + k * step_spp + (rank + 1) * S) for k in range(K)]
        elif tiles or world == 1:
            calls = [(job_lo + k * S, job_lo + (k + 1) * S) for k in range(K)]
        else:
            calls = [(job_lo + k * S + (S * rank) // world, job_lo + k * S + (S * (rank + 1)) // world) for k in range(K)]
    elif tiles:
        calls = [(c, min(c + world * S, job_hi)) for c in range(job_lo, job_hi, world * S)]
    else:
        lo, hi = job_lo + (K * S * rank) // world, job_lo + (K * S * (rank + 1)) // world
        calls = [(c, min(c + S, hi)) for c in range(lo, hi, S)]
    return [c for c in calls if c[1] > c[0]]


def launch_ranks(n):
    """Runs this script as n ranks of one node (torch.distributed.run, one process per GPU, rendezvous on
    127.0.0.1 at a free port) with the same arguments, and returns the launcher's exit status."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    # the ranks' stdout is passed on line by line: rank 0's JSON line to stdout, anything else (gloo announces its
    # connections on stdout) to stderr, so that stdout carries the one line and nothing else
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in proc.stdout:
        dst = sys.stdout if line.lstrip().startswith("{") else sys.stderr
        dst.write(line)
        dst.flush()
    return proc.wait()


def kernel_source_hash():
    from path_tracer_amd.buildinfo import kernel_source_hash as h

    return h()


def scene_rays_per_hit(scene, area_samples):
    from path_tracer_amd import abi

    return sum(area_samples if L.type == abi.LIGHT_RECT_AREA else 1 for L in scene.lights)


def load_counters(workload, kernel, step_samples=None):
    """The rocprofv3 counter summary of `kernel` on `workload` (profiles/counters.json, written by
    tools/make_counters.py from separate --pmc passes over one bench step), or None when there is
    none, when it was measured on other kernel sources than the ones this run uses, or (step_samples
    given) on a step of another size: its figures are per launch, and a launch is as large as the step."""
    try:
        with open(os.path.join(ROOT, "profiles", "counters.json")) as fh:
            entry = json.load(fh).get(workload)
    except (OSError, ValueError):
        return None
    if not entry or entry.get("kernel_source_hash") != kernel_source_hash():
        return None
    if step_samples is not None and entry.get("step_samples") != step_samples:
        return None
    return entry.get("kernels", {}).get(kernel)


def roofline(workload, scene_bytes, ktimes, K, cnt, step_samples):
    """The roof that binds the dominant kernel (the one with the most device time), against MEASURED peaks.

    Three candidates.  (i) Vector-instruction issue: SQ_INSTS_VALU per launch / launch time against N_SIMD x clock /
    cycles-per-instruction, where cycles-per-instruction = max(2.2, 4.13 x share of slow-class opcodes) is what the issue
    micro-benchmark gives for the instruction mix of the kernel's descend loop (tools/valu_mix.py, the census of the compiled
    kernel).
    (ii) The vector-L1 address path: the share of cycles the vector L1 (TCP) was busy (counter), beside the rate of 16-byte
    lane-loads (four per node visit, three to six per primitive test, from this run's work counters) and what the gather
    micro-benchmark sustains.  (iii) HBM: HBM-side bytes per launch / launch time against 8 TB/s.  The largest fraction is
    `bound`.  Launch durations are this run's own (HIP events on the launch stream around every launch of the timed steps);
    instruction counts, busy shares and HBM-side bytes come from the committed counter passes of the same kernel sources
    over a step of the same size, and are null otherwise.  The algorithmic-byte rate is always reported as
    `cache_served_gbs`; it is a roofline fraction only where HBM is the bound."""
    dominant = max(("intersect", "shadow"), key=lambda k: ktimes[k]["ms"])
    name = "k_shadow_any" if dominant == "shadow" else "k_intersect"
    kt = ktimes[dominant]
    if not kt["launches"]:
        return None
    launches_step = kt["launches"] / K
    avg_launch_s = kt["ms"] / kt["launches"] * 1e-3
    alg_bytes_launch = algorithmic_bytes(cnt, dominant) / max(1.0, launches_step)
    cache_served = alg_bytes_launch / avg_launch_s / 1e9
    pmc = load_counters(workload, name, step_samples)
    if pmc and abs(pmc.get("launches_per_step", launches_step) - launches_step) > 0.05 * launches_step:
        pmc = None  # another launch sequence than the one the passes averaged over
    traffic = pmc.get("hbm_bytes_per_launch") if pmc else None
    hbm_frac = traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS if traffic else None
    # lane-loads of the walk, from this run's own work counters (per launch)
    if dominant == "shadow":
        nodes, prims = cnt["shadow_node_visits"], cnt["shadow_prim_tests"]
    else:
        nodes, prims = cnt["node_visits"] - cnt["shadow_node_visits"], cnt["prim_tests"] - cnt["shadow_prim_tests"]
    ll_low, ll_high = (4 * nodes + 3 * prims) / max(1.0, launches_step), (4 * nodes + 6 * prims) / max(1.0, launches_step)
    ll_rate = 0.5 * (ll_low + ll_high) / avg_launch_s / 1e9
    vmem = {
        "lane_loads_per_launch": [ll_low, ll_high],
        "g_lane_loads_per_s": ll_rate,
        "peak_every_lane_its_own_record": N_CU * SHADER_GHZ / VMEM_CYC_LANE_LOAD,
        "peak_runs_of_ten_lanes_per_record": N_CU * SHADER_GHZ / VMEM_CYC_LANE_LOAD_RUNS10,
        "tcp_busy": pmc.get("tcp_busy") if pmc else None,
        "ta_busy": pmc.get("ta_busy") if pmc else None,
        "vmem_read_instructions_per_launch": pmc.get("vmem_read_instructions_per_launch") if pmc else None,
    }
    inst = pmc.get("valu_instructions_per_launch") if pmc else None
    mix = pmc.get("valu_mix") if pmc else None
    valu_rate = inst / avg_launch_s / 1e9 if inst else None
    # the mix of the descend loop, where a walk spends its instructions; the census of the whole persistent loop counts the
    # ray set-up (divisions, a square root) once per instruction like the loop a ray goes round a dozen times, and the kernel
    # issues faster than that mix would allow (config 5: 610 G/s against 606): it is kept beside as a lower bound of the ceiling
    loop = (mix or {}).get("descend_loop") or mix
    cpi = loop["cycles_per_instruction"] if loop else VALU_CYC_SLOW  # without a census: every instruction counted slow-class
    valu_frac = valu_rate / valu_peak_ginst(cpi) if valu_rate else None
    valu = {
        "g_wave_instructions_per_s": valu_rate,
        "peak_for_the_kernels_mix": valu_peak_ginst(cpi),
        "cycles_per_instruction_of_the_mix": cpi,
        "slow_class_share": loop["slow_class_share"] if loop else None,
        "peak_for_the_whole_loops_static_mix": valu_peak_ginst(mix["cycles_per_instruction"]) if mix else None,
        "peak_all_slow_class": valu_peak_ginst(VALU_CYC_SLOW),
        "peak_all_fast_class": valu_peak_ginst(VALU_CYC_ISSUE),
        "frac": valu_frac,
        "useful_frac": valu_frac * pmc["lane_utilisation"] if valu_frac and pmc.get("lane_utilisation") else None,
    }
    r = {
        "kernel": name,
        "avg_launch_ms": avg_launch_s * 1e3,
        "launches_per_step": launches_step,
        "cache_served_gbs": cache_served,
        "algorithmic_bytes_per_launch": alg_bytes_launch,
        "algorithmic_bytes_per_sample": (algorithmic_bytes(cnt, "intersect") + algorithmic_bytes(cnt, "shadow") + 64 * cnt["shaded_hits"]) / step_samples,
        "traffic": traffic,
        "hbm_frac": hbm_frac,
        "valu": valu,
        "vmem_address": vmem,
        "utilisation": {k: pmc[k] for k in ("lane_utilisation", "waves_per_simd", "waves_per_simd_compiled", "ta_busy", "tcp_busy",
                                            "l1_miss_rate", "l2_hit_rate", "wave_time_waiting_on_memory") if k in pmc} if pmc else None,
        "counters_from": pmc.get("measured") if pmc else None,
        "rulers": "profiles/r03_valu_issue.json, profiles/r03_gather_bench.txt (measured on MI355X at %.2f GHz in-kernel)" % SHADER_GHZ,
    }
    # which roof: the largest of the three fractions; without counters, the lane-load count against the gather rate for
    # scenes the caches hold, HBM beyond them
    vmem_frac = vmem["tcp_busy"] if vmem["tcp_busy"] is not None else min(1.0, ll_rate / vmem["peak_every_lane_its_own_record"])
    cands = {"vmem_address": vmem_frac, "valu": valu_frac or 0.0, "hbm": hbm_frac if hbm_frac is not None else (cache_served / HBM_PEAK_GBS if scene_bytes > INFINITY_CACHE_BYTES and not pmc else 0.0)}
    bound = max(cands, key=lambda k: cands[k])
    if bound == "valu":
        r.update({"bound": "valu", "achieved": valu_rate, "peak": valu["peak_for_the_kernels_mix"], "unit": "G wave-instructions/s", "frac": valu_frac,
                  "note": "vector-instruction issue: SQ_INSTS_VALU per launch / this run's launch time against the measured issue rate for the kernel's "
                          "instruction mix; the vector-L1 path (vmem_address.tcp_busy) is as close to its own roof"})
    elif bound == "vmem_address":
        if vmem["tcp_busy"] is not None:
            r.update({"bound": "vmem_address", "achieved": vmem["tcp_busy"], "peak": 1.0, "unit": "share of cycles the vector L1 is busy", "frac": vmem["tcp_busy"],
                      "note": "the vector-L1 address path: every lane gathers 16-byte words of its own node and triangle records; g_lane_loads_per_s beside the "
                              "gather micro-benchmark's rates says the same; vector-instruction issue (valu.frac) is within a few per cent of its own roof"})
        else:
            r.update({"bound": "vmem_address", "achieved": ll_rate, "peak": vmem["peak_every_lane_its_own_record"], "unit": "G 16-byte lane-loads/s", "frac": vmem_frac,
                      "note": "no counter pass for these kernel sources: lane-loads counted from the work counters (4 per node visit, 3-6 per primitive test) against "
                              "the gather micro-benchmark's rate for lanes that share nothing; lanes that share a line make the real demand smaller"})
    else:
        r.update({"bound": "hbm", "achieved": cache_served if traffic is None else traffic / avg_launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": cands["hbm"], "note": "HBM: bytes per launch / launch time against the HBM peak"})
    return r


if __name__ == "__main__":
    main()
