"""The benchmark's one JSON line carries the fields its consumers read: the headline metric, the timing,
the workload, and the two objects `roofline` and `cpu_baseline`.  Runs bench.py itself (a child process)
on the smallest workload with a small step, so a change that breaks the line is caught by the test
suite and not by the harness that runs the benchmark."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), *args]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line, got {len(lines)}"
    return json.loads(lines[0])


def test_the_bench_line_has_its_fields():
    j = run_bench("--workload", "spheres", "--steps", "2", "--warmup", "1", "--spp-per-step", "4", "--cpu-budget", "1")
    assert j["metric"] == "path samples/s" and j["unit"] == "Msamples/s" and j["higher_is_better"] is True
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1
    assert j["dtype"] == "f64" and j["data"] == "synthetic" and j["vs_baseline"] is None
    assert j["scaling"] in ("strong", "weak")
    assert j["value"] > 0 and j["ms_per_step"] > 0
    # value is the whole job's samples over the timed steps
    samples = 512 * 512 * 4 * 2
    assert j["value"] == pytest.approx(samples / (j["ms_per_step"] * 2 * 1e-3) / 1e6, rel=1e-6)
    cfg = j["config"]
    assert "workload" in cfg and "model" not in cfg and cfg["spp_per_step"] == 4
    roof = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms"):
        assert key in roof, key
    assert roof["bound"] in ("hbm", "valu", "vmem_address") and roof["peak"] > 0 and roof["avg_launch_ms"] > 0
    assert 0 < roof["frac"] <= 1.0 and roof["vmem_address"]["g_lane_loads_per_s"] > 0 and roof["valu"]["peak_all_slow_class"] > 500
    cb = j["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    # the bench renders its own workload on both sides once: parity of the measured thing
    par = j["parity"]
    assert par["pixels_over_1e-5"] <= 0.002 * par["pixels"] and par["rmsd"] < 1e-4
    assert isinstance(j["frame_checksum"], int) and j["frame_checksum"] > 0
    assert len(j["kernel_source_hash"]) == 16


def test_counters_of_other_sources_are_not_quoted(tmp_path):
    """profiles/counters.json is stamped with the kernel sources it was measured on; bench.py reports
    traffic and utilisation only when the stamp matches the sources it runs."""
    sys.path.insert(0, ROOT)
    import bench
    from path_tracer_amd.buildinfo import kernel_source_hash

    with open(os.path.join(ROOT, "profiles", "counters.json")) as fh:
        table = json.load(fh)
    for workload, entry in table.items():
        got = bench.load_counters(workload, "k_shadow_any")
        if entry.get("kernel_source_hash") == kernel_source_hash():
            assert got is not None and got["valu_instructions_per_launch"] > 0
        else:
            assert got is None
    assert bench.load_counters("no such workload", "k_shadow_any") is None
