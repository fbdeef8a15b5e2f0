#!/usr/bin/env python3
"""Turns one round of rocprofv3 counter passes (tools/profile_round.sh) into the workload's entry
of profiles/counters.json, the file bench.py reads `roofline.traffic` / `utilisation` from.

usage: make_counters.py <passes dir> <workload> <tag> [counters.json to update]

Per timed kernel (the non-counting instantiations of k_intersect / k_shadow_any / k_shade / ...):
  hbm_bytes_per_launch  (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / launches.  Both counters are in KiB;
                   on gfx950 FETCH_SIZE tallies 128-byte read requests as 64 bytes, so the read side is
                   doubled (calibrated on wide streaming reads: an upper bound for gathers)
  valu_instructions_per_launch   SQ_INSTS_VALU / launches (wave-instructions)
  valu_busy        SQ_ACTIVE_INST_VALU x 4 / (kernel cycles x 1024 SIMDs); kernel cycles = the library's
                   HIP-event time of the same (profiled) pass at 2.4 GHz
  lane_utilisation SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): share of VALU lanes doing work
  waves_per_simd   SQ_WAVE_CYCLES x 4 / (kernel cycles x 1024)
  wave_time_waiting_on_memory   SQ_WAIT_ANY / SQ_WAVE_CYCLES
  ta_busy, tcp_busy  TA_TA_BUSY_sum, TCP_GATE_EN1_sum / (256 CUs x kernel cycles)
  l1_miss_rate     TCP_TCC_READ_REQ_sum / TCP_TOTAL_CACHE_ACCESSES_sum
  l2_hit_rate      TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
The entry carries the hash of the kernel sources it was measured on (path_tracer_amd/buildinfo.py) and
the size of the step the passes ran (step_samples, launches_per_step): bench.py quotes it only for a run
of the same sources and the same step.
"""
import csv, glob, json, os, re, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from path_tracer_amd.buildinfo import kernel_source_hash  # noqa: E402

CLOCK, SIMDS, CUS = 2.4e9, 1024, 256
KEY = {"k_intersect": "intersect", "k_shadow_any": "shadow", "k_shadow_walk": "shadow", "k_shade": "shade", "k_shadow_sum": "shadow_sum",
       "k_shadow_sum_lights": "shadow_sum", "k_shadow_sum_tiled": "shadow_sum", "k_shadow_sum_bytes": "shadow_sum", "k_generate": "generate", "k_resolve": "resolve"}


LEAN_SHADOW = set()  # filled by kernel_of: the shadow kernel without the analytic tests (6 waves per SIMD) was among the launches
INSTANCES = defaultdict(lambda: defaultdict(int))  # kernel -> template arguments -> dispatches seen (which instantiation the workload runs)


def kernel_of(name):
    m = re.search(r"(k_[a-z_]+)(<([^>]*)>)?", name)
    if not m:
        return None
    if m.group(1) == "k_shadow_any" and m.group(3) and [a.strip() for a in m.group(3).split(",")][3:4] == ["true"]:
        LEAN_SHADOW.add(True)
    if m.group(3) and m.group(3).split(",")[0].strip() == "true" and m.group(1) in ("k_intersect", "k_shadow_any", "k_shadow_walk", "k_shade"):
        return None  # the instrumented (counting) instantiation is not the timed one
    if m.group(3):
        INSTANCES[m.group(1)][tuple(a.strip() for a in m.group(3).split(","))] += 1
    return m.group(1)


def mangled_pattern(kernel):
    """Regex for the assembly symbol of the instantiation of `kernel` this workload dispatched most (bool arguments only)."""
    if not INSTANCES.get(kernel):
        return None
    args = max(INSTANCES[kernel].items(), key=lambda kv: kv[1])[0]
    if not all(a in ("true", "false") for a in args):
        return None
    return kernel + "I" + "".join("Lb1E" if a == "true" else "Lb0E" for a in args) + "E"


def read_pass(d):
    out, calls = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(set))
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:  # one run per pass: a directory that was used before also holds the older runs' files
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = kernel_of(row["Kernel_Name"])
                if k:
                    out[k][row["Counter_Name"]] += float(row["Counter_Value"])
                    calls[k][row["Counter_Name"]].add(row["Dispatch_Id"])
    return out, calls


def bench_json(path):
    try:
        return json.loads(open(path).read().strip().splitlines()[-1])
    except (OSError, ValueError, IndexError):
        return None


def main():
    d, workload, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    dest = sys.argv[4] if len(sys.argv) > 4 else None
    passes = {p: read_pass(os.path.join(d, p)) for p in ("fetch", "write", "sq1", "sq2", "m1", "m3")}
    ms = {p: (bench_json(os.path.join(d, p + ".json")) or {}).get("kernel_ms_per_step", {}) for p in passes}
    per_step = {p: (bench_json(os.path.join(d, p + ".json")) or {}).get("kernel_launches_per_step", {}) for p in passes}
    kernels = {}
    for k in sorted(set().union(*[set(v[0]) for v in passes.values()])):
        e = {"measured": tag}
        c = {}
        for p, (vals, calls) in passes.items():
            for name, v in vals.get(k, {}).items():
                c[name] = (v, len(calls[k][name]), p)
        val = lambda n: c[n][0] if n in c else None
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            n = c["FETCH_SIZE"][1]
            e["launches"] = n
            e["fetch_size_kib_per_launch"] = c["FETCH_SIZE"][0] / n
            e["write_size_kib_per_launch"] = c["WRITE_SIZE"][0] / c["WRITE_SIZE"][1]
            e["hbm_bytes_per_launch"] = (2.0 * e["fetch_size_kib_per_launch"] + e["write_size_kib_per_launch"]) * 1024.0
        if "TCC_HIT_sum" in c and val("TCC_HIT_sum") + val("TCC_MISS_sum") > 0:
            e["l2_hit_rate"] = val("TCC_HIT_sum") / (val("TCC_HIT_sum") + val("TCC_MISS_sum"))
        # a pass runs more steps than the timed one (bench.py adds an untimed plain step): the counters sum over all of
        # them, the event time is per step, so the kernel's cycles are scaled by dispatches / launches per step
        def cyc(p, counter):
            key = KEY.get(k, "")
            lps = per_step[p].get(key, 0.0)
            steps = c[counter][1] / lps if lps else 1.0
            return ms[p].get(key, 0.0) * 1e-3 * CLOCK * steps
        if "SQ_INSTS_VALU" in c:
            e["valu_instructions_per_launch"] = val("SQ_INSTS_VALU") / c["SQ_INSTS_VALU"][1]
        if "SQ_ACTIVE_INST_VALU" in c and cyc("sq1", "SQ_ACTIVE_INST_VALU") > 0:
            e["valu_busy"] = val("SQ_ACTIVE_INST_VALU") * 4 / (cyc("sq1", "SQ_ACTIVE_INST_VALU") * SIMDS)
            e["lane_utilisation"] = val("SQ_THREAD_CYCLES_VALU") / (64 * val("SQ_ACTIVE_INST_VALU"))
            e["waves_per_simd"] = val("SQ_WAVE_CYCLES") * 4 / (cyc("sq1", "SQ_WAVE_CYCLES") * SIMDS)
            e["waves_per_simd_compiled"] = 6 if (k == "k_shadow_any" and LEAN_SHADOW) else 5  # csrc/Makefile: MIPT_TRAV_WAVES, MIPT_LEAN_WAVES
            e["kernel_ms_in_counter_pass"] = ms["sq1"].get(KEY.get(k, ""))
        if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
            e["wave_time_waiting_on_memory"] = val("SQ_WAIT_ANY") / val("SQ_WAVE_CYCLES")
        if "SQ_INSTS_VMEM_RD" in c:
            e["vmem_read_instructions_per_launch"] = val("SQ_INSTS_VMEM_RD") / c["SQ_INSTS_VMEM_RD"][1]
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
            e["tcp_accesses_per_launch"] = val("TCP_TOTAL_CACHE_ACCESSES_sum") / c["TCP_TOTAL_CACHE_ACCESSES_sum"][1]
        if "SQ_INSTS_SALU" in c:
            nl = c["SQ_INSTS_SALU"][1]
            e["salu_instructions_per_launch"] = val("SQ_INSTS_SALU") / nl
            e["fp64_valu_instructions_per_launch"] = sum(val(n) or 0.0 for n in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64")) / nl
        if "TA_TA_BUSY_sum" in c and cyc("m1", "TA_TA_BUSY_sum") > 0:
            e["ta_busy"] = val("TA_TA_BUSY_sum") / (CUS * cyc("m1", "TA_TA_BUSY_sum"))
            e["tcp_busy"] = val("TCP_GATE_EN1_sum") / (CUS * cyc("m1", "TCP_GATE_EN1_sum"))
        if "TCP_TCC_READ_REQ_sum" in c and val("TCP_TOTAL_CACHE_ACCESSES_sum"):
            e["l1_miss_rate"] = val("TCP_TCC_READ_REQ_sum") / val("TCP_TOTAL_CACHE_ACCESSES_sum")
        lps = per_step["sq1"].get(KEY.get(k, ""))
        if lps:
            e["launches_per_step"] = lps  # per-launch figures are averages over the launches of such a step
        kernels[k] = e
    # issue classes of the traversal kernels' instructions (tools/valu_mix.py: static census of the instantiation this
    # workload ran, its persistent loop): the ceiling of their vector-instruction rate depends on that mix
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import valu_mix

        asm = valu_mix.compile_asm()
        for k in ("k_shadow_any", "k_intersect"):
            pat = mangled_pattern(k)
            if k not in kernels or not pat:
                continue
            for name, lines in valu_mix.kernel_bodies(asm):
                if pat in name:
                    outer, inner = valu_mix.hot_loop(lines)
                    whole = valu_mix.census(lines[outer[0]:outer[1] + 1]) if outer else valu_mix.census(lines)
                    kernels[k]["valu_mix"] = {"instantiation": pat, "valu_in_persistent_loop": whole["valu"], "slow_class_share": whole["slow_share"],
                                              "cycles_per_instruction": whole["cycles_per_instruction"]}
                    if inner:
                        dl = valu_mix.census(lines[inner[0]:inner[1] + 1])
                        kernels[k]["valu_mix"]["descend_loop"] = {"valu": dl["valu"], "slow_class_share": dl["slow_share"], "cycles_per_instruction": dl["cycles_per_instruction"]}
    except Exception as ex:  # no compiler on this machine: the census is optional
        print("valu_mix skipped:", ex, file=sys.stderr)
    step = bench_json(os.path.join(d, "sq1.json")) or {}
    out = {}
    if dest and os.path.exists(dest):
        out = json.load(open(dest))
    out[workload] = {
        "kernel_source_hash": kernel_source_hash(),
        "measured": tag,
        # the step the passes ran: per-launch averages hold for steps of this size only
        "step_samples": (step.get("counters") or {}).get("samples"),
        "clock_hz_assumed": CLOCK,
        "note": "one bench step per pass under rocprofv3 --pmc (separate passes; kernels run a few % slower than untraced)",
        "kernels": kernels,
    }
    json.dump(out, open(dest, "w") if dest else sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
