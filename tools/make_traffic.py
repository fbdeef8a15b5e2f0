#!/usr/bin/env python3
"""Turn a pmc summary (tools/pmc_summary.py) into profiles/traffic.json.

HBM-side bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / launches: both counters
are in KiB, and on gfx950 FETCH_SIZE tallies 128-byte read requests as 64 bytes, so the
read side is doubled.  That factor is calibrated on wide streaming reads only; for the
gather pattern of the traversal kernels it is an upper bound of the read side.
usage: make_traffic.py <pmc_summary.json> <workload> [existing traffic.json]
"""
import json, sys

pmc = json.load(open(sys.argv[1]))
workload = sys.argv[2]
out = {}
if len(sys.argv) > 3:
    try:
        out = json.load(open(sys.argv[3]))
    except OSError:
        out = {}
entry = {}
for name, c in pmc.items():
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c or not name.startswith("k_"):
        continue
    base = name.split("<")[0]
    if name.endswith("<true>"):  # the instrumented (counting) instantiation is not the timed one
        continue
    n = c["FETCH_SIZE"]["dispatches"]
    fetch_kib = c["FETCH_SIZE"]["sum"] / n
    write_kib = c["WRITE_SIZE"]["sum"] / c["WRITE_SIZE"]["dispatches"]
    e = {
        "launches": n,
        "fetch_size_kib_per_launch": fetch_kib,
        "write_size_kib_per_launch": write_kib,
        "hbm_bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0,
    }
    if "TCC_HIT_sum" in c:
        h, m = c["TCC_HIT_sum"]["sum"], c["TCC_MISS_sum"]["sum"]
        e["l2_hit_rate"] = h / (h + m) if h + m else None
    entry[base] = e
out[workload] = entry
json.dump(out, sys.stdout, indent=1, sort_keys=True)
