#!/usr/bin/env python3
"""Reads a rocprofv3 kernel trace of tools/one_call.py and prints, for the LAST call in it, every k_shadow_any and k_intersect
launch in order with its duration.  usage: python tools/trace_levels.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last call starts at the last k_intersect launch that follows a k_resolve
start = max(i for i, n in enumerate(names) if "k_resolve" in n and i < len(names) - 1) if sum("k_resolve" in n for n in names) > 1 else -1
res = [i for i, n in enumerate(names) if "k_resolve" in n]
lo = res[-2] + 1 if len(res) >= 2 else 0
for kind in ("k_shadow_any", "k_intersect", "k_shade", "k_tail", "k_shadow_sum"):
    ds = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows[lo:] if kind in r["Kernel_Name"]]
    print(kind, len(ds), "launches, total %.2f ms:" % sum(ds), " ".join("%.2f" % x for x in ds))
t0 = int(rows[lo]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows[lo:])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[lo:])
print("span of the call's kernels %.2f ms, sum of their durations %.2f ms" % ((t1 - t0) / 1e6, busy / 1e6))
