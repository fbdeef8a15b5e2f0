#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter CSVs per kernel: usage pmc_summary.py <dir> [<dir> ...] > summary.json"""
import csv, glob, json, os, re, sys
from collections import defaultdict

def short(name):
    m = re.search(r"(k_[a-z_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]

out = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(set))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                out[k][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[k][row["Counter_Name"]].add(row["Dispatch_Id"])
res = {k: {c: {"sum": v, "dispatches": len(calls[k][c])} for c, v in cs.items()} for k, cs in out.items()}
json.dump(res, sys.stdout, indent=1, sort_keys=True)
