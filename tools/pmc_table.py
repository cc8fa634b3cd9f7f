#!/usr/bin/env python3
"""Diagnostic: per-kernel mean of every counter found in rocprofv3 --pmc CSV outputs.
Usage: python tools/pmc_table.py gpurun_out/apmc_1 gpurun_out/apmc_2 ... [substring]"""
import csv, glob, os, sys
from collections import defaultdict
dirs = [a for a in sys.argv[1:] if os.path.isdir(a)]
sub = [a for a in sys.argv[1:] if not os.path.isdir(a)]
acc = defaultdict(lambda: defaultdict(list))
for d in dirs:
    fns = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for fn in fns[-1:]:   # newest run only
        for row in csv.DictReader(open(fn)):
            full = row["Kernel_Name"]
            if sub and not any(s in full for s in sub):
                continue
            k = full.replace("void ", "").replace("(anonymous namespace)::", "")[:48]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
