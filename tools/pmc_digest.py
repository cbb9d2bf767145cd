#!/usr/bin/env python3
"""Digest rocprofv3 --pmc counter_collection.csv files: per kernel (substring match),
sum of each counter over dispatches and per-dispatch average.
usage: tools/pmc_digest.py <dir-or-csv>... [--kernels compare_tiled,sketch_chunks]"""
import collections, csv, glob, os, sys
paths, kernels = [], ["compare_tiled", "compare_generic", "sketch_chunks", "merge_chunks"]
for a in sys.argv[1:]:
    if a.startswith("--kernels="):
        kernels = a.split("=", 1)[1].split(",")
    elif os.path.isdir(a):
        paths += glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True)
    else:
        paths.append(a)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in paths:
    for r in csv.DictReader(open(p)):
        for k in kernels:
            if k in r["Kernel_Name"]:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v):3d} avg/dispatch={sum(v)/len(v):.6g}")
