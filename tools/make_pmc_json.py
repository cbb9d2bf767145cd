#!/usr/bin/env python3
"""Build profiles/compare_pmc_latest.json (read by bench.py for roofline.traffic) from two
rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over `bench.py --no-sketch --no-cpu`.
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024:
  * both counters are in KiB (calibrated here on torch kernels of known size:
    index_elementwise reading 2.048e9 B reports FETCH_SIZE 1.00e6; its 2.56e8 B of stores
    report WRITE_SIZE 2.5e5);
  * gfx950 FETCH_SIZE counts 128-B requests as 64 B (MI355X_MICROARCH.md §HBM) -> x2;
    confirmed on sketch_chunks_kernel: 2.0e9 bases read, FETCH_SIZE*1024 = 1.0e9.
usage: tools/make_pmc_json.py <fetch_dir> <write_dir> <kernel-substring> <out.json> [note]"""
import csv, glob, json, os, sys
fetch_dir, write_dir, kern, out = sys.argv[1:5]
note = sys.argv[5] if len(sys.argv) > 5 else ""

def collect(d, counter):
    vals = []
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            if kern in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return vals

f = collect(fetch_dir, "FETCH_SIZE")
w = collect(write_dir, "WRITE_SIZE")
assert f and w, "kernel not found in PMC output"
fetch = sum(f) / len(f)
write = sum(w) / len(w)
res = {"kernel": kern, "launches_fetch": len(f), "launches_write": len(w),
       "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
       "hbm_read_bytes_per_launch": 2 * fetch * 1024, "hbm_write_bytes_per_launch": write * 1024,
       "hbm_bytes_per_launch": (2 * fetch + write) * 1024,
       "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B), units KiB", "note": note}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
