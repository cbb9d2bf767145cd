#!/usr/bin/env python3
"""Build profiles/{compare,sketch}_pmc_latest.json (read by bench.py for roofline.traffic / roofline.issue)
from rocprofv3 --pmc passes over ONE step of bench.py (tools/profile_round.sh): per kernel
(substring match) the counters are SUMMED over the launches of that step -- a compare pass is one
launch per value window -- and divided by the units of the step (pairs, k-mers).
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024:
  * both counters are in KiB (calibrated in round 1 on torch kernels of known size);
  * gfx950 FETCH_SIZE counts 128-B requests as 64 B (MI355X_MICROARCH.md, HBM section) -> x2;
    confirmed on sketch_chunks_kernel: 1.0e10 bases read, FETCH_SIZE * 1024 * 2 = 1.0e10.
The JSON is stamped with a hash of the kernel's source files; bench.py drops it when they change.
usage: tools/make_pmc_json.py <prefix of the pass dirs> <kernel-substring> <units> <unit name> <out.json> <src files...>"""
import csv, glob, hashlib, json, os, sys
prefix, kern, units, uname, out = sys.argv[1:6]
srcs = sys.argv[6:]
units = float(units)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def collect(tag):
    agg, n = {}, {}
    for p in glob.glob(os.path.join(prefix + tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            if kern in r["Kernel_Name"]:
                c = r["Counter_Name"]
                agg[c] = agg.get(c, 0.0) + float(r["Counter_Value"])
                n[c] = n.get(c, 0) + 1
    return agg, n

c, n = {}, {}
for tag in ("fetch", "write", "sqa", "sqb", "sqc"):
    a, k = collect(tag)
    c.update(a); n.update(k)
assert "FETCH_SIZE" in c and "WRITE_SIZE" in c, "kernel not found in the PMC output"
h = hashlib.sha256()
for s in srcs:
    h.update(open(os.path.join(ROOT, s), "rb").read())
read_b, write_b = 2 * c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
res = {"kernel": kern, "launches_per_pass": n["FETCH_SIZE"], "kernel_src_sha": h.hexdigest()[:16], "kernel_src": srcs,
       "hbm_read_bytes_per_pass": read_b, "hbm_write_bytes_per_pass": write_b, "hbm_bytes_per_pass": read_b + write_b,
       "hbm_bytes_per_launch": (read_b + write_b) / n["FETCH_SIZE"],
       "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B), units KiB",
       "units_per_pass": units, "unit": uname, "source": os.path.basename(prefix.rstrip("_")) + "_pmc_digest.txt",
       "counters_summed_over_the_pass": {k: v for k, v in sorted(c.items())}}
for key, ctr in (("valu", "SQ_INSTS_VALU"), ("salu", "SQ_INSTS_SALU"), ("lds", "SQ_INSTS_LDS"), ("vmem", "SQ_INSTS_VMEM_RD")):
    if ctr in c:
        res[f"{key}_per_{uname}"] = round(c[ctr] / units, 3)
if "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"]:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs: /8 = the kernels' cycles; SQ_* counters are summed over
    # the 256 CUs (1024 SIMDs); a wave64 VALU instruction occupies its SIMD for 4 cycles
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    res["cycles_per_pass"] = cyc
    if "SQ_LDS_IDX_ACTIVE" in c:
        res["lds_active_frac"] = round(c["SQ_LDS_IDX_ACTIVE"] / 256.0 / cyc, 3)
    if "SQ_INSTS_VALU" in c:
        res["valu_issue_frac"] = round(c["SQ_INSTS_VALU"] * 4.0 / 1024.0 / cyc, 3)
    if "SQ_INSTS_SALU" in c:
        res["salu_issue_frac"] = round(c["SQ_INSTS_SALU"] * 4.0 / 1024.0 / cyc, 3)
if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
    res["lds_bank_conflict_share"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 3)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
