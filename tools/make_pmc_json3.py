#!/usr/bin/env python3
"""profiles/<leg>_pmc.json from the rocprofv3 passes of ONE leg (tools/profile_round3.sh over tools/prof_leg.py):
per kernel of the leg, summed over its launches and divided by the passes of the run (1 warm-up + steps):
HBM bytes, launch time (from the kernel-trace stats of the same leg), cycles, and the share of each issue port.

  HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (KiB; gfx950 FETCH_SIZE counts 128-B requests as 64 B,
              MI355X_MICROARCH.md "HBM"; both calibrated in round 1 on copies of known size)
  cycles    = GRBM_GUI_ACTIVE / 8 (summed over the 8 XCDs) ; effective clock = cycles / launch time
  ports     : valu = SQ_INSTS_VALU * 4 / (1024 SIMDs * cycles)   (a wave64 VALU op holds its SIMD 4 cycles)
              salu = SQ_INSTS_SALU / (1024 * cycles)             (instructions per SIMD-cycle; one scalar unit per SIMD)
              vmem = SQ_INSTS_VMEM_RD / (1024 * cycles), lds_inst = SQ_INSTS_LDS / (1024 * cycles)
              lds_active = SQ_LDS_IDX_ACTIVE / (256 CUs * cycles), lds_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  Ports issue in parallel: the fractions are reported one by one and never summed.

Kernels of the per-step pass are divided by the passes of the run; kernels that only run cold (index build,
counting pass: the second substring list) are reported per run.

usage: make_pmc_json3.py <dir prefix of the leg's passes> <passes> <units per pass> <unit> <out.json> <pass kernel substrings, comma separated> <cold kernel substrings> <src files...>"""
import csv, glob, hashlib, json, os, sys
prefix, passes, units, uname, out, subs, cold_subs = sys.argv[1:8]
srcs = sys.argv[8:]
passes, units = float(passes), float(units)
cold_subs = [c for c in cold_subs.split(",") if c]
subs = subs.split(",") + cold_subs


def is_cold(name):
    return any(c in name for c in cold_subs)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    for sub in subs:
        if sub in name:
            return name.split("(")[0].replace("void ", "").strip()[:90]
    return None


ctr = {}
for tag in ("fetch", "write", "sqa", "sqb"):
    for p in glob.glob(os.path.join(prefix + tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            if k is None:
                continue
            if tag == "sqb" and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                continue                                   # every SQ pass carries its own cycles: sqb's are used below, with its LDS counters
            d = ctr.setdefault(k, {})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            if r["Counter_Name"] in ("FETCH_SIZE", "GRBM_GUI_ACTIVE"):
                d["_launches_" + r["Counter_Name"]] = d.get("_launches_" + r["Counter_Name"], 0) + 1
times = {}
for p in glob.glob(os.path.join(prefix + "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        k = short(r["Name"])
        if k is not None:
            # (several kernels can share a short name -- rocPRIM's templates differ only behind the first 90 characters: their
            #  launches are summed, not overwritten; round 4's file booked the sort at one launch's average)
            t = times.setdefault(k, {"calls": 0, "total_ms": 0.0, "avg_ms": 0.0})
            t["calls"] += int(r["Calls"])
            t["total_ms"] += float(r["TotalDurationNs"]) / 1e6
            t["avg_ms"] = t["total_ms"] / max(t["calls"], 1)
h = hashlib.sha256()
for s in srcs:
    h.update(open(os.path.join(ROOT, s), "rb").read())
res = {"kernel_src_sha": h.hexdigest()[:16], "kernel_src": srcs, "passes_in_the_run": passes, "units_per_pass": units, "unit": uname,
       "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B), both sizes in KiB", "kernels": {}}
tot_r = tot_w = tot_ms = 0.0
for k, c in sorted(ctr.items()):
    e = {}
    cold = is_cold(k)
    passes_k = 1.0 if cold else passes
    if cold:
        e["cold_only"] = True
    if "FETCH_SIZE" in c:
        e["hbm_read_bytes_per_pass"] = 2 * c["FETCH_SIZE"] * 1024 / passes_k
        tot_r += 0 if cold else e["hbm_read_bytes_per_pass"]
        e["launches_per_pass"] = c["_launches_FETCH_SIZE"] / passes_k
    if "WRITE_SIZE" in c:
        e["hbm_write_bytes_per_pass"] = c["WRITE_SIZE"] * 1024 / passes_k
        tot_w += 0 if cold else e["hbm_write_bytes_per_pass"]
    if k in times:
        e["ms_per_pass"] = times[k]["total_ms"] / passes_k
        e["avg_launch_ms"] = times[k]["avg_ms"]
        tot_ms += 0 if cold else e["ms_per_pass"]
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc:
        e["cycles_per_pass"] = cyc / passes_k
        if k in times and times[k]["total_ms"] > 0:
            e["effective_clock_ghz"] = round(cyc / (times[k]["total_ms"] * 1e-3) / 1e9, 3)
        ports = {}
        if "SQ_INSTS_VALU" in c: ports["valu"] = round(c["SQ_INSTS_VALU"] * 4.0 / 1024.0 / cyc, 4)
        if "SQ_INSTS_SALU" in c: ports["salu"] = round(c["SQ_INSTS_SALU"] / 1024.0 / cyc, 4)
        if "SQ_INSTS_VMEM_RD" in c: ports["vmem"] = round(c["SQ_INSTS_VMEM_RD"] / 1024.0 / cyc, 4)
        if "SQ_INSTS_LDS" in c: ports["lds_inst"] = round(c["SQ_INSTS_LDS"] / 1024.0 / cyc, 4)
        e["ports"] = ports
    e["counters_per_pass"] = {n: v / passes_k for n, v in sorted(c.items()) if not n.startswith("_")}
    res["kernels"][k] = e
# LDS activity needs the cycles of the SAME pass as the LDS counters (sqb carries its own GRBM_GUI_ACTIVE)
for p in glob.glob(os.path.join(prefix + "sqb", "**", "*counter_collection.csv"), recursive=True):
    per = {}
    for r in csv.DictReader(open(p)):
        k = short(r["Kernel_Name"])
        if k is not None:
            d = per.setdefault(k, {})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for k, d in per.items():
        cyc = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if cyc and "SQ_LDS_IDX_ACTIVE" in d:
            res["kernels"][k].setdefault("ports", {})["lds_active"] = round(d["SQ_LDS_IDX_ACTIVE"] / 256.0 / cyc, 4)
            if d["SQ_LDS_IDX_ACTIVE"]:
                res["kernels"][k]["ports"]["lds_conflict_share"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"], 4)
        if cyc and "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"]:
            res["kernels"][k]["wait_share_of_wave_cycles"] = round(d.get("SQ_WAIT_INST_ANY", 0.0) / d["SQ_WAVE_CYCLES"], 4)
res["hbm_read_bytes_per_pass"] = tot_r
res["hbm_write_bytes_per_pass"] = tot_w
res["hbm_bytes_per_pass"] = tot_r + tot_w
res["kernel_ms_per_pass"] = tot_ms
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: {x: v[x] for x in v if x != "counters_per_pass"} for k, v in res["kernels"].items()}))
print("total", res["hbm_bytes_per_pass"], "B per pass,", tot_ms, "ms of kernels per pass")
