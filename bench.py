#!/usr/bin/env python3
"""bench.py — headline benchmark of the Mash hot path on MI355X.

Metric (BASELINE.json): pairwise Mash distances/sec at s=1000, k=21 (64-bit hashes).
Workload: BASELINE config 3 — `mash triangle`, all-vs-all on N=100 000 pre-built clustered synthetic
sketches: 4.99995e9 pairs per step.  The table (800 MB) is resident in HBM on every rank
before the timed region.  A STEP IS THE PER-TABLE JOB, as `mash triangle` pays it once per table
(CommandTriangle.cpp:129-139): every step starts from a table the library has not seen
(mg_table_invalidate) -- inverted index built, candidates discovered, 8 B written for every pair, candidates
merged -- row-block sharded across the ranks (mg_shard_tri_rows), each rank writing {numer, denom} for its
rows into its own HBM buffer.  `warm_value` is the rate of further passes over a table whose index exists.
Total work is fixed as N grows ("scaling": "strong").  The one exchange is the broadcast of the table from
rank 0 before the timed region (mg_comm rank mode: ncclBroadcast inside libmashgpu; `config.rccl_ranks`).

Output: the LAST stdout line is the headline, one compact JSON object (< 3 KB: metric, value, roofline,
cpu_baseline, one scalar per secondary leg); a few short per-leg lines precede it; every detail (phases,
ports, notes, samples) goes to bench_detail.json next to this file (and to gpurun_out/ when that exists).
What the fields mean is written once, in DESIGN.md section 5.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--n-sketches 100000] [--n-genomes 10000]
"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
CLOCK_HZ = 2.4e9               # engine clock under load (MI355X_MICROARCH.md)
SIMDS = 256 * 4
S = 1000
K = 21
from workloads.checksums import C3_CHECKSUM    # established by tests/test_gpu_parity.py::test_c3_full_size_triangle


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-sketches", type=int, default=100_000)
    ap.add_argument("--n-genomes", type=int, default=10_000)
    ap.add_argument("--genome-len", type=int, default=1_000_000)
    ap.add_argument("--no-sketch", action="store_true", help="skip the secondary sketch measurement")
    ap.add_argument("--no-screen", action="store_true", help="skip the tertiary screen measurement (config 4)")
    ap.add_argument("--no-c5", action="store_true", help="skip the large-sketch triangle (config 5)")
    ap.add_argument("--no-h2h", action="store_true", help="skip the host-to-host legs")
    ap.add_argument("--no-cli", action="store_true", help="skip the CLI end-to-end leg (mash sketch, ours vs the reference CLI)")
    ap.add_argument("--no-brackets", action="store_true", help="skip SURVEY 8d's extremes (all-random / all-identical / clades)")
    ap.add_argument("--n-reads", type=int, default=10_000_000)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline legs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--detail", default=None, help="where the full result goes (default: bench_detail.json next to this file, and gpurun_out/)")
    ap.add_argument("--dry-cpu", action="store_true",
                    help="plumbing test only (CI without GPUs): gloo + CPU tensors, no kernels, output marked dry")
    return ap.parse_args()


def src_sha(*rel):
    h = hashlib.sha256()
    for r in rel:
        h.update(open(os.path.join(ROOT, r), "rb").read())
    return h.hexdigest()[:16]


def load_pmc(name, *src):
    """profiles/<name>: PMC figures of one pass, valid only for the kernel source they were taken on."""
    p = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(p):
        return None
    try:
        d = json.load(open(p))
    except Exception:
        return None
    if d.get("kernel_src_sha") != src_sha(*src):
        return None                                    # stale: the kernel changed since the counters were read
    d.setdefault("source", "profiles/" + name)
    return d


def compare_roofline(eng, pairs, n, s, steps, pmc):
    """Roofline object of one triangle pass from the library's HIP-event records (mg_prof_*): which engine ran, its
    phases' times (index = the per-table build, present in cold steps only), and the dominant kernel against the
    bound that applies.  Meaning of every field: DESIGN.md section 5."""
    phases = {}
    for name in ("compare", "compare_index", "compare_discover", "compare_fill", "compare_fill_aside", "compare_dense", "compare_merge", "compare_join"):
        ms, k = eng.prof_avg_ms(name)
        if k:
            phases[name.replace("compare_", "")] = {"avg_ms": round(ms, 4), "per_pass": k / steps, "ms_per_pass": round(ms * k / steps, 4)}
    sparse = "fill" in phases
    # (fill_aside: the constant fill of a per-table step, written on a stream of its own WHILE the index is built -- its time
    #  lies inside the index phase's, not behind it; what is left of it when the build has ended is the `fill` phase)
    pass_ms = sum(v["ms_per_pass"] for k, v in phases.items() if k != "fill_aside")
    compulsory = pairs * 8 + n * s * 8 + n * 12           # every pair written once, the table read once
    traffic = pmc.get("hbm_bytes_per_pass") if pmc else None
    whole = {"ms": round(pass_ms, 3), "traffic": traffic, "compulsory_bytes": compulsory,
             "traffic_over_compulsory": round(traffic / compulsory, 3) if traffic else None,
             "output_write_bound_frac": round(pairs * 8 / (pass_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if pass_ms > 0 else None}
    model = pairs * (2 * s * 8 + 8) / (pass_ms * 1e-3) / 1e9 if pass_ms > 0 else 0.0
    if "join" in phases:
        # the join engine (compare_join.hip): one kernel writes every pair; what bounds it is instruction issue and the LDS
        # (a counter update per pair and shared hash), not HBM -- its 8 B per pair against the HBM peak is reported because the
        # contract asks for one roof, the update rate beside it
        j = phases["join"]["ms_per_pass"]
        wr = pairs * 8 / (j * 1e-3) / 1e9
        r = {"bound": "hbm", "achieved": round(wr, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(wr / HBM_PEAK_GBS, 4), "traffic": traffic,
             "engine": "join", "kernel": "mg::jn_tile_kernel", "kernel_ms": phases["join"]["avg_ms"], "algorithmic_bytes_per_launch": pairs * 8,
             "phases": phases, "pass": whole, "survey_8d_no_reuse_model_gbs": round(model, 1),
             "note": "issue- and LDS-bound (one counter update per pair and shared hash); HBM sees the lists and 8 B per pair"}
    elif not sparse:
        # the tile engine: one kernel per window, priced with SURVEY 8d's no-reuse model (bounds nothing: DESIGN 4.1b)
        r = {"bound": "hbm", "achieved": round(model, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(model / HBM_PEAK_GBS, 4),
             "traffic": traffic, "engine": "tiles", "kernel": "mg::compare_merged_kernel",
             "kernel_ms": phases.get("compare", {}).get("avg_ms"), "phases": phases, "pass": whole}
    else:
        # the inverted-index engine: the dominant kernel is the fill -- 8 B written per pair (SURVEY 8d's compulsory traffic)
        aside = phases.get("fill_aside")
        f = phases["fill"]["ms_per_pass"] + (aside["ms_per_pass"] if aside else 0.0)
        fill = pairs * 8 / (f * 1e-3) / 1e9
        kname = "mg::sp_fill_chunks_kernel" if aside else "mg::sp_fill_value_kernel"
        kp = (pmc or {}).get("kernels", {}).get(kname)
        r = {"bound": "hbm", "achieved": round(fill, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(fill / HBM_PEAK_GBS, 4),
             "traffic": (kp["hbm_read_bytes_per_pass"] + kp["hbm_write_bytes_per_pass"]) if kp else None,
             "engine": "inverted index", "kernel": kname, "kernel_ms": round(f, 4) if aside else phases["fill"]["avg_ms"],
             "algorithmic_bytes_per_launch": pairs * 8, "phases": phases, "pass": whole,
             "survey_8d_no_reuse_model_gbs": round(model, 1)}
        if aside:
            # two launches share the 8 B per pair: one beside the index build (few workgroups: the build's loads must not
            # queue behind the writes), one at full speed for what is left -- their durations added; the kernel alone, a
            # further pass over the indexed table (roofline_warm), writes at 0.75 of the peak
            r["beside"] = "the index build, on a stream of its own (%.3f ms), then what is left (%.3f ms)" % (aside["ms_per_pass"], phases["fill"]["ms_per_pass"])
            # (rocprofv3 --stats averages over the launches: two per pass)
            r["kernel_launches_per_pass"] = 2
            r["kernel_avg_launch_ms"] = round(f / 2.0, 4)
    if pmc:
        r["ports"] = {k: dict(v.get("ports", {}), ms_per_pass=v.get("ms_per_pass"), effective_clock_ghz=v.get("effective_clock_ghz"),
                              hbm_bytes_per_pass=(v.get("hbm_read_bytes_per_pass", 0) + v.get("hbm_write_bytes_per_pass", 0)))
                      for k, v in pmc.get("kernels", {}).items()}
        r["pmc_source"] = pmc.get("source")
    return r


def compact_roofline(r):
    """The few roofline fields of the headline line (everything else stays in bench_detail.json)."""
    if not r or "phases" not in r:
        return r
    return {"bound": r["bound"], "kernel": r["kernel"], "kernel_ms": r["kernel_ms"], "achieved": r["achieved"], "peak": r["peak"],
            "unit": r["unit"], "frac": r["frac"], "traffic": r["traffic"],
            # the WHOLE step against the same roof: compulsory bytes (every pair written once + the table read
            # once) / the timed step / the HBM peak; and what the per-table index costs of it
            "step_frac": r.get("step_frac"), "index_ms": r.get("index_ms"), **({"beside": r["beside"], "kernel_launches_per_pass": r["kernel_launches_per_pass"], "kernel_avg_launch_ms": r["kernel_avg_launch_ms"]} if r.get("beside") else {}),
            **({"alone": r["alone"]} if r.get("alone") else {}),
            "pass": {"ms": r["pass"]["ms"], "phases_ms": {k: round(v["ms_per_pass"], 3) for k, v in r["phases"].items()},
                     "traffic_over_compulsory": r["pass"]["traffic_over_compulsory"],
                     "output_write_bound_frac": r["pass"]["output_write_bound_frac"]}}


def usable_cpus():
    """The host cores THIS process may run on: os.cpu_count() names the machine's (256 on the MI355X boxes), the scheduler
    affinity and the cgroup's CPU quota say what the container gets (round 6: 16 threads ran 15 x one thread, 64 and 256 no
    faster -- the quota, not the reference's code)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: (t.split()[0], t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            txt = open(path).read().strip()
            if parse:
                q, per = parse(txt)
                if q != "max":
                    n = min(n, max(1, int(int(q) / int(per) + 0.5)))
            else:
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip())
                if int(txt) > 0:
                    n = min(n, max(1, int(int(txt) / per + 0.5)))
        except (OSError, ValueError, IndexError):
            pass
    return n


def cpu_baseline_compare(table_np, nhash_np, lengths_np, budget_s, cores=None, cli=False):
    """Reference compareSketches (incl. distance and p-value) on the host cores, bounded sample: triangle rows of the first M
    sketches of the SAME table, M sized for ~budget_s.  The reference's table (vector<Sketch::Reference>) is built ONCE and
    shared read-only by every thread (round 5 built it per worker: at 256 threads the copies were the measurement); the rows go
    to the threads as many more blocks of equal area than there are threads, taken as they come (ThreadPool.hxx hands out jobs
    the same way).  cli: `mash-ref triangle -p <cores>` on the same sample as a second figure (the reference's whole command:
    .msh read, compare, Phylip text)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyoracle
    cores = cores or min(os.cpu_count() or 1, 16)
    use_ref = pyoracle.ref_available()
    orc = pyoracle.Oracle(ref=use_ref)
    kspace = 4.0 ** K
    from mash_amd import shard

    def run(m):
        t = orc.table_open(table_np[:m], nhash_np[:m], lengths_np[:m])          # (outside the timed region, as the reference reads its .msh first)
        try:
            b = shard.equal_area_row_blocks(m, min(max(cores * 16, 64), m))
            t0 = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:      # ctypes releases the GIL: real parallelism
                list(ex.map(lambda g: orc.triangle_run(t, b[g], b[g + 1], K, kspace), range(len(b) - 1)))
            return time.perf_counter() - t0
        finally:
            orc.table_close(t)

    # calibrate on growing samples until one takes a measurable time (a sample too small for the thread count measures thread start-up)
    m = min(len(table_np), max(400, int(400 * cores ** 0.5)))
    dt = run(m)
    while dt < 0.3 and m < len(table_np):
        m = min(len(table_np), m * 2)
        dt = run(m)
    rate = (m * (m - 1) / 2) / dt
    m2 = int(min(len(table_np), max(m, (2 * rate * budget_s) ** 0.5)))
    dt2 = run(m2) if m2 > m else dt
    pairs = m2 * (m2 - 1) // 2
    r = {"value": pairs / dt2, "unit": "pairs/s", "cores": cores,
         "kind": "reference" if use_ref else "port",
         "sample": f"triangle (compareSketches incl. p-value) on the first {m2} of the same sketches, "
                   f"{pairs} pairs, {cores} threads on one shared table, {dt2:.1f} s"}
    if cli:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import compare_e2e
            if os.path.exists(compare_e2e.REF):
                d = tempfile.mkdtemp(prefix="bench_refcli_")
                try:
                    mc = int(min(len(table_np), max(1000, (2 * r["value"] * min(budget_s, 6.0)) ** 0.5)))
                    f = os.path.join(d, "sample.msh")
                    compare_e2e.write_msh(f, table_np[:mc], nhash_np[:mc], lengths_np[:mc])
                    dtc = compare_e2e.timed([compare_e2e.REF, "triangle", "-p", str(cores), f], "/dev/null")
                    r["mash_ref_cli"] = {"value": mc * (mc - 1) / 2 / dtc, "unit": "pairs/s", "threads": cores,
                                         "sample": f"mash-ref triangle -p {cores} on the first {mc} of the same sketches (.msh read + compare + Phylip text), {dtc:.1f} s"}
                finally:
                    shutil.rmtree(d, ignore_errors=True)
        except Exception as e:
            r["mash_ref_cli"] = {"error": repr(e)}
    return r


def cpu_baseline_sketch(budget_s, threads):
    """addMinHashes + MinHashHeap (the reference's objects) on `threads` host threads, 1 Mbp genomes."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyoracle
    from workloads import synth
    use_ref = pyoracle.ref_available()
    orc = pyoracle.Oracle(ref=use_ref)
    p = orc.params(k=K, s=S)
    genomes = [bytes(synth.synthetic_genome(g, 1_000_000)) for g in range(min(threads, 8))]
    t0 = time.perf_counter()
    n = 0

    def work(i):
        c = 0
        while time.perf_counter() - t0 < budget_s:
            orc.sketch_records([genomes[i % len(genomes)]], p)
            c += 1
        return c

    if threads == 1:
        n = work(0)
    else:
        with ThreadPoolExecutor(threads) as ex:
            n = sum(ex.map(work, range(threads)))
    dt = time.perf_counter() - t0
    return {"value": n * 1e6 / dt, "unit": "bp/s", "cores": threads, "kind": "reference" if use_ref else "port",
            "sample": f"{n} x 1 Mbp synthetic genome, addMinHashes+MinHashHeap, {threads} thread(s), {dt:.1f} s"}


def cpu_baseline_sketch_cli(n_genomes, threads):
    """The reference CLI itself (oracle/_ref/mash-ref: all of the reference's translation units) on
    FASTA files: kseq parsing + sketching + .msh writing, `mash sketch -p <threads>`."""
    from workloads import synth
    exe = os.path.join(ROOT, "oracle", "_ref", "mash-ref")
    if not os.path.exists(exe):
        return None
    d = tempfile.mkdtemp(prefix="bench_refcli_")
    try:
        files = []
        for g in range(n_genomes):
            seq = bytes(synth.synthetic_genome(g, 1_000_000))
            f = os.path.join(d, f"g{g}.fna")
            with open(f, "wb") as fh:
                fh.write(b">g%d\n" % g)
                for o in range(0, len(seq), 80):
                    fh.write(seq[o:o + 80] + b"\n")
            files.append(f)
        out = {}
        for p in sorted({1, threads}):
            nf = n_genomes if p > 1 else max(8, n_genomes // 8)
            t0 = time.perf_counter()
            r = subprocess.run([exe, "sketch", "-p", str(p), "-o", os.path.join(d, f"out{p}"), *files[:nf]],
                               capture_output=True, timeout=300)
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                return None
            out[p] = {"value": nf * 1e6 / dt, "unit": "bp/s", "cores": p, "kind": "reference",
                      "sample": f"mash-ref sketch -p {p} on {nf} FASTA files of 1 Mbp (kseq parse + sketch + .msh), {dt:.1f} s"}
        return out
    finally:
        subprocess.run(["rm", "-rf", d])


def _num(x, digits=4):
    """Short float for the headline line."""
    if x is None or isinstance(x, (str, bool, int)):
        return x
    return float(f"{x:.{digits}g}")


def headline_of(result):
    """The compact object of the LAST stdout line (< 3 KB; DESIGN.md section 5 says what each field is)."""
    cfg = result.get("config", {})
    h = {k: result.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    h["value"], h["ms_per_step"] = _num(h["value"], 6), _num(h["ms_per_step"], 5)
    h["config"] = {"workload": cfg.get("workload"), "parallelism": cfg.get("parallelism"), "output_checksum": cfg.get("output_checksum"),
                   "first_call_ms": cfg.get("first_call_ms"), "rccl_ranks": cfg.get("rccl_ranks"),
                   "table_broadcast_ms": cfg.get("table_broadcast_ms")}
    if (result.get("n_gpus") or 1) > 1:                     # several ranks: who talked to whom, who got which rows, what each index cost
        for k in ("comm", "rank_row_blocks", "row_weight_pairs", "prefix_weight_pairs", "index_ms_by_rank"):
            h["config"][k] = cfg.get(k)
    h["warm_value"], h["warm_ms_per_step"] = _num(result.get("warm_value"), 6), _num(result.get("warm_ms_per_step"), 5)
    if result.get("dry"):
        h["dry"] = True
        h["rank_blocks"] = result.get("rank_blocks")
    h["roofline"] = compact_roofline(result.get("roofline"))
    cb = result.get("cpu_baseline")
    if cb:
        h["cpu_baseline"] = {"value": _num(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                             "sample": cb["sample"][:160]}
        if cb.get("host"):
            h["cpu_baseline"]["host"] = cb["host"]
        if (cb.get("mash_ref_cli") or {}).get("value"):
            h["cpu_baseline"]["mash_ref_cli_pairs_s"] = _num(cb["mash_ref_cli"]["value"])
        if result.get("cpu_baseline_by_cores"):
            h["cpu_baseline"]["by_cores"] = {c: _num(v["value"]) for c, v in result["cpu_baseline_by_cores"].items()}
    sk, c5, scr, cli, br, hh = (result.get(k) or {} for k in ("sketch", "c5", "screen", "cli_e2e", "brackets", "host_to_host"))
    if "value" in sk:
        h["sketch_bp_s"] = _num(sk["value"])
        h["sketch_h2h_bp_s"] = _num((sk.get("host_to_host") or {}).get("value"))
        h["sketch_h2h_packed_bp_s"] = _num((sk.get("host_to_host_packed") or {}).get("value"))
        h["sketch_roofline_frac"] = (sk.get("roofline") or {}).get("frac")
    if "value" in c5:
        h["c5_pairs_s"], h["c5_warm_pairs_s"] = _num(c5["value"]), _num(c5.get("warm_value"))
        r5 = c5.get("roofline") or {}
        h["c5_roofline"] = {"step_frac": r5.get("step_frac"), "index_ms": _num(r5.get("index_ms")),
                            "traffic_over_compulsory": (r5.get("pass") or {}).get("traffic_over_compulsory")}
    if "value" in scr:
        h["screen_reads_s"] = _num(scr["value"])
    if cli:
        h["cli_e2e_speedup"] = {k: _num(v.get("speedup_vs_reference")) for k, v in cli.items() if isinstance(v, dict) and "speedup_vs_reference" in v}
    if br:
        h["brackets_pairs_s"] = {k: [_num(v.get("value")), _num(v.get("warm_value"))] for k, v in br.items() if isinstance(v, dict) and "value" in v}
    if "full_c3_thresholded" in hh:
        h["h2h_thresholded_pairs_s"] = _num(hh["full_c3_thresholded"].get("value"))
    if "full_c3_sparse" in hh:
        h["h2h_full_pairs_s"] = _num(hh["full_c3_sparse"].get("value"))
        # SURVEY 8d(i) defines the metric host -> host; `value` is the device-resident job (table and results stay in HBM)
        h["config"]["host_to_host_pairs_s"] = h["h2h_full_pairs_s"]
        h["config"]["value_is"] = "device-resident (table in, 8 B per pair out, both in HBM); host_to_host_pairs_s = table up, sparse result back"
    errs = [k for k in ("sketch", "c5", "screen", "cli_e2e", "brackets", "host_to_host") if "error" in (result.get(k) or {})]
    errs += [f"brackets.{k}" for k, v in br.items() if isinstance(v, dict) and "error" in v]
    if errs:
        h["leg_errors"] = errs
    h["detail"] = "bench_detail.json"
    return h


def emit(result, detail_path=None):
    """bench_detail.json (everything), a short line per secondary leg, and LAST the headline line."""
    detail = json.dumps(result)
    paths = [detail_path] if detail_path else [os.path.join(ROOT, "bench_detail.json"), os.path.join(ROOT, "gpurun_out", "bench_detail.json")]
    for f in paths:
        if os.path.isdir(os.path.dirname(os.path.abspath(f))):
            try:
                with open(f, "w") as fh:
                    fh.write(detail + "\n")
            except OSError:
                pass
    for leg in ("sketch", "screen", "c5", "host_to_host", "cli_e2e"):
        v = result.get(leg)
        if isinstance(v, dict):
            short = {"leg": leg}
            for k in ("value", "unit", "ms_per_step", "warm_value", "warm_ms_per_step", "error"):
                if k in v:
                    short[k] = _num(v[k], 5)
            rf = v.get("roofline")
            if isinstance(rf, dict):
                short["roofline"] = compact_roofline(rf) if "phases" in rf else {k: rf.get(k) for k in ("bound", "kernel", "kernel_ms", "achieved", "peak", "unit", "frac", "traffic")}
            line = json.dumps(short)
            if len(line) < 1200:
                print(line)
    line = json.dumps(headline_of(result), separators=(",", ":"))
    assert len(line) < 4000, f"headline line too long ({len(line)} bytes)"
    print(line, flush=True)


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from mash_amd import abi, shard
    from workloads import synth_torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    dry = args.dry_cpu
    if dry:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    class _DryEngine:                      # no kernels: only the control flow around them is exercised
        lib = None
        def table_wrap(self, *a, **k):
            return self
        def compare_tri_dev(self, table, rb, re, out_ptr):
            pass
        def prof_enable(self, on=True):
            pass
        def prof_reset(self):
            pass
        def prof_avg_ms(self, name):
            return 0.0, 0
        def free(self):
            pass
        def invalidate(self):
            pass
        def trim(self):
            pass
        def close(self):
            pass

    eng = _DryEngine() if dry else abi.MashGpu(local_rank, stream=torch.cuda.current_stream().cuda_stream)

    # ------------------------------------------------------------------ table: rank 0 builds it, the library broadcasts it
    n = args.n_sketches
    bcast_ms, rccl_ranks, comm_kind = 0.0, 0, "none"
    hashes = nhash = lengths = None
    if rank == 0 or dry:
        hashes, nhash, lengths = synth_torch.clustered_sketch_table(n, S, clusters=max(1, n // 100), device=dev)
    if dry and world > 1:
        barrier()
        t0 = time.perf_counter()
        for t in (hashes, nhash, lengths):
            dist.broadcast(t, 0)
        barrier()
        bcast_ms, comm_kind = (time.perf_counter() - t0) * 1e3, "torch.distributed/gloo (dry)"
    table = comm = None
    if not dry:
        torch.cuda.synchronize()
        root_table = eng.table_wrap(hashes.data_ptr(), nhash.data_ptr(), lengths.data_ptr(), n, S,
                                    keep=(hashes, nhash, lengths)) if rank == 0 else None
        if world > 1:
            # the library's own communicator: ncclCommInitRank on an id handed round by torch.distributed
            def exchange(raw):
                box = [raw]
                dist.broadcast_object_list(box, src=0)
                return box[0]
            ok = torch.ones(1, dtype=torch.int32, device=dev)
            try:
                comm = abi.RankComm(eng, world, rank, exchange)
            except Exception as e:                       # keep every rank in step: fall back together
                print(f"[bench] rank {rank}: library communicator unavailable ({e}); falling back to torch.distributed", file=sys.stderr)
                ok.zero_()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                barrier()
                t0 = time.perf_counter()
                table = comm.table_broadcast(root_table, n, S)     # the one exchange step: sketch table over xGMI
                barrier()
                bcast_ms = (time.perf_counter() - t0) * 1e3
                rccl_ranks, comm_kind = world, "libmashgpu/RCCL"
            else:
                comm = None
                if rank != 0:
                    hashes = torch.empty((n, S), dtype=torch.int64, device=dev)
                    nhash = torch.empty(n, dtype=torch.int32, device=dev)
                    lengths = torch.empty(n, dtype=torch.int64, device=dev)
                barrier()
                t0 = time.perf_counter()
                for t in (hashes, nhash, lengths):
                    dist.broadcast(t, 0)
                barrier()
                bcast_ms = (time.perf_counter() - t0) * 1e3
                comm_kind = "torch.distributed/RCCL"
                table = eng.table_wrap(hashes.data_ptr(), nhash.data_ptr(), lengths.data_ptr(), n, S, keep=(hashes, nhash, lengths))
        else:
            table = root_table
    else:
        table = eng.table_wrap(0, 0, 0, n, S)

    if dry:
        blocks = shard.equal_area_row_blocks(n, world)
        rb, re = blocks[rank], blocks[rank + 1]
    else:
        rb, re = abi.shard_tri_rows(eng.lib, 0, n, world, rank)
        blocks = [abi.shard_tri_rows(eng.lib, 0, n, world, g)[0] for g in range(world)] + [n]
    my_pairs = shard.tri_pairs(rb, re)
    total_pairs = n * (n - 1) // 2
    out = torch.empty((max(my_pairs, 1), 2), dtype=torch.int32, device=dev)

    def step():
        eng.compare_tri_dev(table, rb, re, out.data_ptr())

    if not dry:
        torch.cuda.synchronize()       # table generation (torch stream) -> library stream
    # Several ranks: equal AREAS balance the fill (8 B per pair), not discovery and merge, which cost per ROW -- on C3 a
    # row costs what 60 000 pairs cost, and the block of the short rows (a third of all rows at 8 ranks) would take twice
    # its share.  One measured step with the equal-area split tells what a row costs on THIS table; the blocks are then
    # cut so that pairs + row_weight x rows is the same for every rank (mg_shard_tri_rows_weighted).  Untimed.
    row_weight = 0.0
    if world > 1:
        # A rank's index covers the rows below its block's end only (host_compare.cpp: tri_view), so a block also costs per row
        # of that prefix: prefix_weight = index time per row of the view over fill time per pair (mg_shard_tri_rows_costed).
        if dry:                        # (figures of C3's order, so that the exchange and the re-cut run under gloo)
            fill_ms, dm_ms, ix_ms = my_pairs * 1.34e-9, (re - rb) * 8e-5, re * 7.9e-5
        else:
            eng.prof_enable(True)
            eng.prof_reset()
            table.invalidate()
            step()
            torch.cuda.synchronize()
            fill_ms = dm_ms = ix_ms = 0.0
            for name in ("compare_fill", "compare_fill_aside", "compare_discover", "compare_merge", "compare_index", "compare_join"):
                ms, k = eng.prof_avg_ms(name)
                if name == "compare_index":
                    ix_ms += ms * k
                elif name in ("compare_fill", "compare_fill_aside", "compare_join"):
                    fill_ms += ms * k                    # (what is paid per pair)
                else:
                    dm_ms += ms * k
            eng.prof_enable(False)
        cost = torch.tensor([fill_ms, dm_ms, float(my_pairs), float(re - rb), ix_ms, float(re)], dtype=torch.float64, device=dev)
        dist.all_reduce(cost)
        prefix_weight = 0.0
        if float(cost[0]) > 0 and float(cost[2]) > 0 and float(cost[3]) > 0:
            per_pair = float(cost[0]) / float(cost[2])
            row_weight = (float(cost[1]) / float(cost[3])) / per_pair
            if float(cost[5]) > 0 and os.environ.get("MASHGPU_TRI_PREFIX") != "0":
                prefix_weight = (float(cost[4]) / float(cost[5])) / per_pair
        if row_weight > 0 or prefix_weight > 0:
            if dry:
                blocks = shard.costed_row_blocks(n, world, row_weight, prefix_weight)
            else:
                blocks = [abi.shard_tri_rows_costed(eng.lib, 0, n, world, g, row_weight, prefix_weight)[0] for g in range(world)] + [n]
            rb, re = blocks[rank], blocks[rank + 1]
            my_pairs = shard.tri_pairs(rb, re)
            del out
            out = torch.empty((max(my_pairs, 1), 2), dtype=torch.int32, device=dev)
            if not dry:
                torch.cuda.synchronize()
    # ---- the timed region: the PER-TABLE job.  Every step starts from a table the library has not seen
    # (mg_table_invalidate drops the index and every plan; its blocks go back to the context's pool), as one
    # `mash triangle` pays it (CommandTriangle.cpp:129-139): index build + discover + fill + merge.
    def cold_step():
        table.invalidate()
        step()

    first_call_ms = None
    for w in range(args.warmup):
        tw = time.perf_counter()
        cold_step()
        if w == 0 and not dry:
            torch.cuda.synchronize()
            first_call_ms = (time.perf_counter() - tw) * 1e3    # the very first call of the process: also pays the hipMallocs
    eng.prof_enable(True)
    eng.prof_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cold_step()
    barrier()
    dt = time.perf_counter() - t0
    srcs = ("mash_amd/csrc/compare_sparse.hip", "mash_amd/csrc/compare_dense.hip", "mash_amd/csrc/compare_merged.hip", "mash_amd/csrc/compare_internal.h",
            "mash_amd/csrc/index_build.hip", "mash_amd/csrc/compare_join.hip")
    pmc = load_pmc("compare_c3_cold_pmc.json", *srcs) if (n == 100_000 and world == 1 and not dry) else None
    roofline = compare_roofline(eng, my_pairs, n, S, args.steps, pmc) if not dry else {"bound": "hbm", "dry": True}
    dt = max_over_ranks(dt)
    value = total_pairs * args.steps / dt
    # what every rank's index cost (a rank's index covers the rows below its block's end only: host_compare.cpp tri_view)
    index_ms_by_rank = None
    if world > 1:
        mine = 0.0 if dry else float((roofline.get("phases") or {}).get("index", {}).get("ms_per_pass") or 0.0)
        tens = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(tens, torch.tensor([mine], dtype=torch.float64, device=dev))
        index_ms_by_rank = [round(float(x.item()), 3) for x in tens]
    if not dry and "pass" in roofline:
        roofline["step_frac"] = round(roofline["pass"]["compulsory_bytes"] / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4)
        roofline["index_ms"] = roofline["phases"].get("index", {}).get("ms_per_pass")
    # further passes over the same table (index and plan exist): the warm rate
    eng.prof_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    wdt = max_over_ranks(time.perf_counter() - t0)
    pmc_w = load_pmc("compare_c3_pmc.json", *srcs) if (n == 100_000 and world == 1 and not dry) else None
    roofline_warm = compare_roofline(eng, my_pairs, n, S, args.steps, pmc_w) if not dry else None
    eng.prof_enable(False)
    # the dominant kernel BY ITSELF (a further pass over the indexed table launches it once, nothing beside it): what the
    # per-table step's paced launches are to be read against
    if roofline_warm and roofline.get("beside") and roofline_warm.get("kernel") == "mg::sp_fill_value_kernel":
        roofline["alone"] = {"kernel": roofline_warm["kernel"], "kernel_ms": roofline_warm["kernel_ms"], "achieved": roofline_warm["achieved"],
                             "frac": roofline_warm["frac"]}

    # the produced output (outside the timed region): every pair's denom and numer, as sums
    checksum = None
    if not dry:
        sums = torch.stack([out[:my_pairs, 0].sum(dtype=torch.int64), out[:my_pairs, 1].sum(dtype=torch.int64)])
        if world > 1:
            dist.all_reduce(sums)
        checksum = [int(sums[0].item()), int(sums[1].item())]
        want = C3_CHECKSUM.get((n, S))
        assert int(out[:my_pairs, 1].max()) <= S and int(out[:my_pairs, 0].max()) <= S, "compare output failed sanity check"
        if want is not None:
            assert checksum == list(want), f"compare output checksum {checksum} != verified {want}"

    result = {
        "metric": "pairwise Mash distances/sec (s=1000)",
        "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"mash triangle all-vs-all, {n} clustered synthetic sketches, k={K} s={S}, {total_pairs} pairs/step, "
                               f"per-table job (index build + discover + fill + merge each step), row-block sharded x{world}",
                   "n_sketches": n, "sketch_size": S, "kmer": K, "hash_bits": 64,
                   "parallelism": f"rowblock{world}", "rank_row_blocks": blocks, "row_weight_pairs": round(row_weight, 1),
                   "prefix_weight_pairs": round(prefix_weight, 1) if world > 1 else 0.0,
                   "table_broadcast_ms": round(bcast_ms, 2), "index_ms_by_rank": index_ms_by_rank,
                   "rccl_ranks": rccl_ranks, "comm": comm_kind, "output_checksum": checksum,
                   "first_call_ms": round(first_call_ms, 2) if first_call_ms is not None else None},
        "warm_value": total_pairs * args.steps / wdt, "warm_ms_per_step": wdt * 1e3 / args.steps,
        "roofline": roofline, "roofline_warm": roofline_warm,
    }
    if dry:
        result["dry"] = True               # plumbing test: NOT a measurement
        result["rank_blocks"] = blocks

    single = rank == 0 and world == 1 and not dry
    # ------------------------------------------------------------------ cpu baseline (rank 0, N=1)
    if single and not args.no_cpu:
        m = min(n, 40000)
        # P = nproc (the box's CPU rate: `cpu_baseline`), P = 64, 16 and 1 beside it; the reference CLI
        # itself (`mash-ref triangle -p nproc`) as a second figure of the same box
        sub = (hashes[:m].cpu().numpy().view(np.uint64), nhash[:m].cpu().numpy().astype(np.uint32), lengths[:m].cpu().numpy().astype(np.uint64))
        nproc, usable = os.cpu_count() or 1, usable_cpus()
        by = {}
        for c in sorted({1, min(16, nproc), usable, min(64, nproc), nproc}):
            by[c] = cpu_baseline_compare(*sub, args.cpu_seconds / 5.0, cores=c)
        # the box's rate: the best of them (threads beyond the cores the container may use only get in each other's way), with
        # the reference CLI timed at that thread count
        top = max(v["value"] for v in by.values())
        best = min(c for c in by if by[c]["value"] >= 0.97 * top)      # (the fewest threads that reach it: more only share the same cores)
        result["cpu_baseline"] = cpu_baseline_compare(*sub, args.cpu_seconds / 2.0, cores=best, cli=True)
        result["cpu_baseline"]["host"] = {"cpu_count": nproc, "usable": usable}
        result["cpu_baseline_by_cores"] = {str(c): v for c, v in by.items()}

    # ------------------------------------------------------------------ SURVEY 8d brackets (N=1): the extremes of the merge
    # value = the per-table job (every step from an invalidated table), warm_value = further passes
    if single and not args.no_brackets:
        br = {}
        n1 = min(n, 32768)
        gens = [("all_random", "random", n, lambda: synth_torch.random_sketch_table(n, S, device=dev)),
                ("all_identical", "identical", n, lambda: synth_torch.identical_sketch_table(n, S, device=dev)),
                ("clades_of_1000", "clades", n, lambda: synth_torch.clade_sketch_table(n, S, device=dev)),
                # the worst case of an engine that pays per candidate: ONE clade -- every pair shares ~900 of 1000 hashes
                ("one_clade", "one_clade", n1, lambda: synth_torch.clade_sketch_table(n1, S, clade=n1, device=dev)),
                # the middle of the similarity range: ONE species as a tree of descent -- every pair shares
                # 10 - 50 % of its hashes, no near-copies, no small common pool, rows in random order
                ("one_species", "one_species", n1, lambda: synth_torch.species_sketch_table(n1, S, device=dev))]
        bsteps = max(2, min(args.steps, 5))
        for name, leg, bn_rows, gen in gens:
            try:
                bh, bn, bl = gen()
                bpairs = bn_rows * (bn_rows - 1) // 2
                torch.cuda.synchronize()
                bt = eng.table_wrap(bh.data_ptr(), bn.data_ptr(), bl.data_ptr(), bn_rows, S, keep=(bh, bn, bl))
                eng.compare_tri_dev(bt, 0, bn_rows, out.data_ptr())                 # warm-up (pool, plan)
                eng.prof_enable(True)
                eng.prof_reset()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(bsteps):
                    bt.invalidate()
                    eng.compare_tri_dev(bt, 0, bn_rows, out.data_ptr())
                torch.cuda.synchronize()
                bd = time.perf_counter() - t0
                rf_cold = compare_roofline(eng, bpairs, bn_rows, S, bsteps, None)
                eng.prof_reset()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(bsteps):
                    eng.compare_tri_dev(bt, 0, bn_rows, out.data_ptr())
                torch.cuda.synchronize()
                bw = time.perf_counter() - t0
                bpmc = load_pmc(f"compare_{leg}_pmc.json", *srcs) if n == 100_000 else None
                rf = compare_roofline(eng, bpairs, bn_rows, S, bsteps, bpmc)
                eng.prof_enable(False)
                o = out[:bpairs]
                sums = [int(o[:, 0].sum(dtype=torch.int64).item()), int(o[:, 1].sum(dtype=torch.int64).item())]
                if name == "all_identical":
                    assert sums == [bpairs * S, bpairs * S], f"{name}: {sums}"
                    how = "closed form: every pair s/s"
                else:
                    # an independent implementation on the same table: the tile engine (every pair), or -- where that one is
                    # the engine under test -- the generic kernel on the last rows
                    os.environ["MASHGPU_COMPARE_KERNEL"] = "merged" if rf.get("engine") != "tiles" else "generic"
                    try:
                        rb2 = 0 if rf.get("engine") != "tiles" else bn_rows - 64
                        keep = o[rb2 * (rb2 - 1) // 2 if rb2 else 0:].clone()
                        o.zero_()
                        eng.compare_tri_dev(bt, rb2, bn_rows, out.data_ptr())
                        torch.cuda.synchronize()
                    finally:
                        os.environ.pop("MASHGPU_COMPARE_KERNEL", None)
                    assert torch.equal(keep, out[:keep.shape[0]]), f"{name}: engines disagree"
                    assert sums[1] == bpairs * S, f"{name}: {sums}"
                    how = ("every pair equal to the tile engine's (compare_merged.hip)" if rb2 == 0 else
                           "the last 64 rows equal to the generic kernel's (compare.hip)")
                br[name] = {"value": bpairs * bsteps / bd, "unit": "pairs/s", "ms_per_step": bd * 1e3 / bsteps, "steps": bsteps,
                            "warm_value": bpairs * bsteps / bw, "warm_ms_per_step": bw * 1e3 / bsteps, "n_sketches": bn_rows,
                            "mean_shared_hashes": round(sums[0] / bpairs, 3), "output_checksum": sums, "verified": how,
                            "roofline": rf_cold, "roofline_warm": rf}
                if not args.no_cpu:
                    m = min(bn_rows, 3000)
                    br[name]["cpu_baseline"] = cpu_baseline_compare(bh[:m].cpu().numpy().view(np.uint64), bn[:m].cpu().numpy().astype(np.uint32),
                                                                    bl[:m].cpu().numpy().astype(np.uint64), min(args.cpu_seconds, 4.0))
                bt.free()
                del bh, bn, bl
            except Exception as e:
                br[name] = {"error": repr(e)}
        # what it costs when the index's bucket sorts REFUSE a table: their order check fires -- here forced by
        # the test knob that swaps two entries of one value behind the partition -- the tile build is thrown away and the general
        # sort builds the index (round 4's build); the headline's table, per-table job, same checksum
        try:
            os.environ["MASHGPU_IX_DEBUG_SWAP"] = "1"
            eng.prof_enable(True)
            eng.prof_reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(bsteps):
                table.invalidate()
                eng.compare_tri_dev(table, rb, re, out.data_ptr())
            torch.cuda.synchronize()
            bd = time.perf_counter() - t0
            ims, ik = eng.prof_avg_ms("compare_index")
            eng.prof_enable(False)
            sums = [int(out[:my_pairs, 0].sum(dtype=torch.int64).item()), int(out[:my_pairs, 1].sum(dtype=torch.int64).item())]
            assert checksum is None or sums == checksum, f"refused index: {sums}"
            br["c3_index_refused"] = {"value": total_pairs * bsteps / bd, "unit": "pairs/s", "ms_per_step": bd * 1e3 / bsteps, "steps": bsteps,
                                      "index_ms": round(ims * ik / bsteps, 3), "output_checksum": sums,
                                      "verified": "the headline's checksum; the bucket sorts' order check fired in every step"}
        except Exception as e:
            br["c3_index_refused"] = {"error": repr(e)}
        finally:
            os.environ.pop("MASHGPU_IX_DEBUG_SWAP", None)
            table.invalidate()
        br["workload"] = (f"mash triangle on {n} sketches of s={S}: all-random (every sketch its own values), all-identical (n copies "
                          f"of one sketch), clades of 1000 near-identical sketches (consecutive rows); one clade of {n1} distinct near-copies; "
                          f"one species of {n1} sketches as a tree of descent (pairs share 100 - 500 of 1000 hashes, random row order)")
        result["brackets"] = br

    # ------------------------------------------------------------------ host to host, N=1
    del out
    if single and not args.no_h2h:
        h2h = {}
        try:
            m = min(n, 16384)
            th, tn, tl = (hashes[:m].cpu().numpy().view(np.uint64), nhash[:m].cpu().numpy().astype(np.uint32),
                          lengths[:m].cpu().numpy().astype(np.uint64))
            mp = m * (m - 1) // 2
            buf_c = np.zeros(mp, dtype=abi.COUNTS_DTYPE)
            buf_p = np.zeros(mp, dtype=abi.PAIR_DTYPE)
            buf_c[:] = 0
            buf_p["pass"] = 0                                        # pages exist before the clock starts
            for kind in ("counts", "pairs"):
                best = None
                for _ in range(2):
                    t0 = time.perf_counter()
                    t = eng.table_upload(th, tn, tl)
                    if kind == "counts":
                        eng.compare_tri_host(t, out=buf_c)
                    else:
                        eng._check(eng.lib.mg_compare_tri_pairs_host(eng.ctx, t.handle, 0, m, K, 4.0 ** K, -1.0, -1.0, buf_p.ctypes.data))
                    d = time.perf_counter() - t0
                    t.free()
                    best = d if best is None else min(best, d)
                h2h[kind] = {"value": mp / best, "unit": "pairs/s", "ms": round(best * 1e3, 2),
                             "bytes_per_pair_over_pcie": 8 if kind == "counts" else 32}
            h2h["sample"] = (f"first {m} sketches of the C3 table: table in host memory -> "
                             f"{{numer, denom}} (counts) / {{numer, denom, distance, p-value}} (pairs, device finish) in host "
                             f"memory, {mp} pairs, upload + compare + copy back, pageable memory, best of 2")
            # the whole C3 triangle host to host through the thresholded path (only survivors cross PCIe)
            hh, hn, hl = (hashes.cpu().numpy().view(np.uint64), nhash.cpu().numpy().astype(np.uint32),
                          lengths.cpu().numpy().astype(np.uint64))
            d = None
            for _ in range(2):
                t0 = time.perf_counter()
                t = eng.table_upload(hh, hn, hl)
                res = eng.compare_tri_results(t, K, 4.0 ** K, 0.05, 1.0, capacity=1 << 23)
                d1 = time.perf_counter() - t0
                t.free()
                d = d1 if d is None else min(d, d1)
            h2h["full_c3_thresholded"] = {"value": total_pairs / d, "unit": "pairs/s", "ms": round(d * 1e3, 1),
                                          "survivors": int(len(res)),
                                          "what": "C3 table in host memory -> every pair with distance <= 0.05 as "
                                                  "{row, col, numer, denom, distance, p-value} in host memory"}
            del res
            # ... and with FULL information: the triangle as its exceptions (mg_compare_tri_sparse_host: every pair with
            # numer >= 1; every other pair is {0, min(s, |A| + |B|)}) -- SURVEY 8d(i)'s host-to-host metric for the WHOLE
            # matrix without 40 GB over PCIe.  Checked: as many records as the timed output has non-zero numer, the same sum.
            best, edges = None, None
            for _ in range(2):
                t0 = time.perf_counter()
                t = eng.table_upload(hh, hn, hl)
                edges = eng.compare_tri_sparse(t, capacity=max(1 << 23, total_pairs // 512))
                d = time.perf_counter() - t0
                t.free()
                best = d if best is None else min(best, d)
            ok = checksum is None or (int(edges["numer"].sum(dtype=np.uint64)) == checksum[0])
            assert ok, "sparse matrix: the exceptions' numer do not add up to the matrix's"
            h2h["full_c3_sparse"] = {"value": total_pairs / best, "unit": "pairs/s", "ms": round(best * 1e3, 1), "exceptions": int(len(edges)),
                                     "bytes_over_pcie": int(hh.nbytes + hn.nbytes + hl.nbytes + edges.nbytes),
                                     "what": "C3 table in host memory -> EVERY pair's {numer, denom} in host memory as the rule "
                                             "{0, min(s, |A|+|B|)} + its exceptions {row, col, numer, denom} (mg_compare_tri_sparse_host), "
                                             "upload + index + discover + merge + dense groups as lists + copy back, best of 2; "
                                             "sum of numer == the timed matrix's"}
            del hh, edges
        except Exception as e:
            h2h["error"] = repr(e)
        result["host_to_host"] = h2h

    # ------------------------------------------------------------------ secondary: sketching (config 2)
    if not args.no_sketch and not dry:
        g_blocks = shard.even_blocks(args.n_genomes, world)
        g0, g1 = g_blocks[rank], g_blocks[rank + 1]
        ng, L = g1 - g0, args.genome_len
        bases = synth_torch.synthetic_genomes(g0, g1, L, device=dev)
        off = np.arange(ng + 1, dtype=np.uint64) * np.uint64(L)
        sk_hashes = torch.empty((max(ng, 1), S), dtype=torch.int64, device=dev)
        sk_nhash = torch.empty(max(ng, 1), dtype=torch.int32, device=dev)
        p = eng.params(k=K, s=S)

        def sk_step():
            eng.sketch_dev(bases.data_ptr(), ng * L, off, p, sk_hashes.data_ptr(), sk_nhash.data_ptr())

        torch.cuda.synchronize()
        sk_step()
        eng.prof_enable(True)
        eng.prof_reset()
        barrier()
        t0 = time.perf_counter()
        sk_steps = max(2, args.steps)
        for _ in range(sk_steps):
            sk_step()
        barrier()
        sdt = time.perf_counter() - t0
        sk_ms, sk_launches = eng.prof_avg_ms("sketch")
        eng.prof_enable(False)
        sdt = max_over_ranks(sdt)
        assert int(sk_nhash.min()) == S, "sketch output failed sanity check"
        sk_bytes = ng * (L + 8 * S)                   # 1 B/base in + 8*s B per sketch out
        sk_pmc = load_pmc("sketch_pmc_latest.json", "mash_amd/csrc/sketch.hip", "mash_amd/csrc/kmer_hash.h") \
            if (world == 1 and args.n_genomes == 10_000 and L == 1_000_000) else None
        sk_ach = sk_bytes / (sk_ms * 1e-3) / 1e9 if sk_ms > 0 else 0.0
        sk_valu = None
        if sk_pmc and sk_ms > 0:
            # the VALU port on its own (never summed with the scalar port: they issue side by side).  Issue cost measured
            # with tools/ubench_valu.hip at 2-8 waves per SIMD (profiles/r04_ubench_valu.txt): ~2.7 cycles per wave64
            # two-operand instruction, ~4.3 per three-operand one (v_alignbit, v_mad_u64_u32, v_add3, v_perm, v_mul_lo)
            # (valu_per_kmer: wave-instructions per k-mer = instructions a lane executes per k-mer / 64)
            ach = sk_pmc["valu_per_kmer"] * ng * (L - K + 1) / (sk_ms * 1e-3)              # wave-instructions / s
            sk_valu = {"valu_lane_instr_per_kmer": round(sk_pmc["valu_per_kmer"] * 64, 1), "achieved": round(ach / 1e9, 2), "unit": "G wave-instr/s",
                       "cycles_per_wave_instr": round(SIMDS * CLOCK_HZ / ach, 3),
                       "peak_2_cycle": round(SIMDS * CLOCK_HZ / 2.0 / 1e9, 1), "peak_4_cycle": round(SIMDS * CLOCK_HZ / 4.0 / 1e9, 1),
                       "frac_of_2_cycle_peak": round(ach / (SIMDS * CLOCK_HZ / 2.0), 4), "source": sk_pmc.get("source")}
        sketch = {"metric": "sketched bp/sec (k=21, s=1000)", "value": args.n_genomes * L * sk_steps / sdt,
                  "unit": "bp/s", "ms_per_step": sdt * 1e3 / sk_steps, "steps": sk_steps,
                  "config": {"workload": f"sketch {args.n_genomes} synthetic {L} bp genomes, k={K} s={S}, "
                                         f"ASCII bases resident in HBM, sharded x{world}"},
                  "roofline": {"bound": "hbm", "achieved": round(sk_ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(sk_ach / HBM_PEAK_GBS, 4),
                               "traffic": sk_pmc.get("hbm_bytes_per_launch") if sk_pmc else None,
                               "kernel": "sketch_chunks_kernel<21,0,256,false>", "kernel_ms": round(sk_ms, 3),
                               "launches": sk_launches, "valu": sk_valu}}
        if single and not args.no_cpu:
            cores = min(os.cpu_count() or 1, 16)
            sketch["cpu_baseline"] = cpu_baseline_sketch(min(args.cpu_seconds, 6.0), 1)
            sketch["cpu_baseline_all_cores"] = cpu_baseline_sketch(min(args.cpu_seconds, 6.0), cores)
            cli = cpu_baseline_sketch_cli(200, cores)
            if cli:
                sketch["cpu_baseline_cli"] = cli
        if single and not args.no_h2h:
            # FASTA bytes in HOST memory -> hash lists in HOST memory
            try:
                mg = min(ng, 2000)
                hb = bases[:mg].cpu().numpy().reshape(-1)
                hoff = np.arange(mg + 1, dtype=np.uint64) * np.uint64(L)
                best = None
                for _ in range(2):
                    t0 = time.perf_counter()
                    eng.sketch_host_raw(hb, hoff, p)
                    d = time.perf_counter() - t0
                    best = d if best is None else min(best, d)
                sketch["host_to_host"] = {"value": mg * L / best, "unit": "bp/s", "ms": round(best * 1e3, 1),
                                          "sample": f"{mg} genomes x {L} bp in pageable host memory -> hashes in host memory "
                                                    f"(mg_sketch_host: H2D + kernel + D2H), best of 2"}
                # the same bases handed over PACKED (mg_pack_bases: 2 bits + 1 invalid bit per base, 0.375 B/base over PCIe;
                # mg_sketch_host_packed copies piece i + 1 while piece i is sketched); the packing itself is what parse
                # threads would do while they read, reported on its own
                ascii_h, ascii_n = eng.sketch_host_raw(hb, hoff, p)
                pk_threads = min(os.cpu_count() or 1, 64)
                t0 = time.perf_counter()
                packed, mask, ninv = abi.pack_bases(hb, threads=pk_threads)
                t_pack = time.perf_counter() - t0
                bestp = None
                for _ in range(3):
                    t0 = time.perf_counter()
                    ph, pn = eng.sketch_host_packed_raw(packed, mask, len(hb), hoff, p)
                    d = time.perf_counter() - t0
                    bestp = d if bestp is None else min(bestp, d)
                assert np.array_equal(ph, ascii_h) and np.array_equal(pn, ascii_n), "packed input: other sketches than the ASCII path"
                sketch["host_to_host_packed"] = {"value": mg * L / bestp, "unit": "bp/s", "ms": round(bestp * 1e3, 1),
                                                 "bytes_per_base_over_pcie": 0.375, "same_sketches_as_ascii_path": True,
                                                 "pack": {"value": mg * L / t_pack, "unit": "bp/s", "threads": pk_threads},
                                                 "sample": f"the same {mg} genomes as packed codes + invalid mask in pageable host memory "
                                                           f"(mg_sketch_host_packed), best of 3"}
                del packed, mask, ph, pn, ascii_h, ascii_n
            except Exception as e:
                sketch["host_to_host"] = {"error": repr(e)}
        result["sketch"] = sketch
        bases = sk_hashes = sk_nhash = None

    # ------------------------------------------------------------------ tertiary: screen (config 4)
    # 10^7 x 150 bp reads (0.5 % errors, both strands) against the first 10^5 - 10^3 rows of the
    # C3 table + the real sketches of the 10^3 genomes the reads come from.  Reads are sharded
    # by batch; the one collective is the all-reduce of the observation counters (RCCL) plus
    # an all-gather of the per-rank mixture sketches (mash_amd/screen_dist.py).  A step is the
    # whole job: table build, every batch, counters gathered + exchanged.
    if not args.no_screen and not dry:
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        from mash_amd import screen_dist
        scr = {"metric": "screened reads/sec (150 bp, k=21, s=1000, 100k-sketch database)", "unit": "reads/s"}
        ok = torch.ones(1, dtype=torch.int32, device=dev)
        try:
            RL, NSRC, GL = 150, 1000, 1_000_000
            p = eng.params(k=K, s=S)
            genomes = synth_torch.synthetic_genomes(0, NSRC, GL, device=dev, stride=40000)
            gh = torch.empty((NSRC, S), dtype=torch.int64, device=dev)
            gn = torch.empty(NSRC, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()                 # torch stream -> library stream
            eng.sketch_dev(genomes.data_ptr(), NSRC * GL, np.arange(NSRC + 1, dtype=np.uint64) * np.uint64(GL), p,
                           gh.data_ptr(), gn.data_ptr())
            rest = max(0, min(n, 100_000) - NSRC)
            if hashes is None:                           # ranks > 0: the same table again (the generator is deterministic)
                fh_all, fn_all, _ = synth_torch.clustered_sketch_table(n, S, clusters=max(1, n // 100), device=dev)
                fh, fn = fh_all[:rest], fn_all[:rest]
            else:
                fh, fn = hashes[:rest], nhash[:rest]
            db_h = torch.cat([gh, fh], 0).contiguous()
            db_n = torch.cat([gn, fn], 0).contiguous()
            db_l = torch.full((NSRC + rest,), GL, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            db = eng.table_wrap(db_h.data_ptr(), db_n.data_ptr(), db_l.data_ptr(), NSRC + rest, S, keep=(db_h, db_n, db_l))
            nb = max(4, world)                        # >= one batch per rank; few, large batches
            per_batch = (args.n_reads + nb - 1) // nb
            mine = screen_dist.shard_batches(nb, rank, world)
            batches = [synth_torch.synthetic_reads(genomes, min(per_batch, args.n_reads - b * per_batch), RL, seed=7000 + b)
                       for b in mine]
            del genomes
            torch.cuda.synchronize()
            # The database stays RESIDENT: its key table (and two-tier bound) is built once, before the timed steps
            # (`create_ms`), as the inverted index of a compare table is; a step = reset + every batch + results.
            # One rank: the results are the sparse hits in host memory (what the CLI reads).  Several ranks: the
            # dense counters, summed by the collective.
            t_create = time.perf_counter()
            handles = [(b.data_ptr(), int(b.numel()), b) for b in batches]
            if world == 1:
                session = eng.screen_open(db, p)
                torch.cuda.synchronize()
                scr["create_ms"] = round((time.perf_counter() - t_create) * 1e3, 2)
                scr["key_bound"] = session.tier_note()

                def scr_step():
                    session.reset()
                    for h in handles:
                        session.add_dev(h[0], h[1])
                    hits, mix, _ = session.finish_sparse()
                    return hits, mix
            else:
                local = screen_dist.gpu_local_screen(eng, db, p, resident=True)

                def scr_step():
                    # a local failure must not leave the other ranks alone in the collective
                    try:
                        counts, mix = local(handles)
                    except Exception as e:
                        scr["error"] = repr(e)
                        ok.zero_()
                        counts = torch.zeros((NSRC + rest) * S, dtype=torch.int32, device=dev)
                        mix = np.zeros(0, dtype=np.uint64)
                    return screen_dist.exchange(counts, mix, S)
        except Exception as e:                      # keep every rank in step for the collectives below
            ok.zero_()
            scr["error"] = repr(e)
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            counts, mix = scr_step()
            barrier()
            t0 = time.perf_counter()
            scr_steps = max(5, args.steps)                      # (12 ms each: one stalled step of two halved the rate once)
            for _ in range(scr_steps):
                counts, mix = scr_step()
            barrier()
            qdt = max_over_ranks(time.perf_counter() - t0)
            if world > 1:
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            if world == 1:
                shared = float((counts["row"] < NSRC).sum()) / NSRC          # (`counts` holds the hits here)
                scr["hits"] = int(len(counts))
                session.close()
            else:
                shared = (counts.view(NSRC + rest, S)[:NSRC] > 0).sum(1).float().mean().item()
                local.close()
            # HBM bytes of the fused sketch + probe kernel per step (PMC, per read x reads; stamped like the others)
            spmc = load_pmc("screen_pmc_latest.json", "mash_amd/csrc/sketch.hip", "mash_amd/csrc/kmer_hash.h", "mash_amd/csrc/screen.hip") \
                if (world == 1 and args.n_reads == 10_000_000) else None
            scr_traffic = spmc["hbm_bytes_per_pass"] / spmc["units_per_pass"] * args.n_reads if spmc else None
            assert 500 < shared < 900 and len(mix) == S, f"screen output failed sanity check (shared {shared}, mix {len(mix)})"
            scr.update({"value": args.n_reads * scr_steps / qdt, "ms_per_step": qdt * 1e3 / scr_steps, "steps": scr_steps,
                        "bp_per_s": args.n_reads * RL * scr_steps / qdt,
                        "config": {"workload": f"mash screen: {args.n_reads} synthetic {RL} bp reads (0.5% errors) vs "
                                               f"{NSRC + rest} sketches ({(NSRC + rest) * S} keys), reads resident in HBM, "
                                               f"batch-sharded x{world}, " + ("database resident, sparse hits to host memory" if world == 1 else "database resident, counters all-reduced"),
                                   "mean_shared_hashes_of_sampled_genomes": round(shared, 1)},
                        "roofline": {"bound": "hbm", "achieved": round(args.n_reads * (RL + 1) * scr_steps / qdt / 1e9, 1),
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(args.n_reads * (RL + 1) * scr_steps / qdt / 1e9 / HBM_PEAK_GBS, 4),
                                     "traffic": scr_traffic, "kernel": "sketch_chunks_kernel<21,0,256> (fused table probe)",
                                     "note": "1 B/base streamed once; integer-ALU bound like sketching (one murmur per "
                                             "k-mer), table probes filtered by the key bound; whole-step time: reset, every batch, "
                                             "results (table build once per database: create_ms)"}})
            db.free()
            # --- a database that mixes genome sizes: 10 000 of the rows become virus-sized sketches
            # (30 kbp: their bottom-s hashes reach 1/32 of the hash range, those of the 1 Mbp rows 1/1000), so the
            # largest key stops bounding anything -- with one tier every 32nd k-mer of the mixture probes the table.
            if world == 1 and not args.no_brackets:
                try:
                    nv = min(10_000, (NSRC + rest) // 10)
                    vh, vn, vl = synth_torch.random_sketch_table(nv, S, seed=5, bits=59, length=30_000, device=dev)
                    keep = NSRC + rest - nv
                    mh = torch.cat([db_h[:keep], vh], 0).contiguous()
                    mn = torch.cat([db_n[:keep], vn], 0).contiguous()
                    ml = torch.cat([db_l[:keep], vl], 0).contiguous()
                    torch.cuda.synchronize()
                    mdb = eng.table_wrap(mh.data_ptr(), mn.data_ptr(), ml.data_ptr(), keep + nv, S, keep=(mh, mn, ml))
                    mixed = {"workload": f"the same {args.n_reads} reads vs {keep} sketches of 1 Mbp genomes + {nv} of 30 kbp genomes"}
                    seen = {}
                    for tiers in ("1", "0"):
                        os.environ["MASHGPU_SCREEN_TIERS"] = tiers
                        with eng.screen_open(mdb, p) as ms:
                            note = ms.tier_note()
                            def mstep():
                                ms.reset()
                                for h in handles:
                                    ms.add_dev(h[0], h[1])
                                return ms.finish_sparse()[0]
                            mstep()
                            torch.cuda.synchronize()
                            t0 = time.perf_counter()
                            for _ in range(scr_steps):
                                mh_hits = mstep()
                            mdt = (time.perf_counter() - t0) / scr_steps
                        seen[tiers] = mh_hits
                        mixed["two_tiers" if tiers == "1" else "one_tier"] = {"ms_per_step": round(mdt * 1e3, 2), "value": args.n_reads / mdt,
                                                                              "unit": "reads/s", "key_bound": note}
                    del os.environ["MASHGPU_SCREEN_TIERS"]
                    assert np.array_equal(seen["1"], seen["0"]), "two-tier and one-tier screens differ"
                    mixed["hits"] = int(len(seen["1"]))
                    mixed["verified"] = "hits of the two-tier and the one-tier screen identical"
                    scr["mixed_database"] = mixed
                    mdb.free()
                    del mh, mn, ml, vh
                except Exception as e:
                    os.environ.pop("MASHGPU_SCREEN_TIERS", None)
                    scr["mixed_database"] = {"error": repr(e)}
            del db_h, db_n, counts
        elif "error" not in scr:
            scr["error"] = "failed on another rank"
        result["screen"] = scr
        batches = handles = None

    # ------------------------------------------------------------------ config 5: triangle at s = 10 000 (N = 1 only)
    if single and not args.no_c5:
        import gc
        c5 = {"metric": "pairwise Mash distances/sec (s=10000, 64-bit hashes)", "unit": "pairs/s"}
        try:
            table.free()
            table = None
            hashes = nhash = lengths = None
            gc.collect()
            torch.cuda.empty_cache()
            eng.trim()                                   # the s = 1000 index blocks are of no use to this shape
            S5 = 10000
            n5 = n
            h5, nh5, l5 = synth_torch.clustered_sketch_table(n5, S5, clusters=max(1, n5 // 100), pool=15000, private=4000,
                                                             device=dev, block=2000)
            torch.cuda.synchronize()
            t5 = eng.table_wrap(h5.data_ptr(), nh5.data_ptr(), l5.data_ptr(), n5, S5, keep=(h5, nh5, l5))
            pairs5 = n5 * (n5 - 1) // 2
            out5 = torch.empty((pairs5, 2), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            tc = time.perf_counter()
            eng.compare_tri_dev(t5, 0, n5, out5.data_ptr())            # the first call of this shape: also pays the hipMallocs (32 GB of index)
            torch.cuda.synchronize()
            first5 = (time.perf_counter() - tc) * 1e3
            eng.prof_enable(True)
            eng.prof_reset()
            torch.cuda.synchronize()
            steps5 = 2
            t0 = time.perf_counter()
            for _ in range(steps5):                                    # the per-table job
                t5.invalidate()
                eng.compare_tri_dev(t5, 0, n5, out5.data_ptr())
            torch.cuda.synchronize()
            d5 = time.perf_counter() - t0
            srcs5 = srcs
            rf5_cold = compare_roofline(eng, pairs5, n5, S5, steps5, load_pmc("compare_c5_cold_pmc.json", *srcs5) if n5 == 100_000 else None)
            if "pass" in rf5_cold:
                rf5_cold["step_frac"] = round(rf5_cold["pass"]["compulsory_bytes"] / (d5 / steps5) / 1e9 / HBM_PEAK_GBS, 4)
                rf5_cold["index_ms"] = rf5_cold["phases"].get("index", {}).get("ms_per_pass")
            eng.prof_reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps5):                                    # further passes
                eng.compare_tri_dev(t5, 0, n5, out5.data_ptr())
            torch.cuda.synchronize()
            w5 = time.perf_counter() - t0
            pmc5 = load_pmc("compare_c5_pmc.json", *srcs5) if n5 == 100_000 else None
            rf5 = compare_roofline(eng, pairs5, n5, S5, steps5, pmc5)
            eng.prof_enable(False)
            sums5 = [int(out5[:, 0].sum(dtype=torch.int64).item()), int(out5[:, 1].sum(dtype=torch.int64).item())]
            assert int(out5[:, 1].min()) == S5 and int(out5[:, 0].max()) <= S5, "c5 output failed sanity check"
            want5 = C3_CHECKSUM.get((n5, S5))
            if want5 is not None:
                assert sums5 == list(want5), f"c5 checksum {sums5} != verified {want5}"
            c5.update({"value": pairs5 * steps5 / d5, "ms_per_step": d5 * 1e3 / steps5, "steps": steps5,
                       "warm_value": pairs5 * steps5 / w5, "warm_ms_per_step": w5 * 1e3 / steps5,
                       "config": {"workload": f"mash triangle, {n5} clustered synthetic sketches of s={S5} 64-bit hashes "
                                              f"(k=31 style), {pairs5} pairs/step, per-table job, 1 GPU", "output_checksum": sums5,
                                  "first_call_ms": round(first5, 1)},
                       "roofline": rf5_cold, "roofline_warm": rf5})
            t5.free()
            del out5, h5
        except Exception as e:
            c5["error"] = repr(e)
        result["c5"] = c5

    # --- the CLI end to end: `mash sketch -p 16`, ours and the reference CLI on the same files
    if single and not args.no_cli and not dry:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import sketch_e2e
            torch.cuda.synchronize()
            sk_cli = sketch_e2e.run(genomes=12000, length=50000, threads=16, reps=3)
            sk_cli["speedup_vs_reference"] = sk_cli.get("speedup_vs_ref_same_threads")
            result["cli_e2e"] = {"sketch": sk_cli, "host_cores": os.cpu_count()}
        except Exception as e:
            result["cli_e2e"] = {"error": repr(e)}
        # ... and the compare commands: `mash triangle`, `triangle -E -d`, `dist -d` (tools/compare_e2e.py)
        try:
            import compare_e2e
            ct = synth_torch.clustered_sketch_table(n, S, clusters=max(1, n // 100), device=dev)
            tab = (ct[0].cpu().numpy().view(np.uint64), ct[1].cpu().numpy().astype(np.uint32), ct[2].cpu().numpy().astype(np.uint64))
            del ct
            result["cli_e2e"].update(compare_e2e.run(n_big=min(n, 20000), n_filter=n, n_small=min(n, 3000), threads=16, table=tab))
        except Exception as e:
            result["cli_e2e"]["compare_error"] = repr(e)

    if rank == 0:
        emit(result, args.detail)
    if table is not None:
        table.free()
    if comm is not None:
        comm.close()
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
