"""The driver keeps only the tail of bench.py's stdout and parses its LAST line (round 3's 18 KB line was cut and
parsed as nothing): the headline object must stay small whatever the detail holds, round-trip through json and carry
the contract's keys."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _fat_result():
    note = "x" * 700
    phases = {k: {"avg_ms": 1.2345, "per_pass": 1.0, "ms_per_pass": 1.2345} for k in ("index", "discover", "fill", "merge")}
    rf = {"bound": "hbm", "achieved": 5952.1, "peak": 8000.0, "unit": "GB/s", "frac": 0.744, "traffic": 40040000000, "engine": "inverted index",
          "kernel": "mg::sp_fill_value_kernel", "kernel_ms": 6.72, "algorithmic_bytes_per_launch": 39999600000, "phases": phases,
          "pass": {"ms": 25.1, "traffic": 9.9e10, "compulsory_bytes": 4.08e10, "traffic_over_compulsory": 2.4, "output_write_bound_frac": 0.2},
          "step_frac": 0.2031, "index_ms": 7.9,
          "ports": {f"kernel{i}": {"valu": 0.6, "note": note} for i in range(12)}, "note": note}
    leg = {"value": 1.2345678e11, "unit": "pairs/s", "ms_per_step": 12.3, "warm_value": 2.2e11, "warm_ms_per_step": 5.0, "roofline": rf,
           "roofline_warm": rf, "cpu_baseline": {"value": 1.9e6, "unit": "pairs/s", "cores": 16, "kind": "reference", "sample": note},
           "config": {"workload": note}, "note": note}
    return {"metric": "pairwise Mash distances/sec (s=1000)", "value": 1.99e11, "unit": "pairs/s", "n_gpus": 1, "steps": 20, "warmup": 5,
            "ms_per_step": 25.1, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "mash triangle all-vs-all, 100000 clustered synthetic sketches, k=21 s=1000, 4999950000 pairs/step, per-table job "
                                   "(index build + discover + fill + merge each step), row-block sharded x1",
                       "parallelism": "rowblock1", "rank_row_blocks": list(range(9)), "output_checksum": [2122078313, 4999950000000],
                       "first_call_ms": 63.6, "rccl_ranks": 0, "table_broadcast_ms": 0.0, "note": note},
            "warm_value": 3.4e11, "warm_ms_per_step": 14.7, "roofline": rf, "roofline_warm": rf,
            "cpu_baseline": {"value": 1.86e6, "unit": "pairs/s", "cores": 256, "kind": "reference", "sample": note},
            "cpu_baseline_by_cores": {c: {"value": 1.0e5 * int(c), "unit": "pairs/s", "cores": int(c), "kind": "reference", "sample": note} for c in ("1", "16", "256")},
            "brackets": {k: dict(leg) for k in ("all_random", "all_identical", "clades_of_1000", "one_clade", "one_species")} | {"workload": note},
            "host_to_host": {"counts": leg, "pairs": leg, "full_c3_thresholded": leg, "full_c3_sparse": leg, "sample": note},
            "sketch": dict(leg, host_to_host=leg), "screen": dict(leg, mixed_database={"note": note}), "c5": leg,
            "cli_e2e": {"sketch": {"speedup_vs_reference": 3.2, "stages": {str(i): note for i in range(5)}},
                        "triangle": {"speedup_vs_reference": 12.0}, "host_cores": 16}}


def test_headline_line_is_small_and_complete(capsys, tmp_path):
    import bench
    res = _fat_result()
    assert len(json.dumps(res)) > 30000                    # the detail is as fat as round 3's line was
    bench.emit(res, str(tmp_path / "detail.json"))
    out = capsys.readouterr().out.rstrip().splitlines()
    assert sum(len(l) + 1 for l in out) < 8000             # everything printed fits the tail the driver keeps
    line = out[-1]
    assert len(line) < 4096
    h = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "warm_value"):
        assert k in h, k
    assert h["config"]["workload"].startswith("mash triangle") and h["config"]["output_checksum"] == [2122078313, 4999950000000]
    rf = h["roofline"]
    for k in ("bound", "kernel", "kernel_ms", "achieved", "peak", "unit", "frac", "traffic", "pass"):
        assert k in rf, k
    assert 0 < rf["frac"] <= 1 and rf["pass"]["phases_ms"]["index"] > 0
    # the WHOLE step against the same roof and what the per-table index costs of it: where the driver keeps them
    assert 0 < rf["step_frac"] < rf["frac"] and rf["index_ms"] > 0
    assert set(h["cpu_baseline"]["by_cores"]) == {"1", "16", "256"}                              # P = 1, 16, nproc
    assert h["h2h_full_pairs_s"] and h["h2h_thresholded_pairs_s"] and "step_frac" in h["c5_roofline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in h["cpu_baseline"], k
    assert h["sketch_bp_s"] and h["c5_pairs_s"] and h["screen_reads_s"] and h["cli_e2e_speedup"]["sketch"] == 3.2
    assert set(h["brackets_pairs_s"]) == {"all_random", "all_identical", "clades_of_1000", "one_clade", "one_species"}
    assert json.loads(open(tmp_path / "detail.json").read())["brackets"]["one_clade"]["note"]     # the detail kept everything


def test_cpu_baseline_shares_one_table_between_its_threads():
    """bench.py's cpu_baseline: the reference's table is built once per run and only read by the worker
    threads (round 5 rebuilt it in every worker: 256 threads measured their copies), the rows go out as many more blocks than
    threads, and the sample says so.  Rates cannot be asserted on a shared CI box; that two threads do the same pairs as one,
    through pyoracle.table_open / triangle_run, can."""
    import importlib.util
    import numpy as np
    from oracle import pyoracle
    from workloads import synth
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(b)
    finally:
        sys.argv = argv
    table, nh, ln = synth.clustered_sketches(700, 200, clusters=7, seed=2, pool=300, private=80)
    r1 = b.cpu_baseline_compare(table, nh, ln, 0.3, cores=1)
    r2 = b.cpu_baseline_compare(table, nh, ln, 0.3, cores=2)
    for r, c in ((r1, 1), (r2, 2)):
        assert r["cores"] == c and r["unit"] == "pairs/s" and r["value"] > 0 and "one shared table" in r["sample"]
    orc = pyoracle.Oracle(ref=pyoracle.ref_available())
    t = orc.table_open(table[:50], nh[:50], ln[:50])
    try:
        assert orc.triangle_run(t, 0, 50, 21, 4.0 ** 21) == 50 * 49 // 2
        assert orc.triangle_run(t, 10, 20, 21, 4.0 ** 21) == sum(range(10, 20))
    finally:
        orc.table_close(t)
