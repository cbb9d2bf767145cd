"""bench.py's roofline object on the CPU: compare_roofline() fed with the phase times a run reports (a stand-in for the
library's HIP-event records) and with PMC figures -- every shape the line can take (inverted-index engine cold and
warm, with and without PMC figures, a table of copies that runs no discover / merge, the tile engine) comes out with
the fields DESIGN.md section 5 describes and with the arithmetic it states; the compact form of the headline keeps
the contract's fields."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class FakeEngine:
    def __init__(self, phases):
        self.phases = phases                      # name -> (avg ms, launches)

    def prof_avg_ms(self, name):
        return self.phases.get(name, (0.0, 0))


N, S = 100_000, 1000
PAIRS = N * (N - 1) // 2
SRCS = ("mash_amd/csrc/compare_sparse.hip", "mash_amd/csrc/compare_dense.hip", "mash_amd/csrc/compare_merged.hip", "mash_amd/csrc/compare_internal.h")


def _pmc():
    port = {"valu": 0.6, "salu": 0.1, "vmem": 0.02, "lds_inst": 0.1, "lds_active": 0.3}
    return {"hbm_bytes_per_pass": 5.7e10, "source": "profiles/x.json",
            "kernels": {"mg::sp_fill_value_kernel": {"hbm_read_bytes_per_pass": 1e6, "hbm_write_bytes_per_pass": PAIRS * 8 + 4e7, "ports": port, "ms_per_pass": 6.7},
                        "void mg::sp_discover_kernel<false, false>(mg::SparseArgs)": {"hbm_read_bytes_per_pass": 1e10, "hbm_write_bytes_per_pass": 1e8, "ports": port}}}


def test_inverted_index_engine_cold_pass_with_pmc():
    eng = FakeEngine({"compare_index": (9.5, 2), "compare_fill": (6.72, 2), "compare_discover": (3.56, 2), "compare_merge": (3.93, 2)})
    r = bench.compare_roofline(eng, PAIRS, N, S, 2, _pmc())
    assert r["bound"] == "hbm" and r["kernel"] == "mg::sp_fill_value_kernel" and r["unit"] == "GB/s" and r["engine"] == "inverted index"
    assert r["algorithmic_bytes_per_launch"] == PAIRS * 8
    assert r["achieved"] == pytest.approx(PAIRS * 8 / 6.72e-3 / 1e9, rel=1e-3) and r["frac"] == pytest.approx(r["achieved"] / 8000.0, abs=1e-4)
    assert 0.5 < r["frac"] < 1.0 and r["kernel_ms"] == 6.72
    assert r["pass"]["ms"] == pytest.approx(9.5 + 6.72 + 3.56 + 3.93, abs=1e-3)
    assert set(r["phases"]) == {"index", "fill", "discover", "merge"} and r["phases"]["index"]["ms_per_pass"] == 9.5
    assert r["traffic"] == pytest.approx(PAIRS * 8, rel=0.01)
    assert r["pass"]["compulsory_bytes"] == PAIRS * 8 + N * S * 8 + N * 12
    assert 1.2 < r["pass"]["traffic_over_compulsory"] < 1.6
    assert r["pass"]["output_write_bound_frac"] == pytest.approx(PAIRS * 8 / (r["pass"]["ms"] * 1e-3) / 1e9 / 8000.0, abs=1e-3)
    assert r["survey_8d_no_reuse_model_gbs"] > 100 * 8000
    for k, v in r["ports"].items():               # each port on its own, none above 1
        assert all(0 <= v[p] <= 1.0 for p in ("valu", "salu", "vmem", "lds_inst", "lds_active")), (k, v)
    c = bench.compact_roofline(r)
    assert set(c) == {"bound", "kernel", "kernel_ms", "achieved", "peak", "unit", "frac", "traffic", "pass", "step_frac", "index_ms"}
    assert c["pass"]["phases_ms"] == {"index": 9.5, "discover": 3.56, "fill": 6.72, "merge": 3.93} and c["frac"] <= 1
    assert len(json.dumps(c)) < 600


def test_per_table_step_with_the_fill_beside_the_index_build():
    """A per-table step writes its constant on a stream of its own while the index is built (compare_fill_aside) and ends what
    is left behind the build (compare_fill): the aside launch lies INSIDE the index phase -- the pass is the sum of the others --
    and the dominant kernel's 8 B per pair are priced over both launches' durations."""
    eng = FakeEngine({"compare_index": (10.1, 2), "compare_fill_aside": (9.8, 2), "compare_fill": (0.45, 2), "compare_discover": (0.36, 2),
                      "compare_dense": (0.24, 2), "compare_merge": (0.14, 2)})
    pm = _pmc()
    pm["kernels"]["mg::sp_fill_chunks_kernel"] = {"hbm_read_bytes_per_pass": 2e6, "hbm_write_bytes_per_pass": PAIRS * 8 + 1e7, "ports": {}, "ms_per_pass": 10.25}
    r = bench.compare_roofline(eng, PAIRS, N, S, 2, pm)
    assert r["kernel"] == "mg::sp_fill_chunks_kernel" and r["kernel_ms"] == pytest.approx(10.25) and "index build" in r["beside"]
    assert r["achieved"] == pytest.approx(PAIRS * 8 / 10.25e-3 / 1e9, rel=1e-3) and r["frac"] == pytest.approx(r["achieved"] / 8000.0, abs=1e-4)
    assert r["pass"]["ms"] == pytest.approx(10.1 + 0.45 + 0.36 + 0.24 + 0.14, abs=1e-3)
    assert r["traffic"] == pytest.approx(PAIRS * 8, rel=0.01)
    c = bench.compact_roofline(r)
    assert c["beside"] == r["beside"] and c["pass"]["phases_ms"]["fill_aside"] == 9.8 and len(json.dumps(c)) < 900
    assert c["kernel_launches_per_pass"] == 2 and c["kernel_avg_launch_ms"] == pytest.approx(10.25 / 2)      # (what rocprofv3 --stats averages)
    # the same kernel by itself (bench.py takes it from the further passes it times next): it travels in the headline
    r["alone"] = {"kernel": "mg::sp_fill_value_kernel", "kernel_ms": 6.65, "achieved": 6015.0, "frac": 0.7519}
    assert bench.compact_roofline(r)["alone"]["frac"] == 0.7519 and len(json.dumps(bench.compact_roofline(r))) < 1100


def test_without_pmc_and_table_of_copies_and_tile_engine():
    r = bench.compare_roofline(FakeEngine({"compare_fill": (7.0, 3), "compare_discover": (2.4, 3), "compare_merge": (0.09, 3)}), PAIRS, N, S, 3, None)
    assert r["traffic"] is None and r["pass"]["traffic"] is None and r["pass"]["traffic_over_compulsory"] is None and "ports" not in r
    assert r["frac"] == pytest.approx(PAIRS * 8 / 7.0e-3 / 1e9 / 8000.0, abs=1e-3) and "index" not in r["phases"]
    # nothing but copies: the fill is the whole pass
    r = bench.compare_roofline(FakeEngine({"compare_fill": (6.7, 2)}), PAIRS, N, S, 2, None)
    assert r["pass"]["ms"] == pytest.approx(6.7) and set(r["phases"]) == {"fill"}
    # the tile engine: one kernel name, the mandated model as achieved / frac (it bounds nothing there: DESIGN 4.1b)
    r = bench.compare_roofline(FakeEngine({"compare": (80.4, 4)}), PAIRS, N, S, 2, None)
    assert r["kernel"] == "mg::compare_merged_kernel" and r["pass"]["ms"] == pytest.approx(160.8) and r["frac"] > 10 and r["engine"] == "tiles"
    json.dumps(r)


def test_committed_pmc_files_are_current_or_dropped():
    """profiles/compare_<leg>_pmc.json carries the hash of the kernel sources it was read on; bench.py drops a file
    whose hash differs from the running sources (the line then says traffic: null) -- never a stale figure."""
    for leg in ("c3", "c3_cold", "c5", "random", "identical", "clades"):
        p = os.path.join(ROOT, "profiles", f"compare_{leg}_pmc.json")
        d = bench.load_pmc(f"compare_{leg}_pmc.json", *SRCS)
        if d is None:
            continue                                       # absent or stale: dropped
        assert json.load(open(p))["kernel_src_sha"] == bench.src_sha(*SRCS)
        assert d["hbm_bytes_per_pass"] > 0 and d["kernels"]
