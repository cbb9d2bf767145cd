"""bench.py's roofline object on the CPU: compare_roofline() fed with the phase times a run reports (a stand-in for the
library's HIP-event records) and with the committed PMC files -- every shape the driver line can take (inverted-index
engine with and without PMC figures, a table of copies that runs no discover / merge, the tile engine) comes out with
the fields DESIGN.md section 5 describes and with the arithmetic it states."""
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


def test_inverted_index_engine_with_committed_pmc():
    pmc = bench.load_pmc("compare_c3_pmc.json", "mash_amd/csrc/compare_sparse.hip", "mash_amd/csrc/compare_merged.hip",
                         "mash_amd/csrc/compare_internal.h")
    assert pmc is not None, "profiles/compare_c3_pmc.json is stale: the kernel sources changed since the counters were read"
    eng = FakeEngine({"compare_fill": (6.72, 2), "compare_discover": (3.56, 2), "compare_merge": (4.44, 2)})
    r = bench.compare_roofline(eng, PAIRS, N, S, 2, pmc)
    assert r["bound"] == "hbm" and r["kernel"] == "mg::sp_fill_const_wave_kernel" and r["unit"] == "GB/s"
    assert r["algorithmic_bytes_per_launch"] == PAIRS * 8
    assert r["achieved"] == pytest.approx(PAIRS * 8 / 6.72e-3 / 1e9, rel=1e-3) and r["frac"] == pytest.approx(r["achieved"] / 8000.0, abs=1e-4)
    assert 0.5 < r["frac"] < 1.0
    assert r["pass_ms"] == pytest.approx(6.72 + 3.56 + 4.44, abs=1e-3)
    # the fill's PMC bytes are the bytes it must write; the pass moves 1.3-1.5 x the compulsory bytes
    assert r["traffic"] == pytest.approx(PAIRS * 8, rel=0.01)
    assert r["pass"]["compulsory_bytes"] == PAIRS * 8 + N * S * 8 + N * 12
    assert 1.2 < r["pass"]["traffic_over_compulsory"] < 1.6 and 0 < r["pass"]["measured_hbm_frac"] < 1
    assert r["pass"]["output_write_bound_frac"] == pytest.approx(PAIRS * 8 / (r["pass_ms"] * 1e-3) / 1e9 / 8000.0, abs=1e-3)
    assert r["survey_8d_no_reuse_model"]["bytes_per_pair"] == 2 * S * 8 + 8 and r["survey_8d_no_reuse_model"]["frac"] > 100
    ports = r["ports"]
    assert "mg::sp_merge_rows_kernel<false>" in ports and "mg::sp_discover_kernel<false, false>" in ports
    for k, v in ports.items():                    # each port on its own, none above 1; cold-only kernels (index build) left out
        assert all(0 <= v[p] <= 1.0 for p in ("valu", "salu", "vmem", "lds_inst", "lds_active")), (k, v)
        assert "sp_index_scatter" not in k
    assert r["pmc_source"] == "profiles/compare_c3_pmc.json"
    json.dumps(r)


def test_without_pmc_and_table_of_copies_and_tile_engine():
    r = bench.compare_roofline(FakeEngine({"compare_fill": (7.0, 3), "compare_discover": (2.4, 3), "compare_merge": (0.09, 3)}), PAIRS, N, S, 3, None)
    assert r["traffic"] is None and r["pass"]["traffic"] is None and r["pass"]["traffic_over_compulsory"] is None and "ports" not in r
    assert r["frac"] == pytest.approx(PAIRS * 8 / 7.0e-3 / 1e9 / 8000.0, abs=1e-3)
    # nothing but copies: the fill phase (fill + class pairs) is the whole pass
    r = bench.compare_roofline(FakeEngine({"compare_fill": (14.5, 2)}), PAIRS, N, S, 2, None)
    assert r["pass_ms"] == pytest.approx(14.5) and "class" in r["note"]
    # the tile engine: one kernel name, the mandated model as achieved / frac (it bounds nothing there either)
    r = bench.compare_roofline(FakeEngine({"compare": (80.4, 4)}), PAIRS, N, S, 2, None)
    assert r["kernel"] == "mg::compare_merged_kernel" and r["pass_ms"] == pytest.approx(160.8) and r["frac"] > 10
    assert r["algorithmic_bytes_per_pair"] == 2 * S * 8 + 8
    json.dumps(r)


def test_every_committed_pmc_file_is_current():
    """the five compare legs: profiles/compare_<leg>_pmc.json carries the hash of the kernel sources the default
    engine is built from today (bench.py would drop it otherwise, and the driver line would lose its PMC figures)"""
    for leg in ("c3", "c5", "random", "identical", "clades"):
        assert bench.load_pmc(f"compare_{leg}_pmc.json", "mash_amd/csrc/compare_sparse.hip", "mash_amd/csrc/compare_merged.hip",
                              "mash_amd/csrc/compare_internal.h") is not None, leg


def test_sketch_pmc_restamp_is_documented():
    """profiles/sketch_pmc_latest.json was read on the round-2 sketch.hip; round 3 changed the file inside the probe
    instantiation only, and the file was re-stamped after tools/isa_same.py showed the sketching kernel's instructions
    unchanged -- the JSON says so, the report is committed, and the screen file (whose kernel did change) stays stale."""
    d = json.load(open(os.path.join(ROOT, "profiles", "sketch_pmc_latest.json")))
    assert bench.load_pmc("sketch_pmc_latest.json", "mash_amd/csrc/sketch.hip", "mash_amd/csrc/kmer_hash.h") is not None
    assert "restamped" in d and "isa_same.py" in d["restamped"]["why"]
    rep = open(os.path.join(ROOT, "profiles", "r03_sketch_isa_check.txt")).read()
    assert "sketch_chunks_kernel<21, 0, 256, false>" in rep and "same multiset of instructions" in rep
    assert bench.load_pmc("screen_pmc_latest.json", "mash_amd/csrc/sketch.hip", "mash_amd/csrc/kmer_hash.h", "mash_amd/csrc/screen.hip") is None
