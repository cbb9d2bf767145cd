"""The join engine of mash_amd/csrc/compare_join.hip -- the list kernels (jn_emit / jn_heads / jn_groups / jn_gend / jn_levels;
the device sort and scan replaced by std::stable_sort and std::partial_sum), the count of shared hashes and jn_tile_kernel -- run
on the CPU (tools/hipemu: work-items as fibers) on a code image made by a std::stable_sort, and EVERY pair is compared with
the loop of compareSketches (CommandDistance.cpp:347-385): a tree of descent over three blocks of rows (with and without the
early stop), row ranges that cut through blocks, an index on a permuted table, rows of every length over one pool (positions
that differ between the rows of a pair), copies of rows (kept out of the index), near-copies (holder lists of 64 on both
sides), unrelated rows, and rect jobs (queries located in the table's sorted values)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "emu", "join_emu_main.cpp")
INC = ["-I" + os.path.join(ROOT, "tools", "hipemu"), "-I" + os.path.join(ROOT, "mash_amd", "csrc")]
CASES = ["species", "ranges", "ragged", "copies", "near", "random", "rect"]


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("emu") / "join_emu")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-DMG_HIP_EMU", "-DHIPEMU_FIBERS", *INC, SRC, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.mark.parametrize("case", CASES)
def test_join_kernels_on_the_cpu(emu, case):
    r = subprocess.run([emu, case], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "all cases agree" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_join_kernels_on_the_cpu_random_tables(emu):
    """tables of random shape (2 .. 200 rows, s 1 .. 120, trees / pools / pools with copies, ragged rows, row ranges, permuted
    indexes, with and without the early stop): `join_emu fuzz <seed> <cases>`; 160 cases of four other seeds ran clean when
    this was written"""
    r = subprocess.run([emu, "fuzz", "20250930", "25"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "all cases agree" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
