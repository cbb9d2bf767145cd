"""The dense-group kernels of mash_amd/csrc/compare_dense.hip -- dn_encode_kernel (both forms: an entry and eight entries
per work-item) and dn_pairs_kernel (tiles of 8 and 32 rows, 4 / 8 / 16 rows side by side, extras counted from the bit
planes and from the rows' lists) -- run on the CPU (tools/hipemu: work-items as fibers) on an index made by a
std::stable_sort, and EVERY pair inside every group is compared with the loop of compareSketches
(CommandDistance.cpp:347-385): near-copies, loose clusters with dozens of extras per word, 5 / 12 / 20 extras in ONE gap of
the word in which s is reached (bit planes 0+2, 2+3, the flagged word that falls back to the lists), rows shorter than s,
a universe three times s over two blocks of rows, universes of one word.  The run ends that the encode clips are checked
too, and the two encode kernels must produce the same bytes.  (Mutations of the kernels -- a plane's weight, the flag's
threshold, the word counter, the order of the extras' list -- each fail at least one case.)"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "emu", "dense_emu_main.cpp")
INC = ["-I" + os.path.join(ROOT, "tools", "hipemu"), "-I" + os.path.join(ROOT, "mash_amd", "csrc")]
CASES = ["near", "loose", "clumped", "clumped12", "clumped5", "short", "wide", "one_word"]


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("emu") / "dense_emu")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-DMG_HIP_EMU", "-DHIPEMU_FIBERS", *INC, SRC, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.mark.parametrize("case", CASES)
def test_dense_kernels_on_the_cpu(emu, case):
    r = subprocess.run([emu, case], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "all cases agree" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_dense_kernels_on_the_cpu_random_groups(emu):
    """groups of random shape (2 .. 150 rows, pools of 1 .. 3.5 s, 30 .. 98 % kept, up to 25 extras in one gap, short rows):
    `dense_emu fuzz <seed> <cases>`; 900 cases of six other seeds ran clean when this was written"""
    r = subprocess.run([emu, "fuzz", "20250926", "40"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "all cases agree" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]

