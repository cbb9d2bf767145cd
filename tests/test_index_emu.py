"""The index-build kernels of mash_amd/csrc/index_build.hip run on the CPU (tools/hipemu: work-items as fibers of one thread,
barriers and wave operations as context switches) and every array they produce is compared with a std::stable_sort statement
of the inverted index: values, rows, group starts and ends, code and position images, the statistics -- on tables with one
window and many, one sort pass to three, tiles of several pieces, ragged and empty rows, rows kept out of the index, genomes
of many sizes, values up to the top bit, values held by hundreds of rows, buckets beyond the LDS capacity (the two-level
sort: random values, and values held by 5 000 and 7 000 rows); a table in which two neighbouring values have thousands of
holders each must raise the flag (the caller then builds the index by the general sort).
The kernels' index arithmetic is thereby pinned without a GPU; under ThreadSanitizer (MASH_EMU_TSAN=1, minutes: work-items as
OS threads) a missing barrier is a reported data race."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "emu", "index_emu_main.cpp")
INC = ["-I" + os.path.join(ROOT, "tools", "hipemu"), "-I" + os.path.join(ROOT, "mash_amd", "csrc")]

FAST = ["random_small", "random_two_blocks", "clusters", "clusters_windows", "ragged", "ragged_windows", "copies_out", "top_bit", "one_row",
        "pieces", "clade", "big_random", "big_clade", "big_clade_fit", "twins", "big_twins"]


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("emu") / "index_emu")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-DMG_HIP_EMU", "-DHIPEMU_FIBERS", *INC, SRC, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.fixture(scope="module")
def fast_runs(emu):
    """every fast case as a process of its own, a few at a time (a case is one thread of context switches: seconds to half a minute)"""
    from concurrent.futures import ThreadPoolExecutor
    def run(case):
        return subprocess.run([emu, case], capture_output=True, text=True, timeout=900)
    with ThreadPoolExecutor(max_workers=max(1, min(8, (os.cpu_count() or 2) // 2))) as ex:
        return dict(zip(FAST, ex.map(run, FAST)))


@pytest.mark.parametrize("case", FAST)
def test_index_kernels_on_the_cpu(fast_runs, case):
    r = fast_runs[case]
    assert r.returncode == 0 and "all cases agree" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.skipif(not os.environ.get("MASH_EMU_SLOW"), reason="minutes of context switches: MASH_EMU_SLOW=1")
@pytest.mark.parametrize("case", ["sizes", "many_buckets", "three_passes"])
def test_index_kernels_on_the_cpu_slow(emu, case):
    r = subprocess.run([emu, case], capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0 and "all cases agree" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.skipif(not os.environ.get("MASH_EMU_SLOW"), reason="a quarter of an hour of context switches: MASH_EMU_SLOW=1")
def test_index_kernels_on_the_cpu_random_tables(emu):
    """`index_emu fuzz <seed> <cases>`: tables of random shape and kind with bucket widths from a twentieth to twenty times the
    plan's (full buckets, tiles in pieces, buckets beyond the LDS; windows of hundreds of buckets)"""
    r = subprocess.run([emu, "fuzz", "20250926", "45"], capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0 and "all cases agree" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.skipif(not os.environ.get("MASH_EMU_TSAN"), reason="ThreadSanitizer over 512 OS threads per workgroup: MASH_EMU_TSAN=1")
def test_index_kernels_under_thread_sanitizer(tmp_path):
    exe = str(tmp_path / "index_emu_tsan")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-DMG_HIP_EMU", "-pthread", *INC, SRC, "-o", exe], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no ThreadSanitizer runtime here: " + r.stderr[-200:])
    r = subprocess.run([exe, "ragged_windows"], capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, r.stdout[-2000:] + r.stderr[:3000]
