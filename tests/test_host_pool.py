"""CPU test of the parse pool of the CLI (mash_amd/host/parse_pool.h): tests/host_pool_test.cpp compiled with g++ and run
on files it writes itself -- order, equality with the sequential parse, errors in input order, look-ahead limits, runs
of ready files, jobs dealt to the workers.  No GPU, no libmashgpu."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_parse_pool_hands_files_over_in_order(tmp_path):
    exe = str(tmp_path / "host_pool_test")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-o", exe, os.path.join(HERE, "host_pool_test.cpp"),
                    os.path.join(ROOT, "mash_amd", "host", "fastx.cpp"), "-lz", "-lpthread"], check=True)
    scratch = tmp_path / "files"
    scratch.mkdir()
    r = subprocess.run([exe, str(scratch)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.startswith("OK 160 files")


def test_parse_pool_under_thread_sanitizer(tmp_path):
    """The same program built with -fsanitize=thread (and untimed condition waits: gcc 11's libtsan does not model
    pthread_cond_clockwait): no data race, no lock misuse reported."""
    exe = str(tmp_path / "host_pool_tsan")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-DHOSTPOOL_UNTIMED_WAIT", "-o", exe,
                        os.path.join(HERE, "host_pool_test.cpp"), os.path.join(ROOT, "mash_amd", "host", "fastx.cpp"), "-lz", "-lpthread"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        import pytest
        pytest.skip("no ThreadSanitizer runtime here: " + r.stderr[-200:])
    scratch = tmp_path / "files"
    scratch.mkdir()
    r = subprocess.run([exe, str(scratch)], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "OK 160 files" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
    assert "ThreadSanitizer" not in r.stderr, r.stderr[:3000]
