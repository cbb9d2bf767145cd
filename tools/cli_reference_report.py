#!/usr/bin/env python3
"""Replay tests/golden/cli/cases.json through mash_amd/bin/mash and report every case whose stdout
differs from the reference CLI's (first differing line shown) instead of stopping at the first."""
import json, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "tests", "golden", "cli")
MASH = os.path.join(ROOT, "mash_amd", "bin", "mash")
bad = 0
only_extra = "--extra" in sys.argv
for case in json.load(open(os.path.join(CLI, "cases.json"))):
    if only_extra and case.get("confirmed_on_gpu", True):
        continue
    d = tempfile.mkdtemp(prefix="clirep_")
    for f in os.listdir(os.path.join(CLI, "in")):
        shutil.copy(os.path.join(CLI, "in", f), d)
    msg = None
    for s in case["setup"]:
        r = subprocess.run([MASH, *s], cwd=d, capture_output=True)
        if r.returncode != 0:
            msg = "setup %s failed: %s" % (s, r.stderr[-200:])
            break
    if msg is None:
        r = subprocess.run([MASH, *case["cmd"]], cwd=d, capture_output=True)
        want = open(os.path.join(CLI, case["name"] + ".out"), "rb").read()
        if r.returncode != 0:
            msg = "exit %d: %s" % (r.returncode, r.stderr[-200:])
        elif r.stdout != want:
            a, b = r.stdout.splitlines(), want.splitlines()
            i = next((k for k in range(min(len(a), len(b))) if a[k] != b[k]), min(len(a), len(b)))
            msg = "line %d: got %r want %r (lines %d vs %d)" % (i, a[i][:120] if i < len(a) else None, b[i][:120] if i < len(b) else None, len(a), len(b))
    print(("DIFF " if msg else "same ") + case["name"] + (": " + msg if msg else ""), flush=True)
    bad += msg is not None
    shutil.rmtree(d)
print("differing cases:", bad)
