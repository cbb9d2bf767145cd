"""CLI parity against the REFERENCE CLI ITSELF.

tests/golden/cli/*.out are the stdout of `oracle/_ref/mash-ref` -- the reference's own, unmodified
sources built with three shim headers for Cap'n Proto and a binomial tail for GSL (oracle/Makefile,
target `refcli`; tests/golden/make_cli_golden.py wrote the fixtures in the build container, where
/root/reference is mounted).  Every case is replayed here, argument for argument, through
mash_amd/bin/mash on the GPU and must print the same bytes: sketching modes (-i, -n, -Z, -a, -z, -r,
-m, -c, -M, -S, -l, -p, gz), info / paste, dist, triangle and screen with their option variants."""
import json, os, shutil, subprocess
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CLI = os.path.join(HERE, "golden", "cli")
MASH = os.path.join(ROOT, "mash_amd", "bin", "mash")
CASES = json.load(open(os.path.join(CLI, "cases.json")))

# cases where this CLI is known to print something else than the reference (none may be added
# silently: each entry says what differs)
KNOWN_DIFFERENCES = {}


def _params():
    out = []
    for c in CASES:
        marks = []
        if not c.get("confirmed_on_gpu", True):
            # written when no GPU time was left to replay them: they report (XPASS / xfail) without
            # deciding the suite until a GPU run has confirmed them and the flag is flipped
            marks.append(pytest.mark.xfail(strict=False, reason="reference CLI fixture not yet replayed on a GPU"))
        out.append(pytest.param(c, id=c["name"], marks=marks))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("case", _params())
def test_cli_prints_what_the_reference_cli_prints(case, tmp_path):
    if case["name"] in KNOWN_DIFFERENCES:
        pytest.xfail(KNOWN_DIFFERENCES[case["name"]])
    assert os.path.exists(MASH), "mash_amd/bin/mash is not built"
    for f in os.listdir(os.path.join(CLI, "in")):
        shutil.copy(os.path.join(CLI, "in", f), tmp_path)
    limit = 300 if case.get("confirmed_on_gpu", True) else 45
    for s in case["setup"]:
        r = subprocess.run([MASH, *s], cwd=tmp_path, capture_output=True, timeout=limit)
        assert r.returncode == 0, (s, r.stderr[-300:])
    r = subprocess.run([MASH, *case["cmd"]], cwd=tmp_path, capture_output=True, timeout=limit)
    assert r.returncode == 0, (case["cmd"], r.stderr[-300:])
    want = open(os.path.join(CLI, case["name"] + ".out"), "rb").read()
    assert r.stdout == want, case["name"]
