"""The reference CLI compiled against the boundary (SURVEY.md 8b last row).

oracle/_ref/mash-ref-gpu (make -C oracle refcli-gpu) = the reference's 21 translation units, unmodified,
with Sketch::initFromFiles (Sketch.cpp:105-253) and the two `compare` workers (CommandDistance.cpp:306-334,
CommandTriangle.cpp:200-214) bound to libmashgpu.so by oracle/gpu_boundary.cpp.  It travels to the GPU box
with the snapshot (oracle/_ref/ is git-ignored, not gpurun-ignored).  Here it must

  * pass the reference's own three `make test` recipes (Makefile.in:94-115) against test/ref/*, and
  * print, byte for byte, the committed outputs of the UNBOUND reference CLI (tests/golden/cli/*.out)
    for thirteen fixtures that go through the replaced functions: sketching plain / -n / -M / gz / -p /
    k = 31 s = 10 000 inputs, `dist` and `triangle` from files and sketches, tables, both filters.

A failure here means the C ABI cannot stand in for the functions it claims to replace."""
import gzip, json, os, shutil, subprocess
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CLI = os.path.join(HERE, "golden", "cli")
GOLD = os.path.join(HERE, "golden")
BOUND = os.path.join(ROOT, "oracle", "_ref", "mash-ref-gpu")
MASH = os.path.join(ROOT, "mash_amd", "bin", "mash")
CASES = {c["name"]: c for c in json.load(open(os.path.join(CLI, "cases.json")))}

THROUGH_THE_BOUNDARY = [
    "sketch3_dump", "sketch_noncanonical_k31", "sketch_counts", "sketch_gz", "sketch_threads",
    "dist_files", "dist_sketch_vs_file", "dist_table", "dist_maxd", "dist_maxp",
    "triangle_files", "triangle_edge_maxd", "c5_triangle_k31_s10000",
]

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bound():
    if not os.path.exists(BOUND):
        pytest.skip("oracle/_ref/mash-ref-gpu not built (make -C oracle refcli-gpu, where /root/reference is mounted)")
    return BOUND


@pytest.mark.parametrize("name", THROUGH_THE_BOUNDARY)
def test_bound_reference_cli_prints_what_the_reference_cli_prints(bound, name, tmp_path):
    case = CASES[name]
    for f in os.listdir(os.path.join(CLI, "in")):
        shutil.copy(os.path.join(CLI, "in", f), tmp_path)
    for s in case["setup"]:
        r = subprocess.run([bound, *s], cwd=tmp_path, capture_output=True, timeout=300)
        assert r.returncode == 0, (s, r.stderr[-300:])
    r = subprocess.run([bound, *case["cmd"]], cwd=tmp_path, capture_output=True, timeout=300)
    assert r.returncode == 0, (case["cmd"], r.stderr[-300:])
    assert r.stdout == open(os.path.join(CLI, name + ".out"), "rb").read(), name


def test_bound_reference_cli_passes_the_reference_make_test(bound, tmp_path):
    """testSketch / testDist / testScreen of the reference's Makefile.in:94-115 (the genome FASTA files are
    not in the reference tree: the golden sketches stand in for them, written by this repository's CLI)."""
    for n in ("reads1.fastq", "reads2.fastq"):
        with gzip.open(os.path.join(GOLD, n + ".gz"), "rb") as fi, open(tmp_path / n, "wb") as fo:
            shutil.copyfileobj(fi, fo)
    run = lambda *a: subprocess.run([*a], cwd=tmp_path, capture_output=True, check=True).stdout
    run(bound, "sketch", "-r", "-I", "reads", "reads1.fastq", "reads2.fastq", "-o", "reads.msh")
    dump = run(bound, "info", "-d", "reads.msh").decode()
    a, b = dump.index('\t\t\t"counts" :'), dump.index("\t\t}\n\t]")          # the golden predates the counts block of -r
    assert dump[:a] + dump[b:] == open(os.path.join(GOLD, "reads.json")).read()
    run(MASH, "json2msh", os.path.join(GOLD, "genomes.json"), "genomes.msh")
    assert run(bound, "info", "-d", "genomes.msh") == open(os.path.join(GOLD, "genomes.json"), "rb").read()
    assert run(bound, "dist", "genomes.msh", "reads.msh") == open(os.path.join(GOLD, "genomes.dist"), "rb").read()
    assert run(bound, "screen", "genomes.msh", "reads1.fastq", "reads2.fastq") == open(os.path.join(GOLD, "screen"), "rb").read()


def test_bound_reference_cli_really_calls_the_library(bound, tmp_path):
    """Without a usable device the bound functions fail loudly (they have no CPU path), while what was not
    replaced keeps working: the binding is in the call path, not beside it."""
    for f in ("g1.fa", "g3.fa"):
        shutil.copy(os.path.join(CLI, "in", f), tmp_path)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    r = subprocess.run([bound, "sketch", "-o", "x", "g1.fa", "g3.fa"], cwd=tmp_path, capture_output=True, env=env)
    assert r.returncode != 0 and b"no usable GPU" in r.stderr
    ok = subprocess.run([bound, "sketch", "-o", "x", "g1.fa", "g3.fa"], cwd=tmp_path, capture_output=True)
    assert ok.returncode == 0
    r = subprocess.run([bound, "dist", "x.msh", "x.msh"], cwd=tmp_path, capture_output=True, env=env)
    assert r.returncode != 0 and b"no usable GPU" in r.stderr
    info = subprocess.run([bound, "info", "-t", "x.msh"], cwd=tmp_path, capture_output=True, env=env)
    assert info.returncode == 0 and info.stdout.count(b"\n") == 3
