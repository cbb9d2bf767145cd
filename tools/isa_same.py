#!/usr/bin/env python3
"""Did a kernel's machine code change between two revisions of its source?

    python tools/isa_same.py REV_OLD REV_NEW "void mg::sketch_chunks_kernel<21, 0, 256, false>(mg::SketchArgs)" [--part 2]

Compiles mash_amd/csrc/sketch.hip (with the headers of the same revision) for gfx950 to assembly at both revisions and
compares the named kernel: identical text; or the same MULTISET of instructions once kernel-argument offsets of scalar
loads are masked (a struct that grew moves them) -- i.e. the scheduler ordered independent instructions differently,
nothing else; or different.  Used to decide whether PMC counters read on the old revision still describe the kernel
(profiles/sketch_pmc_latest.json was re-stamped on that ground; the probe instantiation, which did change, was not)."""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDRS = ["kmer_hash.h", "kmer_stream.h", "sketch_internal.h", "screen_internal.h"]


def asm(rev, part, d):
    os.makedirs(d, exist_ok=True)
    for f in ["sketch.hip"] + HDRS:
        data = subprocess.run(["git", "show", f"{rev}:mash_amd/csrc/{f}"], capture_output=True, cwd=ROOT, check=True).stdout
        open(os.path.join(d, f), "wb").write(data)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", f"-DSK_PART={part}", "-x", "hip", "-S",
                    "--cuda-device-only", "-o", "out.s", "sketch.hip"], cwd=d, check=True, capture_output=True)
    return open(os.path.join(d, "out.s")).read()


def body(txt, want):
    for m in re.finditer(r"^(_ZN2mg\w+):\s*;[^\n]*\n(.*?)\n\s*s_endpgm", txt, re.S | re.M):
        if subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip() == want:
            lines = [l.split(";")[0].strip() for l in m.group(2).split("\n")]
            return [l for l in lines if l and not l.startswith(".")]
    raise SystemExit(f"kernel not found: {want}")


def main():
    old, new, kernel = sys.argv[1:4]
    part = int(sys.argv[sys.argv.index("--part") + 1]) if "--part" in sys.argv else 2
    with tempfile.TemporaryDirectory() as t:
        a, b = body(asm(old, part, os.path.join(t, "a")), kernel), body(asm(new, part, os.path.join(t, "b")), kernel)
    mask = lambda l: re.sub(r"(s_load_dword\w*\s+s\[?[\d:]+\]?,\s*s\[0:1\],\s*)0x[0-9a-f]+", r"\1KERNARG", l)
    print(f"{kernel}: {len(a)} instructions at {old}, {len(b)} at {new}")
    if a == b:
        print("verdict: identical")
    elif collections.Counter(map(mask, a)) == collections.Counter(map(mask, b)):
        moved = sum(1 for x, y in zip(a, b) if x != y)
        print(f"verdict: same multiset of instructions (kernel-argument offsets masked); {moved} lines sit at other positions "
              "(independent instructions in another order)")
    else:
        only_a = collections.Counter(map(mask, a)) - collections.Counter(map(mask, b))
        only_b = collections.Counter(map(mask, b)) - collections.Counter(map(mask, a))
        print(f"verdict: DIFFERENT ({sum(only_a.values())} instructions only in the old, {sum(only_b.values())} only in the new)")


if __name__ == "__main__":
    main()
