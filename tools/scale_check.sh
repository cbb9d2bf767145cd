#!/bin/bash
# tools/scale_check.sh -- everything there is to run the minute a box with several MI355X shows up
# (no such box was available to rounds 1-6, so NO multi-GPU number in this repository is measured; what each rank
#  would pay for its block is measured on ONE GPU by tools/ranks_on_one_gpu.py).
#
#   bash tools/scale_check.sh [BENCH_JSON]      (from the repository root, extension built: python -c 'import __graft_entry__ as g; g.build()')
#
#   1. the sharded parity tests on DISTINCT devices (MASHGPU_TEST_DEVICES adds the real device list to the
#      tests' lists that repeat device 0): compare / screen / sketch / rect split, local communicator = real
#      ncclCommInitAll + peer broadcast;
#   2. the CLI on all devices against the single-device CLI, byte for byte (sketch, dist, triangle);
#   3. bench.py --gpus 1,2,4,8 (whatever divides the box) exactly as the driver launches it, then checks:
#        - the N=1 value is within 3 % of BENCH_JSON's (default: the newest BENCH_r*.json),
#        - config.rccl_ranks == N and config.comm == "libmashgpu/RCCL" on every N > 1 line,
#        - config.output_checksum identical for every N (the same 5e9 pairs, whoever computed them),
#      and prints value, speed-up over N=1 and table_broadcast_ms per N.
# Exit code 0 = all of it held.  Output: gpurun_out/scale_check/ (one JSON line per N in scale.jsonl).
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/scale_check
mkdir -p $OUT
NG=$(python - <<'EOF'
import torch
print(torch.cuda.device_count())
EOF
)
echo "GPUs visible: $NG"
if [ "$NG" -lt 2 ]; then
    echo "scale_check: needs >= 2 GPUs (found $NG); the single-GPU forms of these tests run in pytest -m gpu"
    exit 3
fi
DEVS=$(seq -s, 0 $((NG - 1)))
fail=0

echo "== 1. sharded parity tests on devices $DEVS"
MASHGPU_TEST_DEVICES=$DEVS python -m pytest tests -m gpu -q -x -k "sharded or rect_split or rank_communicator or dscreen" \
    > $OUT/tests.log 2>&1 || fail=1
tail -3 $OUT/tests.log

echo "== 2. CLI on all devices vs one device"
python - > $OUT/cli.log 2>&1 <<'EOF' || fail=1
import os, subprocess, tempfile, numpy as np
from workloads import synth
mash = os.path.abspath("mash_amd/bin/mash")
d = tempfile.mkdtemp()
rng = np.random.default_rng(3)
files = []
for i in range(96):
    p = os.path.join(d, f"g{i}.fa")
    open(p, "wb").write(b">s\n" + synth._rand_dna(rng, int(rng.integers(50_000, 400_000))) + b"\n")
    files.append(p)
def run(env, *a):
    return subprocess.run([mash, *a], cwd=d, capture_output=True, check=True, env=dict(os.environ, **env)).stdout
one, many = {"MASH_GPU_DEVICES": "0"}, {"MASH_GPU_DEVICES": "all"}
run(one, "sketch", "-o", "one", *files); run(many, "sketch", "-o", "many", *files)
assert open(os.path.join(d, "one.msh"), "rb").read() == open(os.path.join(d, "many.msh"), "rb").read(), "sketch differs"
for cmd in (("dist", "one.msh", "one.msh"), ("dist", "-d", "0.3", "one.msh", *files[:3]), ("triangle", "one.msh"), ("triangle", "-E", "one.msh")):
    assert run(one, *cmd) == run(many, *cmd), cmd
print("cli: sketch / dist / triangle identical on 1 and on all devices")
EOF
tail -2 $OUT/cli.log

echo "== 3. bench.py over N"
: > $OUT/scale.jsonl
for N in 1 2 4 8; do
    [ $N -gt $NG ] && break
    if [ $N -eq 1 ]; then
        python bench.py --gpus 1 --steps 3 --warmup 1 --no-sketch --no-screen --no-c5 --no-h2h --no-brackets --no-cpu > $OUT/n$N.json 2> $OUT/n$N.err || fail=1
    else
        python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
            bench.py --gpus $N --steps 3 --warmup 1 --no-sketch --no-screen --no-c5 --no-h2h --no-brackets --no-cpu > $OUT/n$N.json 2> $OUT/n$N.err || fail=1
    fi
    grep '^{' $OUT/n$N.json | tail -1 >> $OUT/scale.jsonl
done
REF=${1:-$(ls BENCH_r*.json 2>/dev/null | sort | tail -1)}
python - "$REF" $OUT/scale.jsonl <<'EOF' || fail=1
import json, sys
ref, lines = sys.argv[1], [json.loads(l) for l in open(sys.argv[2]) if l.strip()]
ok = True
base = lines[0]
want = None
try:
    r = json.load(open(ref))
    r = r.get("parsed", r)                       # the driver's BENCH_rNN.json wraps the line in "parsed"
    want = r.get("value")
except Exception as e:
    print("no reference bench line:", e)
if want:
    dev = abs(base["value"] / want - 1)
    print(f"N=1 {base['value']:.4g} {base['unit']} vs {ref} {want:.4g}: {dev * 100:.1f} % {'ok' if dev <= 0.03 else 'DEVIATES > 3 %'}")
    ok &= dev <= 0.03
for l in lines:
    c = l["config"]
    line_ok = l["n_gpus"] == 1 or (c.get("rccl_ranks") == l["n_gpus"] and c.get("comm") == "libmashgpu/RCCL")
    line_ok &= c.get("output_checksum") == base["config"].get("output_checksum")
    ok &= line_ok
    print(f"N={l['n_gpus']}: {l['value']:.4g} {l['unit']}  x{l['value'] / base['value']:.2f} over N=1  ms/step {l['ms_per_step']:.2f}  "
          f"broadcast {c.get('table_broadcast_ms')} ms  ranks {c.get('rccl_ranks')} ({c.get('comm')})  blocks {c.get('rank_row_blocks')}  "
          f"index ms by rank {c.get('index_ms_by_rank')}  checksum {c.get('output_checksum')}  "
          f"{'ok' if line_ok else 'FAILED'}")
sys.exit(0 if ok else 1)
EOF
[ $fail -eq 0 ] && echo "scale_check: all held" || echo "scale_check: FAILED (see $OUT/)"
exit $fail
