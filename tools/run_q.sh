cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, time, json
sys.path.insert(0, '.')
import torch
from mash_amd import abi, synth_torch
torch.cuda.init(); dev = torch.device("cuda", 0)
eng = abi.MashGpu(0, stream=torch.cuda.current_stream().cuda_stream)
for n in (3000, 40000):
    h, nh, ln = synth_torch.clustered_sketch_table(n, 1000, clusters=max(1, n // 100), device=dev)
    torch.cuda.synchronize()
    t = eng.table_wrap(h.data_ptr(), nh.data_ptr(), ln.data_ptr(), n, 1000, keep=(h, nh, ln))
    pairs = n * (n - 1) // 2
    out = torch.empty((pairs, 2), dtype=torch.int32, device=dev)
    for e in ("windows", "direct"):
        os.environ.pop("MASHGPU_COMPARE_KERNEL", None); os.environ["MASHGPU_COMPARE_WINDOWS"] = "1"
        if e == "direct": os.environ["MASHGPU_COMPARE_KERNEL"] = "direct"
        eng.compare_tri_dev(t, 0, n, out.data_ptr()); torch.cuda.synchronize()
        eng.prof_enable(True); eng.prof_reset()
        t0 = time.perf_counter(); eng.compare_tri_dev(t, 0, n, out.data_ptr()); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(n, e, "%.2f ms" % (dt * 1e3), eng.prof_avg_ms("compare"), flush=True)
        eng.prof_enable(False)
    t.free()
PY
cd /tmp && export TMPDIR=/tmp
MASHGPU_COMPARE_DIRECT=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/q_trace -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-sketch --no-screen --no-cpu --no-h2h --no-c5 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for p in glob.glob("gpurun_out/q_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "compare_direct" in r["Kernel_Name"]:
            print(r["Kernel_Name"][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, "ms", r.get("Grid_Size"), r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Scratch_Size"))
PY
