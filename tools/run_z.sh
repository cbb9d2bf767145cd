cd $GRAFT_REPO_ROOT
for seed in 2 3 4; do
timeout 600 python tests/fuzz_sketch.py --n 6000 --seed $seed --seconds 100 --dump gpurun_out/sfuzz > gpurun_out/z_sfuzz$seed.txt 2>&1; echo rc=$?
tail -8 gpurun_out/z_sfuzz$seed.txt | cut -c1-600
done
