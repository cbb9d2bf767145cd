cd $GRAFT_REPO_ROOT
timeout 900 python tools/cli_e2e.py --genomes 12000 --len 50000 > gpurun_out/o_cli_e2e.json 2> gpurun_out/o_cli_e2e.err
cat gpurun_out/o_cli_e2e.json; tail -c 400 gpurun_out/o_cli_e2e.err
MASH_AMD_NO_STREAM=1 timeout 600 python tools/cli_e2e.py --genomes 12000 --len 50000 --only-sketch > gpurun_out/o_cli_e2e_nostream.json 2>/dev/null
cat gpurun_out/o_cli_e2e_nostream.json
timeout 300 python tools/query_latency.py > gpurun_out/o_query_latency.json 2> gpurun_out/o_query_latency.err; tail -c 1500 gpurun_out/o_query_latency.json
