cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
(timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03_t5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_t5.log; tail -8 gpurun_out/r03_t5.log)
timeout -k 10 120 python scripts/run_case.py forward_median_c4 > gpurun_out/r03_forward_median2.json 2>&1 || exit 1
timeout -k 10 300 python scripts/bench_c5_chain.py > gpurun_out/r03_c5_chain.json 2> gpurun_out/r03_c5_chain.err || { tail -20 gpurun_out/r03_c5_chain.err; exit 1; }
tail -c 500 gpurun_out/r03_forward_median2.json; cat gpurun_out/r03_c5_chain.json
