cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
timeout -k 10 300 python scripts/sweep_typed.py --rounds 6 "" "STAGE2T_DEPTH=3" "STAGE2T_DEPTH=3,STAGE2T_ZPB=50" "STAGE2T_DEPTH=3,STAGE2T_LDS_KB=52" "STAGE2T_ZPB=40" > gpurun_out/r03_sweep_typed2.log 2>&1 || exit 1
timeout -k 10 200 python scripts/sweep.py --rounds 8 "" "STAGE2_DEPTH=3" "STAGE2_DEPTH=3,STAGE2_USE_ALT=1" "STAGE2_USE_ALT=1" > gpurun_out/r03_sweep_depth.log 2>&1 || exit 1
timeout -k 10 200 python scripts/sweep.py --method nearest --rounds 6 "" "STAGE2_DEPTH=3" > gpurun_out/r03_sweep_depth_nearest.log 2>&1 || exit 1
timeout -k 10 200 python scripts/sweep.py --nz 25 --resident 200 --rounds 8 "" "STAGE2_DEPTH=3" "STAGE2_USE_ALT=1" > gpurun_out/r03_sweep_depth_short.log 2>&1 || exit 1
timeout -k 10 200 python scripts/sweep.py --nz 1 --resident 40 --rounds 8 "" "FEW=0" > gpurun_out/r03_sweep_few.log 2>&1 || exit 1
timeout -k 10 200 python scripts/sweep.py --nz 1 --resident 40 --rounds 8 --method nearest "" "FEW=0" > gpurun_out/r03_sweep_few_nearest.log 2>&1 || exit 1
(timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ragged or backward_methods or gather_and_lds" > gpurun_out/r03_t3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_t3.log; tail -4 gpurun_out/r03_t3.log)
cat gpurun_out/r03_sweep_typed2.log; tail -n 4 gpurun_out/r03_sweep_depth.log; tail -n 2 gpurun_out/r03_sweep_depth_nearest.log; tail -n 3 gpurun_out/r03_sweep_depth_short.log; tail -n 2 gpurun_out/r03_sweep_few.log gpurun_out/r03_sweep_few_nearest.log
