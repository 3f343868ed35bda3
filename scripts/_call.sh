cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
(timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -m gpu -k "stored or typed" > gpurun_out/r03_t2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_t2.log; tail -15 gpurun_out/r03_t2.log)
timeout -k 10 300 python scripts/sweep_typed.py "" "TYPED_STAGED2=0" "STAGE2T_NT=1024" "STAGE2T_NT=256" "STAGE2T_LDS_KB=52" "STAGE2T_PAIR=0" "STAGE2T_ZPB=50" > gpurun_out/r03_sweep_typed.log 2>&1 || exit 1
timeout -k 10 300 python scripts/sweep_typed.py --method nearest "" "TYPED_STAGED2=0" "STAGE2T_NT=1024" > gpurun_out/r03_sweep_typed_nearest.log 2>&1 || exit 1
timeout -k 10 300 python scripts/sweep_typed.py --dtype uint8 "" "TYPED_STAGED2=0" > gpurun_out/r03_sweep_typed_u8.log 2>&1 || exit 1
timeout -k 10 200 python scripts/sweep.py "" "STAGE2_TW=256" "STAGE2_DEPTH=3" > gpurun_out/r03_sweep_bilinear_tw.log 2>&1 || exit 1
cat gpurun_out/r03_sweep_typed.log gpurun_out/r03_sweep_typed_nearest.log gpurun_out/r03_sweep_typed_u8.log; tail -n 4 gpurun_out/r03_sweep_bilinear_tw.log
