cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
(timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -m gpu -k "stored or typed or fill or busy" > gpurun_out/r03_t4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_t4.log; tail -6 gpurun_out/r03_t4.log)
timeout -k 10 300 python scripts/sweep_typed.py --rounds 6 "" "TYPED_STAGED2=0" "STAGE2T_ZPB=25" > gpurun_out/r03_sweep_typed3.log 2>&1 || exit 1
timeout -k 10 300 python scripts/sweep_typed.py --rounds 4 --nz 25 "" "TYPED_STAGED2=0" > gpurun_out/r03_sweep_typed_short.log 2>&1 || exit 1
for z in 8 25 100; do FIMEX_AMD_FWD_ZPB=$z timeout -k 10 120 python scripts/run_case.py forward_mean_c4 --tuning-build 2>/dev/null | tail -c 330; echo " ZPB=$z"; done > gpurun_out/r03_forward_zpb.log 2>&1
timeout -k 10 900 python scripts/collect_profiles.py r03 bilinear_nz200_default > gpurun_out/r03_collect_default.log 2>&1 || { tail -30 gpurun_out/r03_collect_default.log; exit 1; }
cat gpurun_out/r03_sweep_typed3.log gpurun_out/r03_sweep_typed_short.log gpurun_out/r03_forward_zpb.log; tail -50 gpurun_out/r03_collect_default.log
