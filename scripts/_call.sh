cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
(timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bicubic" > gpurun_out/r03_t7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_t7.log; tail -5 gpurun_out/r03_t7.log)
timeout -k 10 200 python scripts/sweep.py --method bicubic --bicubic-fast --rounds 5 "" "STAGE2_ABLATE=3" "STAGE2_USE_ALT=1" > gpurun_out/r03_bicubic_fast_pk.log 2>&1 || exit 1
timeout -k 10 200 python scripts/sweep.py --method bicubic --bicubic-fast --nz 25 --resident 200 --rounds 6 "" "STAGE2_USE_ALT=1" > gpurun_out/r03_bicubic_fast_pk_short.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --method bicubic --bicubic-fast --no-extras --cpu-seconds 0 > gpurun_out/r03_bench_bicubic_fast.json 2> gpurun_out/r03_bench_bicubic_fast.err || { tail -5 gpurun_out/r03_bench_bicubic_fast.err; exit 1; }
tail -n 3 gpurun_out/r03_bicubic_fast_pk.log; tail -n 2 gpurun_out/r03_bicubic_fast_pk_short.log; python -c "
import json; d=json.loads(open('gpurun_out/r03_bench_bicubic_fast.json').read()); print(d['ms_per_step'], d['roofline']['frac'], d['config']['tuned_shape'], d['verified_all_slices_vs_gather'], d['config']['output_placement']['ms_at_each'])"
