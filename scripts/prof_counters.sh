#!/bin/bash
# Collects rocprofv3 evidence for one bench configuration into gpurun_out/prof/<tag>/ :
#   kernel trace + stats, then separate PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass).
# usage: scripts/prof_counters.sh <tag> [bench.py args...]
set -e
tag=$1; shift
out=gpurun_out/prof/$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-extras "$@" > $out/bench_trace.json 2> $out/bench_trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-extras "$@" > $out/bench_fetch.json 2> $out/bench_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-extras "$@" > $out/bench_write.json 2> $out/bench_write.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/pmc_l2 -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-extras "$@" > $out/bench_l2.json 2> $out/bench_l2.err
find $out -name "*.csv" | head -50
