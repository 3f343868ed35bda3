#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the staged kernel on the synthetic aligned / misaligned plans of scripts/bench_aligned.py
set -e
export TMPDIR=/tmp
out=gpurun_out/prof_aligned
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 scripts/bench_aligned.py > $out/run_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $out/l2 -- python3 scripts/bench_aligned.py > $out/run_l2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for tag in ("fetch", "l2"):
    f = glob.glob("gpurun_out/prof_aligned/%s/**/*_counter_collection.csv" % tag, recursive=True)[0]
    per = collections.defaultdict(list)
    for x in csv.DictReader(open(f)):
        if "staged_apply" in x["Kernel_Name"]:
            per[x["Counter_Name"]].append(float(x["Counter_Value"]))
    for k, v in per.items():
        # 13 launches per plan (3 warm + 10 timed), 4 plans in order
        groups = [v[i * 13:(i + 1) * 13] for i in range(4)]
        print(k, [round(sum(g) / max(len(g), 1) / 1e6, 3) for g in groups], "(M units per launch; plans: aligned, misaligned, 1.73 unrotated, C2)")
PY
