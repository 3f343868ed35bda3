#!/usr/bin/env python3
"""Runs scripts/sweep.py under rocprofv3 --pmc (one pass per counter group) and prints, per variant,
kernel time and the L2 / fabric counters of the apply kernel.
usage: python scripts/sweep_counters.py [--nz N] [--method M] variant...   (variants as for sweep.py)
FETCH_SIZE is doubled (gfx950 tallies 128-B fabric reads at 64 B; calibrated with scripts/calib/calib.hip)."""
import argparse, collections, csv, glob, os, subprocess, sys, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_REQ_sum", "TCC_MISS_sum", "TCC_HIT_sum"]]

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nz", type=int, default=200)
    ap.add_argument("--method", default="bilinear")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sweepc"))
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    n = len(a.variants)
    res = collections.defaultdict(dict)
    os.environ["TMPDIR"] = "/tmp"
    for gi, grp in enumerate(GROUPS):
        d = os.path.join(a.out, "g%d" % gi)
        shutil.rmtree(d, ignore_errors=True)
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + grp + ["--output-format", "csv", "-d", d, "--",
               sys.executable, os.path.join(ROOT, "scripts", "sweep.py"), "--nz", str(a.nz), "--method", a.method,
               "--rounds", "1", "--reps", "1"] + a.variants
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stdout[-2000:], r.stderr[-2000:]); sys.exit(1)
        f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
        rows = [x for x in csv.DictReader(open(f)) if "_apply" in x["Kernel_Name"]]
        per = collections.defaultdict(list)
        for x in rows:
            per[x["Counter_Name"]].append((int(x["Dispatch_Id"]), float(x["Counter_Value"]),
                                           (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e6))
        for cname, lst in per.items():
            lst.sort()
            assert len(lst) == 2 * n, (cname, len(lst), n)
            for i, v in enumerate(a.variants):
                res[v][cname] = lst[n + i][1]
                res[v]["ms_" + cname] = lst[n + i][2]
    print("%-34s %8s %9s %9s %9s %9s %7s" % ("variant", "ms", "fetchGB", "writeGB", "L2req(M)", "L2miss(M)", "hit%"))
    for v in a.variants:
        r = res[v]
        print("%-34s %8.3f %9.2f %9.2f %9.1f %9.1f %7.1f" % (
            v or "(defaults)", r["ms_FETCH_SIZE"], 2 * r["FETCH_SIZE"] * 1024 / 1e9, r["WRITE_SIZE"] * 1024 / 1e9,
            r["TCC_REQ_sum"] / 1e6, r["TCC_MISS_sum"] / 1e6, 100 * r["TCC_HIT_sum"] / max(r["TCC_REQ_sum"], 1)), flush=True)

if __name__ == "__main__":
    main()
