#!/usr/bin/env python3
"""Runs bench.py under rocprofv3 and writes the summaries the judge reads into gpurun_out/profiles_<tag>/ :
  <tag>_<name>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary of the bench command
  <tag>_<name>_pmc.json             per-launch HBM-side traffic of the apply kernel from separate --pmc passes
                                    (FETCH_SIZE doubled as calibrated with scripts/calib/calib.hip, WRITE_SIZE as is)
  <tag>_<name>_bench.json           the bench JSON line of the traced run
usage: python scripts/collect_profiles.py <tag> <name> [bench.py args...]
Copy the files into profiles/ (tracked) afterwards; profiles/hbm_traffic.json is assembled from the *_pmc.json files."""
import collections, csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def run(cmd, **kw):
    print("run:", " ".join(cmd[:7]), "...", flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, **kw)
    if r.returncode != 0:
        print(r.stdout[-3000:], r.stderr[-3000:]); sys.exit(1)
    return r

def main():
    tag, name, bargs = sys.argv[1], sys.argv[2], sys.argv[3:]
    out = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)
    os.makedirs(out, exist_ok=True)
    os.environ["TMPDIR"] = "/tmp"
    work = os.path.join("/tmp", "prof_%s_%s" % (tag, name))
    shutil.rmtree(work, ignore_errors=True)
    # --no-tune: the trace and the counters are those of ONE workgroup shape (the plan's default, or the second one with
    # --tuning-build and FIMEX_AMD_STAGE2_USE_ALT=1 in the environment), not a mix of the tuning launches
    bench = [sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-seconds", "0", "--no-extras", "--no-tune", "--placements", "1"] + bargs
    r = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", work + "/trace", "--"] + bench + ["--steps", "20", "--warmup", "3"])
    open(os.path.join(out, "%s_%s_bench.json" % (tag, name)), "w").write(r.stdout.strip().splitlines()[-1] + "\n")
    ks = glob.glob(work + "/trace/**/*_kernel_stats.csv", recursive=True)[0]
    shutil.copy(ks, os.path.join(out, "%s_%s_kernel_stats.csv" % (tag, name)))
    pmc = {}
    for grp in (["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_REQ_sum", "TCC_MISS_sum", "TCC_HIT_sum"], ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"]):
        d = work + "/pmc_" + grp[0]
        run(["rocprofv3", "--kernel-trace", "--pmc"] + grp + ["--output-format", "csv", "-d", d, "--"] + bench + ["--steps", "4", "--warmup", "1"])
        f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
        per = collections.defaultdict(list)
        for x in csv.DictReader(open(f)):
            if "_apply" in x["Kernel_Name"]:
                per[x["Counter_Name"]].append(float(x["Counter_Value"]))
                pmc["kernel"] = x["Kernel_Name"].split("(")[0]
        for k, v in per.items():
            pmc[k] = sum(v) / len(v)
            pmc[k + "_launches"] = len(v)
    pmc["fetch_bytes_per_launch"] = 2 * pmc["FETCH_SIZE"] * 1024  # gfx950: 128-B fabric reads tallied at 64 B
    pmc["write_bytes_per_launch"] = pmc["WRITE_SIZE"] * 1024
    pmc["hbm_bytes_per_launch"] = pmc["fetch_bytes_per_launch"] + pmc["write_bytes_per_launch"]
    pmc["note"] = ("FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE x2 per MI355X_MICROARCH.md (HBM section), confirmed here by "
                   "scripts/calib/calib.hip: 4 GiB read -> FETCH_SIZE 2097167 KiB in every access shape tried")
    json.dump(pmc, open(os.path.join(out, "%s_%s_pmc.json" % (tag, name)), "w"), indent=1)
    print(open(os.path.join(out, "%s_%s_kernel_stats.csv" % (tag, name))).read()[:1500])
    print(json.dumps(pmc, indent=1))

if __name__ == "__main__":
    main()
