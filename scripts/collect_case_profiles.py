#!/usr/bin/env python3
"""rocprofv3 evidence for the secondary kernels: scripts/run_case.py behind `rocprofv3 ... -- python3 scripts/run_case.py <case>`,
once with --kernel-trace --stats, once each with --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, MI355X_MICROARCH.md, HBM
section).  Writes gpurun_out/profiles_<tag>/<tag>_<case>_{kernel_stats.csv,pmc.json,run.json}; copy them into profiles/.
usage: python scripts/collect_case_profiles.py <tag> <case> [<case> ...]"""
import collections, csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


ABNORMAL = []


def run(cmd, note_file=None):
    print("run:", " ".join(cmd[:8]), "...", flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", timeout=280)
    if r.returncode != 0:
        # A profiled run that ends abnormally AFTER its result line (seen with the cooperative launches of the small-batch fills)
        # is not passed over: its status and the end of its stderr are written next to the profile, listed at the end of this
        # script's output, and the script itself exits non-zero.  The profile files are kept for reading.
        if '"case"' in r.stdout and note_file:
            with open(note_file, "a") as f:
                f.write("command: %s\nstatus: %d\n--- stdout (end)\n%s\n--- stderr (end)\n%s\n\n" % (" ".join(cmd), r.returncode, r.stdout[-1500:], r.stderr[-6000:]))
            ABNORMAL.append((cmd[3] if len(cmd) > 3 else "", r.returncode, note_file))
            print("ABNORMAL EXIT: status %d after the result line; see %s" % (r.returncode, note_file), flush=True)
            return r
        print(r.stdout[-2000:], r.stderr[-2000:], flush=True)
        sys.exit(1)
    return r


def main():
    tag, cases = sys.argv[1], sys.argv[2:]
    out = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)
    os.makedirs(out, exist_ok=True)
    os.environ["TMPDIR"] = "/tmp"
    prog = [sys.executable, os.path.join(ROOT, "scripts", "run_case.py")]
    for case in cases:
        work = "/tmp/prof_%s_%s" % (tag, case)
        shutil.rmtree(work, ignore_errors=True)
        note = os.path.join(out, "%s_%s_abnormal_exit.txt" % (tag, case))
        r = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", work + "/trace", "--"] + prog + [case], note)
        runrec = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"case"')][-1])
        json.dump(runrec, open(os.path.join(out, "%s_%s_run.json" % (tag, case)), "w"), indent=1)
        ks = glob.glob(work + "/trace/**/*_kernel_stats.csv", recursive=True)[0]
        shutil.copy(ks, os.path.join(out, "%s_%s_kernel_stats.csv" % (tag, case)))
        calls = runrec["reps"] + runrec.get("warm", 2)  # run_case.py makes two warm-up calls
        pmc = {"case": case, "calls_profiled": calls}
        for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
            d = work + "/pmc_" + cnt
            run(["rocprofv3", "--kernel-trace", "--pmc", cnt, "--output-format", "csv", "-d", d, "--"] + prog + [case], note)
            f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
            per = collections.defaultdict(float)
            for x in csv.DictReader(open(f)):
                if runrec["kernel_pattern"] in x["Kernel_Name"]:
                    per[x["Kernel_Name"].split("(")[0][:90]] += float(x["Counter_Value"])
            pmc[cnt + "_KiB_per_call_by_kernel"] = {k: v / calls for k, v in per.items()}
            pmc[cnt + "_KiB_per_call"] = sum(per.values()) / calls
        pmc["fetch_bytes_per_call"] = 2 * pmc["FETCH_SIZE_KiB_per_call"] * 1024  # gfx950: 128-B fabric reads tallied at 64 B
        pmc["write_bytes_per_call"] = pmc["WRITE_SIZE_KiB_per_call"] * 1024
        pmc["hbm_bytes_per_call"] = pmc["fetch_bytes_per_call"] + pmc["write_bytes_per_call"]
        pmc["ms_avg_unprofiled_events"] = runrec["ms_avg"]
        pmc["traffic_GBps"] = pmc["hbm_bytes_per_call"] / runrec["ms_avg"] / 1e6
        if runrec.get("bytes_must_move"):
            pmc["traffic_over_bytes_must_move"] = pmc["hbm_bytes_per_call"] / runrec["bytes_must_move"]
        pmc["note"] = "FETCH_SIZE x2 (gfx950 correction, calibrated with scripts/calib/calib.hip); kernels matched by '%s'" % runrec["kernel_pattern"]
        json.dump(pmc, open(os.path.join(out, "%s_%s_pmc.json" % (tag, case)), "w"), indent=1)
        print(json.dumps(runrec)[:600], flush=True)
        print(json.dumps(pmc)[:900], flush=True)
    if ABNORMAL:
        print("abnormal exits of profiled runs:", ABNORMAL, flush=True)
        sys.exit(3)


if __name__ == "__main__":
    main()
