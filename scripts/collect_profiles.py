#!/usr/bin/env python3
"""rocprofv3 evidence for a bench.py line AS THE DRIVER RUNS IT (default: the library places the output batch, the plan's workgroup
shape is tuned, every extra is measured), written to gpurun_out/profiles_<tag>/ :
  <tag>_<name>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of the whole command (every launch of every kernel:
                                  the probing and tuning launches, the verification and the extras are in these averages)
  <tag>_<name>_timed.json         the K timed launches alone, picked from the per-dispatch trace: the K launches of the apply
                                  kernel that precede the first launch of the gather kernel (bench.py's verification follows the
                                  timed steps directly); average, min, max -- this is the figure that has to agree with the
                                  bench line's roofline.kernel_ms_avg and must not exceed its ms_per_step
  <tag>_<name>_pmc.json           HBM-side bytes of those same K launches from separate --pmc passes (FETCH_SIZE doubled: the
                                  gfx950 correction of MI355X_MICROARCH.md, calibrated with scripts/calib/calib.hip; WRITE_SIZE as is)
  <tag>_<name>_bench.json         the bench JSON line of the traced run
usage: python scripts/collect_profiles.py <tag> <name> [bench.py args...]     (bench.py runs with --steps 20 --warmup 5)
Copy the files into profiles/ afterwards; scripts/assemble_traffic.py builds profiles/hbm_traffic.json from the *_pmc.json files."""
import csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS, WARMUP = 20, 5


def run(cmd):
    print("run:", " ".join(cmd[:7]), "...", flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", timeout=400)
    if r.returncode != 0:
        print(r.stdout[-3000:], r.stderr[-3000:])
        sys.exit(1)
    return r


def timed_dispatches(rows, name_key):
    """rows of a per-dispatch CSV in dispatch order -> the STEPS launches of the staged apply kernel before the first gather launch."""
    rows = sorted(rows, key=lambda x: int(x.get("Dispatch_Id") or x.get("Dispatch_ID") or 0))
    is_gather = lambda n: ("bilinear_apply" in n or "nearest_apply" in n or "bicubic_apply" in n) and "staged" not in n
    first_gather = next((i for i, x in enumerate(rows) if is_gather(x[name_key])), len(rows))
    staged = [x for x in rows[:first_gather] if "staged_apply" in x[name_key]]
    return staged[-STEPS:]


def main():
    tag, name, bargs = sys.argv[1], sys.argv[2], sys.argv[3:]
    out = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)
    os.makedirs(out, exist_ok=True)
    os.environ["TMPDIR"] = "/tmp"
    work = os.path.join("/tmp", "prof_%s_%s" % (tag, name))
    shutil.rmtree(work, ignore_errors=True)
    bench = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(STEPS), "--warmup", str(WARMUP), "--cpu-seconds", "0"] + bargs
    r = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", work + "/trace", "--"] + bench)
    line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1]
    open(os.path.join(out, "%s_%s_bench.json" % (tag, name)), "w").write(line + "\n")
    b = json.loads(line)
    shutil.copy(glob.glob(work + "/trace/**/*_kernel_stats.csv", recursive=True)[0], os.path.join(out, "%s_%s_kernel_stats.csv" % (tag, name)))
    tr = list(csv.DictReader(open(glob.glob(work + "/trace/**/*_kernel_trace.csv", recursive=True)[0])))
    t = timed_dispatches(tr, "Kernel_Name")
    dur = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) * 1e-6 for x in t]
    timed = {"kernel": t[0]["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0] if t else None, "launches": len(dur), "ms_avg": sum(dur) / max(len(dur), 1),
             "ms_min": min(dur or [0]), "ms_max": max(dur or [0]), "bench_kernel_ms_avg": b["roofline"]["kernel_ms_avg"], "bench_ms_per_step": b["ms_per_step"],
             "bench_frac": b["roofline"]["frac"], "source_placement": b["config"].get("source_placement"), "output_placement": b["config"].get("output_placement"), "tuned_shape": b["config"].get("tuned_shape"),
             "how": "the %d launches of the staged apply kernel that precede the first gather-kernel launch (the verification) in the per-dispatch trace" % STEPS}
    json.dump(timed, open(os.path.join(out, "%s_%s_timed.json" % (tag, name)), "w"), indent=1)
    pmc = {"kernel": timed["kernel"], "launches_counted": STEPS}
    for grp in (["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"], ["TCC_REQ_sum", "TCC_MISS_sum", "TCC_HIT_sum"]):
        d = work + "/pmc_" + grp[0]
        rp = run(["rocprofv3", "--kernel-trace", "--pmc"] + grp + ["--output-format", "csv", "-d", d, "--"] + bench + ["--no-extras"])
        rows = list(csv.DictReader(open(glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0])))
        for c in grp:
            tc = timed_dispatches([x for x in rows if x["Counter_Name"] == c], "Kernel_Name")
            vals = [float(x["Counter_Value"]) for x in tc]
            pmc[c] = sum(vals) / max(len(vals), 1)
            pmc[c + "_launches"] = len(vals)
        try:
            pmc["ms_per_step_in_pass_" + grp[0]] = json.loads([l for l in rp.stdout.splitlines() if l.startswith('{"metric"')][-1])["ms_per_step"]
        except Exception:
            pass
    pmc["fetch_bytes_per_launch"] = 2 * pmc["FETCH_SIZE"] * 1024  # gfx950: 128-B fabric reads tallied at 64 B
    pmc["write_bytes_per_launch"] = pmc["WRITE_SIZE"] * 1024
    pmc["hbm_bytes_per_launch"] = pmc["fetch_bytes_per_launch"] + pmc["write_bytes_per_launch"]
    pmc["algorithmic_bytes_per_launch"] = b["roofline"]["algorithmic_bytes_per_launch"]
    pmc["traffic_over_algorithmic"] = pmc["hbm_bytes_per_launch"] / b["roofline"]["algorithmic_bytes_per_launch"]
    pmc["tile"] = b["roofline"]["tile"]
    pmc["note"] = ("FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE x2 per MI355X_MICROARCH.md (HBM section), confirmed by scripts/calib/calib.hip "
                   "(4 GiB read -> FETCH_SIZE 2097167 KiB in every access shape tried); counters of the timed launches only")
    json.dump(pmc, open(os.path.join(out, "%s_%s_pmc.json" % (tag, name)), "w"), indent=1)
    print(open(os.path.join(out, "%s_%s_kernel_stats.csv" % (tag, name))).read()[:1200])
    print(json.dumps(timed, indent=1))
    print(json.dumps(pmc, indent=1))


if __name__ == "__main__":
    main()
