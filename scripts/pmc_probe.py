#!/usr/bin/env python3
"""PMC probe of the apply kernel under tuning variants: one rocprofv3 --pmc pass per counter group and variant, bench.py as
the profiled program (directly after --), per-launch averages of the *_apply* kernel printed as JSON lines.
usage: python scripts/pmc_probe.py out.jsonl "VAR=1,VAR2=3" "..." -- "CNT_A CNT_B" "CNT_C" ...   (bench args via BENCH_ARGS env)"""
import collections, csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def main():
    out = sys.argv[1]
    rest = sys.argv[2:]
    i = rest.index("--")
    variants, groups = rest[:i], rest[i + 1:]
    os.environ["TMPDIR"] = "/tmp"
    bargs = os.environ.get("BENCH_ARGS", "--steps 3 --warmup 1").split()
    with open(out, "a") as fo:
        for v in variants:
            env = dict(os.environ)
            for kv in filter(None, v.split(",")):
                k, val = kv.split("=")
                env["FIMEX_AMD_" + k] = val
            rec = {"variant": v}
            for g in groups:
                work = "/tmp/pmc_probe"
                shutil.rmtree(work, ignore_errors=True)
                cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + g.split() + ["--output-format", "csv", "-d", work, "--",
                       sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-seconds", "0", "--no-extras", "--no-verify", "--tuning-build"] + bargs
                print("run:", v, "|", g, flush=True)
                try:
                    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd="/tmp", timeout=150)
                except subprocess.TimeoutExpired:
                    rec["error_" + g] = "timeout"
                    print("timeout", flush=True)
                    break  # no further GPU step after a hang
                if r.returncode != 0:
                    rec["error_" + g] = (r.stdout[-500:] + r.stderr[-500:])
                    continue
                f = glob.glob(work + "/**/*_counter_collection.csv", recursive=True)[0]
                per = collections.defaultdict(list)
                for x in csv.DictReader(open(f)):
                    if "_apply" in x["Kernel_Name"]:
                        per[x["Counter_Name"]].append(float(x["Counter_Value"]))
                        rec["kernel"] = x["Kernel_Name"].split("(")[0][:80]
                for k, vals in per.items():
                    rec[k] = sum(vals) / len(vals)
                try:
                    rec["ms_" + g.split()[0]] = json.loads(r.stdout.strip().splitlines()[-1])["roofline"]["kernel_ms_avg"]
                except Exception:
                    pass
            fo.write(json.dumps(rec) + "\n")
            fo.flush()
            print(json.dumps(rec), flush=True)

if __name__ == "__main__":
    main()
