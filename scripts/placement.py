#!/usr/bin/env python3
"""Where the output batch lies in device memory moves the headline launch by several per cent.  One script for
the experiments around that effect (it replaces round 2's scripts/bench_placement{,2..8}.py, whose results stay in
profiles/calib/r02_placement*.jsonl):

  windows   one plan, one source batch, the 3.2 GB output batch at --positions windows --step-mib apart inside ONE allocation:
            median launch time per window (optionally with the loads or the stores of the kernel switched off: --ablate,
            tuning build).  Prints one JSON line per window.  This is also the program profiled by `pmc`.
  realloc   the output batch freed and allocated again behind other allocations of varying size (a fresh torch allocation per
            trial): one JSON line per trial.
  matrix    K source batches x K output batches, every one its own allocation: the launch time of every pair (does the source's
            allocation matter as the output's does?).
  pmc       runs `windows` under rocprofv3 --pmc, one pass (= one process) per counter group, keeps the PER-INSTANCE values of
            every counter (JSON output) and relates them to the launch time of each window: which counter moves with the time?
            Writes <out>_<group>.json (per window: ms, per-counter sum / max / min over instances, the instance vector) and
            prints a summary.  The profiled program is the Python interpreter directly after `--`.

usage: python scripts/placement.py windows [--positions 10] [--step-mib 704] [--reps 5] [--ablate] [--shape 0|1]
       python scripts/placement.py realloc [--trials 8]
       python scripts/placement.py pmc --out gpurun_out/r03_placement_pmc [--positions 10] -- "CNT_A CNT_B" "CNT_C" ...
"""
import argparse
import glob
import gzip
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def setup(tuning):
    import numpy as np
    import torch
    from fimex_amd import capi as fa
    import workloads, bench
    if tuning:
        fa.use_tuning_build(True)
    fa.load()
    fa.set_device(0)
    st = torch.cuda.current_stream().cuda_stream
    wl = workloads.BilinearRotatedPole()
    plan, _, _ = bench.build_plan(fa, torch, wl, fa.BILINEAR, st)
    return np, torch, fa, wl, plan, st, bench


def median_ms(np, torch, plan, d_in, nz, out_ptr, st, reps):
    ts = []
    for r in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        plan.apply_device(d_in.data_ptr(), nz, out_ptr, st)
        b.record()
        torch.cuda.synchronize()
        if r > 0:
            ts.append(a.elapsed_time(b))
    return float(np.median(ts)), ts


def mode_windows(args):
    np, torch, fa, wl, plan, st, bench = setup(args.ablate or args.shape is not None)
    if args.shape is not None:
        os.environ["FIMEX_AMD_STAGE2_USE_ALT"] = str(args.shape)
    nz = args.nz
    nout = nz * wl.outX * wl.outY
    d_in = bench.make_slices(torch, wl.base_field(), nz)
    step = args.step_mib * 1024 * 1024 // 4
    arena = torch.empty(nout + (args.positions - 1) * step, dtype=torch.float32, device="cuda")
    print(json.dumps({"arena_ptr": hex(arena.data_ptr()), "source_ptr": hex(d_in.data_ptr()), "positions": args.positions,
                      "step_MiB": args.step_mib, "reps": args.reps, "launches_per_window": (args.reps + 1) * (4 if args.ablate else 1)}), flush=True)
    for k in range(args.positions):
        w = arena[k * step:k * step + nout]
        rec = {"window": k, "offset_MiB": k * args.step_mib}
        if args.ablate:
            for name, flag in (("complete", 0), ("stores_only", 1), ("loads_only", 2), ("neither", 3)):
                os.environ["FIMEX_AMD_STAGE2_ABLATE"] = str(flag)
                rec["ms_" + name], _ = median_ms(np, torch, plan, d_in, nz, w.data_ptr(), st, args.reps)
        else:
            rec["ms"], rec["ms_all"] = median_ms(np, torch, plan, d_in, nz, w.data_ptr(), st, args.reps)
        print(json.dumps(rec), flush=True)


def mode_realloc(args):
    np, torch, fa, wl, plan, st, bench = setup(False)
    nz = args.nz
    nout = nz * wl.outX * wl.outY
    d_in = bench.make_slices(torch, wl.base_field(), nz)
    keep = []
    for t in range(args.trials):
        out = torch.empty(nout, dtype=torch.float32, device="cuda")
        ms, _ = median_ms(np, torch, plan, d_in, nz, out.data_ptr(), st, args.reps)
        print(json.dumps({"trial": t, "out_ptr": hex(out.data_ptr()), "ms": ms}), flush=True)
        del out
        keep.append(torch.empty((t + 1) * 300 * 1024 * 1024 // 4, dtype=torch.float32, device="cuda"))  # shifts the next one
        torch.cuda.empty_cache()


def mode_matrix(args):
    """Does the SOURCE batch's allocation matter too?  K source batches and K output batches, each its own allocation (separate
    hipMalloc calls: different physical memory), the launch timed for every pair."""
    np, torch, fa, wl, plan, st, bench = setup(False)
    nz, K = args.nz, args.trials
    nout = nz * wl.outX * wl.outY
    first = bench.make_slices(torch, wl.base_field(), nz)
    srcs, outs, spacers = [first], [], []
    for k in range(1, K):
        spacers.append(torch.empty((k * 173) << 20, dtype=torch.uint8, device="cuda"))  # shifts what the allocator hands out next
        srcs.append(first.clone())
    for k in range(K):
        spacers.append(torch.empty((k * 211 + 64) << 20, dtype=torch.uint8, device="cuda"))
        outs.append(torch.empty(nout, dtype=torch.float32, device="cuda"))
    print(json.dumps({"sources": [hex(t.data_ptr()) for t in srcs], "outputs": [hex(t.data_ptr()) for t in outs]}), flush=True)
    for i, src in enumerate(srcs):
        row = []
        for j, out in enumerate(outs):
            ms, _ = median_ms(np, torch, plan, src, nz, out.data_ptr(), st, args.reps)
            row.append(round(ms, 4))
        print(json.dumps({"source": i, "ms_by_output": row}), flush=True)


def parse_counter_json(path):
    """rocprofv3 --output-format json: per dispatch of the apply kernel, {counter name: [value per instance record]}."""
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt") as f:
        doc = json.load(f)
    tool = doc["rocprofiler-sdk-tool"]
    tool = tool[0] if isinstance(tool, list) else tool
    names = {}
    for c in tool.get("counters", []):
        cid = c.get("id", {})
        names[cid.get("handle") if isinstance(cid, dict) else cid] = c.get("name")
    kernels = {}
    for k in tool.get("kernel_symbols", []):
        kernels[k.get("kernel_id")] = k.get("formatted_kernel_name") or k.get("kernel_name") or ""
    out = []
    for rec in tool.get("callback_records", {}).get("counter_collection", []):
        info = rec.get("dispatch_data", {}).get("dispatch_info", {})
        kname = kernels.get(info.get("kernel_id"), "")
        if "_apply" not in kname:
            continue
        per = {}
        for r in rec.get("records", []):
            cid = r.get("counter_id", {})
            h = cid.get("handle") if isinstance(cid, dict) else cid
            per.setdefault(names.get(h, str(h)), []).append(float(r.get("value", 0.0)))
        out.append({"dispatch_id": info.get("dispatch_id"), "kernel": kname.split("(")[0][-60:], "counters": per})
    out.sort(key=lambda d: d["dispatch_id"])
    return out


def mode_pmc(args, groups):
    os.environ["TMPDIR"] = "/tmp"
    os.makedirs(os.path.dirname(os.path.abspath(args.out)) or ".", exist_ok=True)
    summary = []
    for gi, g in enumerate(groups):
        work = "/tmp/placement_pmc_%d" % gi
        shutil.rmtree(work, ignore_errors=True)
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + g.split() + ["--output-format", "json", "csv", "-d", work, "--",
               sys.executable, os.path.abspath(__file__), "windows", "--positions", str(args.positions), "--step-mib", str(args.step_mib),
               "--reps", str(args.reps), "--nz", str(args.nz)]
        print("run:", g, flush=True)
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", timeout=240)
        except subprocess.TimeoutExpired:
            print("timeout: no further GPU step", flush=True)
            break
        if r.returncode != 0:
            print("failed:", r.stdout[-1500:], r.stderr[-1500:], flush=True)
            continue
        windows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{"window"')]
        tag = "%s_%s" % (args.out, g.split()[0])
        js = glob.glob(work + "/**/*_results.json", recursive=True)
        if not js:
            print("no JSON output under", work, os.listdir(work), flush=True)
            continue
        with open(js[0], "rb") as fi, gzip.open(tag + "_raw.json.gz", "wb") as fo:
            shutil.copyfileobj(fi, fo)  # raw per-instance records, kept for offline reading
        try:
            disp = parse_counter_json(js[0])
        except Exception as e:
            print("could not parse %s: %r" % (js[0], e), flush=True)
            continue
        per_win = args.reps + 1
        rows = []
        for w in windows:
            mine = disp[w["window"] * per_win + 1:(w["window"] + 1) * per_win]  # the timed launches of this window
            row = {"window": w["window"], "ms": w["ms"]}
            for cname in sorted({c for d in mine for c in d["counters"]}):
                vecs = [d["counters"][cname] for d in mine if cname in d["counters"]]
                n = min(len(v) for v in vecs)
                mean = [sum(v[i] for v in vecs) / len(vecs) for i in range(n)]
                row[cname] = {"sum": sum(mean), "max": max(mean), "min": min(mean), "instances": n, "per_instance": mean}
            rows.append(row)
        json.dump({"group": g, "launches_of_apply_kernel": len(disp), "windows": rows}, open(tag + ".json", "w"))
        for row in rows:
            line = {"group": g.split()[0], "window": row["window"], "ms": round(row["ms"], 4)}
            for k, v in row.items():
                if isinstance(v, dict):
                    line[k] = {"sum": v["sum"], "max/mean": v["max"] / (v["sum"] / v["instances"]) if v["sum"] else None, "n": v["instances"]}
            summary.append(line)
            print(json.dumps(line), flush=True)
    json.dump(summary, open(args.out + "_summary.json", "w"), indent=1)


def main():
    argv = sys.argv[1:]
    groups = []
    if "--" in argv:
        i = argv.index("--")
        argv, groups = argv[:i], argv[i + 1:]
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["windows", "realloc", "pmc", "matrix"])
    ap.add_argument("--positions", type=int, default=10)
    ap.add_argument("--step-mib", type=int, default=704)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--nz", type=int, default=200)
    ap.add_argument("--trials", type=int, default=8)
    ap.add_argument("--ablate", action="store_true")
    ap.add_argument("--shape", type=int, default=None)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "placement_pmc"))
    args = ap.parse_args(argv)
    if args.mode == "windows":
        mode_windows(args)
    elif args.mode == "realloc":
        mode_realloc(args)
    elif args.mode == "matrix":
        mode_matrix(args)
    else:
        mode_pmc(args, groups or ["TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL", "TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL", "TCC_EA0_WRREQ_LEVEL TCC_TOO_MANY_EA_WRREQS_STALL",
                                  "TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL"])


if __name__ == "__main__":
    main()
