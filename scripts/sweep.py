#!/usr/bin/env python3
"""Interleaved A/B timing of tuning variants of one apply kernel on the bench workload (one process, one GPU).
usage: python scripts/sweep.py [--nz 200] [--method bilinear] "ZPB=40,BILINEAR_ZC=8" "ZPB=20" ...
Each variant is a comma-separated list of FIMEX_AMD_<NAME>=value settings ("" = defaults)."""
import argparse, os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nz", type=int, default=200)
    ap.add_argument("--method", default="bilinear")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--bicubic-fast", action="store_true")
    ap.add_argument("--resident", type=int, default=0,
                    help="slices resident in HBM (default: nz); larger than nz: every launch takes the next nz slices, so short batches are timed cold")
    ap.add_argument("variants", nargs="*", default=[""])
    a = ap.parse_args()
    import torch
    from fimex_amd import capi as fa
    import workloads, bench
    fa.use_tuning_build(True)  # the build that reads the FIMEX_AMD_<NAME> switches
    fa.load(); fa.set_device(0)
    stream = torch.cuda.current_stream().cuda_stream
    wl = workloads.BilinearRotatedPole()
    method = {"bilinear": fa.BILINEAR, "bicubic": fa.BICUBIC, "nearest": fa.NEAREST_NEIGHBOR}[a.method]
    plans = {}
    res = max(a.resident, a.nz)
    d_in = bench.make_slices(torch, wl.base_field(), res)
    d_out = torch.empty((res, wl.outY, wl.outX), dtype=torch.float32, device="cuda")
    in_b, out_b = 4 * wl.inX * wl.inY, 4 * wl.outX * wl.outY
    turn = [0]
    keys = set()
    for v in a.variants:
        for kv in filter(None, v.split(",")):
            keys.add(kv.split("=")[0])
    def setenv(v):
        for k in keys:
            os.environ.pop("FIMEX_AMD_" + k, None)
        for kv in filter(None, v.split(",")):
            k, val = kv.split("=")
            os.environ["FIMEX_AMD_" + k] = val
    for v in a.variants:  # tile shape knobs (STAGE_TW, STAGE_PER, STAGE_K) act when the plan is built
        setenv(v)
        plans[v] = bench.build_plan(fa, torch, wl, method, stream, bicubic=fa.BICUBIC_FAST if a.bicubic_fast else None)[0]
        info = plans[v].info()
        print("plan %-30s tile %sx%s staged cells %s" % (v or "(defaults)", info.get("tileW"), info.get("tileH"), info.get("stagedCells")), flush=True)
    times = {v: [] for v in a.variants}
    for r in range(a.rounds + 1):
        for v in a.variants:
            setenv(v)
            for _ in range(a.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                k = (turn[0] * a.nz) % (res - a.nz + 1)
                turn[0] += 1
                plans[v].apply_device(d_in.data_ptr() + k * in_b, a.nz, d_out.data_ptr() + k * out_b, stream)
                e1.record()
                torch.cuda.synchronize()
                if r > 0:
                    times[v].append(e0.elapsed_time(e1))
    cells = a.nz * wl.outX * wl.outY
    for v in a.variants:
        t = np.array(times[v])
        print("%-40s median %.3f ms  min %.3f ms  -> %.0f Mcells/s" % (v or "(defaults)", np.median(t), t.min(), cells / np.median(t) / 1e3), flush=True)

if __name__ == "__main__":
    main()
