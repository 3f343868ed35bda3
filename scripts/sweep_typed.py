#!/usr/bin/env python3
"""Interleaved timing of tuning variants of the stored-type regrid (fimex_amd_regrid_apply_typed_device) on the benchmark
geometry: 200 slices of packed shorts (or bytes), device resident, one plan per variant (the staged form of a stored type is
built at the first apply of that plan, under the switches set then).
usage: python scripts/sweep_typed.py [--method bilinear|nearest] [--dtype int16|uint8] [--nz 200] "VAR=1,VAR2=3" "" ..."""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nz", type=int, default=200)
    ap.add_argument("--method", default="bilinear")
    ap.add_argument("--dtype", default="int16")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("variants", nargs="*", default=[""])
    a = ap.parse_args()
    import torch
    from fimex_amd import capi as fa
    import workloads, bench
    fa.use_tuning_build(True)
    fa.load(); fa.set_device(0)
    st = torch.cuda.current_stream().cuda_stream
    wl = workloads.BilinearRotatedPole()
    method = {"bilinear": fa.BILINEAR, "nearest": fa.NEAREST_NEIGHBOR, "bicubic": fa.BICUBIC}[a.method]
    d_f = bench.make_slices(torch, wl.base_field(), a.nz)
    if a.dtype == "int16":
        d_in = (torch.nan_to_num(d_f, nan=-327.67) * 100).to(torch.int16); code, bad, eb = fa.CDM_SHORT, -32767.0, 2
    else:
        d_in = ((torch.nan_to_num(d_f, nan=200.0) - 200) * 1.2).clamp(0, 255).to(torch.uint8); code, bad, eb = fa.CDM_UCHAR, 0.0, 1
    del d_f
    d_out = torch.empty((a.nz, wl.outY, wl.outX), dtype=d_in.dtype, device="cuda")
    keys = {kv.split("=")[0] for v in a.variants for kv in filter(None, v.split(","))}

    def setenv(v):
        for k in keys:
            os.environ.pop("FIMEX_AMD_" + k, None)
        for kv in filter(None, v.split(",")):
            k, val = kv.split("=")
            os.environ["FIMEX_AMD_" + k] = val

    plans, ref = {}, None
    for v in a.variants:
        setenv(v)
        plans[v] = bench.build_plan(fa, torch, wl, method, st)[0]
        fa.regrid_apply_typed_device(plans[v], d_in.data_ptr(), code, a.nz, bad, d_out.data_ptr(), st)
        torch.cuda.synchronize()
        if ref is None:
            ref = d_out.clone()
        elif not torch.equal(ref, d_out):
            print("variant %r differs from the first one in %d elements" % (v, int((ref != d_out).sum())), flush=True)
    times = {v: [] for v in a.variants}
    for r in range(a.rounds + 1):
        for v in a.variants:
            setenv(v)
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fa.regrid_apply_typed_device(plans[v], d_in.data_ptr(), code, a.nz, bad, d_out.data_ptr(), st); e1.record()
                torch.cuda.synchronize()
                if r: times[v].append(e0.elapsed_time(e1))
    b_alg = a.nz * eb * (wl.inX * wl.inY + wl.outX * wl.outY)
    for v in a.variants:
        t = float(np.median(times[v]))
        print(json.dumps({"method": a.method, "dtype": a.dtype, "variant": v, "ms_median": t, "ms_min": float(np.min(times[v])),
                          "frac_of_8TBps_on_stored_bytes": b_alg / (t * 1e-3) / 8e12}), flush=True)


if __name__ == "__main__":
    main()
