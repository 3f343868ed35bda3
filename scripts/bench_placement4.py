#!/usr/bin/env python3
"""Is some device memory slower to write than other?  A 3.2 GB window slides through one 20 GiB allocation in steps of 512 MiB:
time of a plain fill (write), of a sum (read) and of the headline launch writing its output there (source fixed elsewhere).
usage: python scripts/bench_placement4.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fimex_amd import capi as fa
import workloads, bench
fa.use_tuning_build(True)
fa.load(); fa.set_device(0)
st = torch.cuda.current_stream().cuda_stream
wl = workloads.BilinearRotatedPole()
nz = 200
plan, _, _ = bench.build_plan(fa, torch, wl, fa.BILINEAR, st)
nin, nout = nz * wl.inX * wl.inY, nz * wl.outX * wl.outY
base = bench.make_slices(torch, wl.base_field(), nz).view(-1)
os.environ["FIMEX_AMD_STAGE2_USE_ALT"] = "0"

def med(fn, reps=6):
    ts = []
    for r in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        if r >= 2: ts.append(a.elapsed_time(b))
    return float(np.median(ts))

step = 512 * 1024 * 1024 // 4
big = torch.empty(20 * 1024 * 1024 * 1024 // 4, dtype=torch.float32, device="cuda")
big.zero_()
free, total = torch.cuda.mem_get_info()
print(json.dumps({"free_GiB": free / 2 ** 30, "total_GiB": total / 2 ** 30, "big_ptr": hex(big.data_ptr()), "source_ptr": hex(base.data_ptr())}), flush=True)
k = 0
while k * step + nout <= big.numel():
    w = big[k * step:k * step + nout]
    print(json.dumps({"offset_MiB": 512 * k, "fill_ms": med(lambda: w.fill_(1.0)), "sum_ms": med(lambda: w.sum()),
                      "regrid_ms": med(lambda: plan.apply_device(base.data_ptr(), nz, w.data_ptr(), st))}), flush=True)
    k += 1
