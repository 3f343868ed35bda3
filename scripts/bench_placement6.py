#!/usr/bin/env python3
"""Which half of the headline launch feels the placement of the output batch?  The sliding 3.2 GB output window of
bench_placement4.py with the source loads switched off (stores only), the stores switched off (loads only) and neither (tuning
build, STAGE2_ABLATE).
usage: python scripts/bench_placement6.py"""
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

def med(ablate, w):
    os.environ["FIMEX_AMD_STAGE2_ABLATE"] = str(ablate)
    ts = []
    for r in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); plan.apply_device(base.data_ptr(), nz, w.data_ptr(), st); b.record(); torch.cuda.synchronize()
        if r >= 2: ts.append(a.elapsed_time(b))
    return float(np.median(ts))

step = 1024 * 1024 * 1024 // 4
big = torch.empty(20 * 1024 * 1024 * 1024 // 4, dtype=torch.float32, device="cuda")
k = 0
while k * step + nout <= big.numel():
    w = big[k * step:k * step + nout]
    print(json.dumps({"offset_GiB": k, "ms_complete": med(0, w), "ms_stores_only": med(1, w), "ms_loads_only": med(2, w), "ms_neither": med(3, w)}), flush=True)
    k += 1
