#!/usr/bin/env python3
"""The time of the headline launch against the distance between its source and output batches: views at offsets of 0 .. 2 GiB
(steps of 64 MiB) into one oversized output allocation, the source fixed; then the same for the source with the output fixed.
usage: python scripts/bench_placement3.py"""
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
step = 64 * 1024 * 1024 // 4
nsteps = 33
base = bench.make_slices(torch, wl.base_field(), nz).view(-1)

def timed(d_in, d_out, shape):
    os.environ["FIMEX_AMD_STAGE2_USE_ALT"] = str(shape)
    ts = []
    for r in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); plan.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), st); b.record(); torch.cuda.synchronize()
        if r >= 2: ts.append(a.elapsed_time(b))
    return float(np.median(ts))

big_out = torch.empty(nout + step * nsteps, dtype=torch.float32, device="cuda")
for k in range(nsteps):
    d_out = big_out[k * step:k * step + nout]
    print(json.dumps({"moved": "output", "offset_MiB": 64 * k, "distance_in_minus_out_MiB": (base.data_ptr() - d_out.data_ptr()) / 2 ** 20,
                      "ms_1024_threads": timed(base, d_out, 0), "ms_512_threads": timed(base, d_out, 1)}), flush=True)
d_out = big_out[:nout]
del base
big_in = torch.empty(nin + step * nsteps, dtype=torch.float32, device="cuda")
src = bench.make_slices(torch, wl.base_field(), 1).view(-1)
for k in range(nsteps):
    d_in = big_in[k * step:k * step + nin]
    d_in.view(nz, -1).copy_(src.view(1, -1).expand(nz, -1))
    print(json.dumps({"moved": "source", "offset_MiB": 64 * k, "distance_in_minus_out_MiB": (d_in.data_ptr() - d_out.data_ptr()) / 2 ** 20,
                      "ms_1024_threads": timed(d_in, d_out, 0), "ms_512_threads": timed(d_in, d_out, 1)}), flush=True)
