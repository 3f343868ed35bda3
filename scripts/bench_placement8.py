#!/usr/bin/env python3
"""Source and output batch inside ONE allocation: output directly below the source, directly above it, 1 GiB and 4 GiB below it,
against two separate allocations; four fresh allocations of each layout with other allocations in between.
usage: python scripts/bench_placement8.py"""
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
src = bench.make_slices(torch, wl.base_field(), 1).view(1, -1)
GiB = 2 ** 30 // 4

def timed(d_in, d_out, shape):
    os.environ["FIMEX_AMD_STAGE2_USE_ALT"] = str(shape)
    ts = []
    for r in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); plan.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), st); b.record(); torch.cuda.synchronize()
        if r >= 2: ts.append(a.elapsed_time(b))
    return float(np.median(ts))

for trial in range(4):
    for label in ("separate", "out|in", "in|out", "out|1GiB|in", "out|4GiB|in"):
        torch.cuda.empty_cache()
        junk = torch.empty((91 + 257 * trial) * 262144, dtype=torch.float32, device="cuda")
        if label == "separate":
            a_in = torch.empty(nin, dtype=torch.float32, device="cuda"); a_out = torch.empty(nout, dtype=torch.float32, device="cuda")
            d_in, d_out = a_in, a_out
        else:
            gap = {"out|in": 0, "in|out": 0, "out|1GiB|in": GiB, "out|4GiB|in": 4 * GiB}[label]
            arena = torch.empty(nin + nout + gap, dtype=torch.float32, device="cuda")
            if label == "in|out": d_in, d_out = arena[:nin], arena[nin:nin + nout]
            else: d_out, d_in = arena[:nout], arena[nout + gap:nout + gap + nin]
        d_in.view(nz, -1).copy_(src.expand(nz, -1))
        print(json.dumps({"trial": trial, "layout": label, "in_minus_out_MiB": (d_in.data_ptr() - d_out.data_ptr()) / 2 ** 20,
                          "ms_1024_threads": timed(d_in, d_out, 0), "ms_512_threads": timed(d_in, d_out, 1)}), flush=True)
        del d_in, d_out, junk
        if label == "separate": del a_in, a_out
        else: del arena
